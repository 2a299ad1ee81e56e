#!/bin/bash
# sweep of the (waves per lane, species per wave) instances of step_kernel on bench shapes; prints roofline fractions
run() {  # N nx B spl W G
  CATINT_PNP_KERNEL=2 CATINT_PNP_WAVES_PER_GRID=$5 CATINT_PNP_SPECIES_PER_WAVE=$6 timeout -k 10 120 python bench.py --no-cpu-baseline --large-batch 0 --physical-steps 0 \
     --nspecies $1 --nx $2 --batch $3 --steps-per-launch $4 --steps $7 --warmup $8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$1 nx=$2 B=$3 spl=$4 W=$5 G=$6  %.3e  frac %.3f' % (d['value'], d['roofline']['frac']))"
}
for B in 1024 2048 4096 8192 16384; do
  for wg in "1 1" "1 2" "1 3" "2 1" "2 2" "3 1" "4 1"; do set -- $wg; run 3 512 $B 256 $1 $2 512 256; done
done
for wg in "1 1" "1 2" "1 3" "2 1" "2 2" "3 1" "4 1"; do set -- $wg; run 3 512 1024 1 $1 $2 256 64; done
