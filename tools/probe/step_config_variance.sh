#!/bin/bash
# run-to-run spread of step_kernel (W waves per lane, G species per wave) on the headline shape, with and without LDS balancing
run() {  # W G steps warmup nobalance
  if [ "$5" = "1" ]; then export CATINT_PNP_NO_LDS_BALANCE=1; else unset CATINT_PNP_NO_LDS_BALANCE; fi
  CATINT_PNP_KERNEL=2 CATINT_PNP_WAVES_PER_GRID=$1 CATINT_PNP_SPECIES_PER_WAVE=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --large-batch 0 --physical-steps 0 --steps $3 --warmup $4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('W=$1 G=$2 steps=$3 warm=$4 nobalance=$5  %.3e  frac %.3f' % (d['value'], d['roofline']['frac']))"
}
for i in 1 2 3 4 5; do run 3 1 256 64 0; run 3 1 256 64 1; run 1 3 256 64 0; run 1 3 256 64 1; done
