#!/bin/bash
# lane-team kernel (block cyclic reduction, a workgroup per operating point) against the sweep kernel (block Thomas, a team per
# operating point) over batch sizes
run() {  # N nx B steps kernel
  if [ "$5" = "default" ]; then unset CATINT_NEWTON_KERNEL; else export CATINT_NEWTON_KERNEL=$5; fi
  timeout -k 10 400 python tools/newton_bench.py --nspecies $1 --nx $2 --batch $3 --steps $4 --warmup 1 --mpb --stern > /tmp/o.json 2>/tmp/o.err && python -c "
import json; d=json.loads(open('/tmp/o.json').read()); print('N=$1 nx=$2 B=$3 $5: its/s %.3g  timesteps/s %.3g  ok %d/%d' % (d['newton_iterations_per_s'], d['timesteps_per_s'], d['lanes_ok'], $3))" || tail -3 /tmp/o.err; if grep -q HSA_STATUS /tmp/o.err; then exit 1; fi
}
for cfg in "8 4096 1024 2" "8 4096 8192 2" "6 1024 4096 3" "6 1024 32768 3" "8 512 8192 4" "8 512 65536 4" "6 512 16384 4" "4 512 16384 4" "3 512 16384 4"; do
  set -- $cfg
  run $1 $2 $3 $4 team; run $1 $2 $3 $4 sweep
done
