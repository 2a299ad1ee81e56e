#!/bin/bash
# lane-pair kernel (pnp_lane2.hip) against the lane kernel and the lane-team kernels, one device, one call
# usage: bash tools/probe/lane2_probe.sh ["N NX B" ...]
if [ $# -eq 0 ]; then set -- "8 512 1024" "8 512 2048" "8 512 4096" "8 512 8192" "8 512 16384" "8 512 32768" "6 1024 8192" "6 1024 32768" "8 4096 1024" "8 4096 8192"; fi
for spec in "$@"; do
  read N NX B <<< "$spec"
  line="N=$N nx=$NX B=$B:"
  for k in lane2 lane team; do
    r=$(CATINT_NEWTON_KERNEL=$k timeout -k 10 300 python tools/newton_bench.py --nspecies $N --nx $NX --batch $B --steps 8 --warmup 2 --stern --mpb 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g' % d['timesteps_per_s'])")
    line="$line $k $r"
  done
  echo "$line"
done
