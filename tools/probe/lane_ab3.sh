#!/bin/bash
# A/B of builds on ONE device in ONE call at B = 8192 (latency-bound: reproducible to ~0.3 %), lane and lane-pair kernel
LIBS=("$@")
for K in lane lane2; do
  for round in 1 2 3; do
    for lib in "${LIBS[@]}"; do
      r=$(CATINT_PNP_LIB=$PWD/$lib CATINT_NEWTON_KERNEL=$K python tools/newton_bench.py --nspecies 8 --nx 512 --batch 8192 --steps 10 --warmup 2 --stern --mpb 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g' % d['timesteps_per_s'])")
      echo "$K N=8 nx=512 B=8192 round $round $lib: $r"
    done
  done
done
