#!/bin/bash
# A/B of lane-kernel builds on ONE device in ONE call at the HBM-bound shapes (the rate there is bimodal from run to run: six runs each)
LIBS=("$@")
for spec in "8 512 32768" "6 1024 32768" "8 512 65536"; do
  read N NX B <<< "$spec"
  for round in 1 2 3 4 5 6; do
    for lib in "${LIBS[@]}"; do
      r=$(CATINT_PNP_LIB=$PWD/$lib CATINT_NEWTON_KERNEL=lane python tools/newton_bench.py --nspecies $N --nx $NX --batch $B --steps 10 --warmup 2 --stern --mpb 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g' % d['timesteps_per_s'])")
      echo "N=$N nx=$NX B=$B round $round $lib: $r"
    done
  done
done
