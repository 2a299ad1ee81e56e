#!/bin/bash
# A/B of pair-kernel builds (N = 4: 5 x 5 blocks) on ONE device in ONE call.
# usage: bash tools/probe/pair_ab.sh libA.so libB.so ...   (paths relative to the repo root), two interleaved rounds per shape
LIBS=("$@")
for spec in "4 512 1024 --stern" "4 512 1024 --stern --mpb" "4 512 1024 --reactions" "4 512 4096 --stern" "4 256 2048 --stern" "4 128 4096 --stern" "4 128 4096 --stern --mpb" "4 64 8192 --stern"; do
  read N NX B FLAGS <<< "$spec"
  for round in 1 2; do
    for lib in "${LIBS[@]}"; do
      r=$(CATINT_PNP_LIB=$PWD/$lib python tools/newton_bench.py --nspecies $N --nx $NX --batch $B --steps 20 --warmup 3 $FLAGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g steps/s %.4g it/s' % (d['timesteps_per_s'], d.get('newton_iterations_per_s', 0)))")
      echo "N=$N nx=$NX B=$B $FLAGS round $round $lib: $r"
    done
  done
done
