#!/bin/bash
# A/B of pair-kernel builds at N = 3 (4 x 4 blocks) on ONE device in ONE call.
LIBS=("$@")
for spec in "3 512 1024 --stern --mpb" "3 512 4096 --stern --mpb" "3 512 1024 --reactions" "3 256 2048 --stern --mpb" "3 128 4096 --stern --mpb" "3 512 1024 --stern"; do
  read N NX B FLAGS <<< "$spec"
  for round in 1 2; do
    for lib in "${LIBS[@]}"; do
      r=$(CATINT_PNP_LIB=$PWD/$lib python tools/newton_bench.py --nspecies $N --nx $NX --batch $B --steps 20 --warmup 3 $FLAGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g steps/s %.4g it/s' % (d['timesteps_per_s'], d.get('newton_iterations_per_s', 0)))")
      echo "N=$N nx=$NX B=$B $FLAGS round $round $lib: $r"
    done
  done
done
