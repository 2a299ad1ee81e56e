#!/bin/bash
# Lane kernel (pnp_lane.hip): parity tests, then timesteps/s against the default kernel choice on the VERDICT r02 shapes.
# usage (GPU box): bash tools/probe/lane_probe.sh [quick]
set -o pipefail
O=gpurun_out/lane_probe.txt
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_lane.py -x -q 2>&1 | tail -15 | tee -a $O || exit 1
for spec in "8 512 8192" "6 1024 32768" "8 512 32768" "8 4096 1024" "3 512 8192"; do
  set -- $spec
  for k in lane ""; do
    echo "== N=$1 nx=$2 B=$3 kernel=${k:-default}" | tee -a $O
    CATINT_NEWTON_KERNEL=$k timeout -k 10 300 python tools/newton_bench.py --nspecies $1 --nx $2 --batch $3 --steps 10 --warmup 2 --stern --mpb 2>&1 | tail -2 | tee -a $O || exit 1
  done
done
