#!/bin/bash
# Lane kernel (pnp_lane.hip): parity tests, then timesteps/s against the default kernel choice on the VERDICT r02 shapes.
# usage (GPU box): bash tools/probe/lane_probe.sh [perf|quick]     perf: no tests; quick: lane kernel only on the two headline shapes
set -o pipefail
O=gpurun_out/lane_probe.txt
: > $O
if [ "$1" != perf ] && [ "$1" != quick ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_lane.py -x -q 2>&1 | tail -15 | tee -a $O || exit 1
fi
SHAPES=("8 512 8192" "6 1024 32768" "8 512 32768" "8 4096 1024" "3 512 8192")
KERNELS=(lane "")
if [ "$1" = quick ]; then SHAPES=("8 512 8192" "8 512 32768" "6 1024 32768"); KERNELS=(lane); fi
for spec in "${SHAPES[@]}"; do
  set -- $spec
  for k in "${KERNELS[@]}"; do
    echo "== N=$1 nx=$2 B=$3 kernel=${k:-default}" | tee -a $O
    CATINT_NEWTON_KERNEL=$k timeout -k 10 300 python tools/newton_bench.py --nspecies $1 --nx $2 --batch $3 --steps 10 --warmup 2 --stern --mpb 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('   timesteps/s %.4g  ms/step %.3f  its/step %.3f ok %d' % (d['timesteps_per_s'], d['ms_per_step'], d['mean_newton_iterations_per_step'], d['lanes_ok']))" | tee -a $O || exit 1
  done
done
