#!/bin/bash
# wider randomised campaign on the GPU box: physical mode with the lane / lane-pair kernel forced (N >= 2 / N >= 5 so that the forced kernel
# takes most configurations), ten new seeds each; default kernels, ten new seeds; compat, five new seeds
set -o pipefail
O=gpurun_out/fuzz_more; mkdir -p $O
for s in 21 22 23 24 25 26 27 28 29 30; do CATINT_NEWTON_KERNEL=lane FUZZ_NMIN=2 FUZZ_SEED=$s FUZZ_CASES=120 timeout -k 10 300 python tests/fuzz/fuzz_newton.py 2>&1 | grep -v "^ok " ; done > $O/lane.txt
for s in 31 32 33 34 35 36 37 38 39 40; do CATINT_NEWTON_KERNEL=lane2 FUZZ_NMIN=5 FUZZ_SEED=$s FUZZ_CASES=120 timeout -k 10 300 python tests/fuzz/fuzz_newton.py 2>&1 | grep -v "^ok " ; done > $O/lane2.txt
for s in 41 42 43 44 45 46 47 48 49 50; do FUZZ_SEED=$s FUZZ_CASES=120 timeout -k 10 300 python tests/fuzz/fuzz_newton.py 2>&1 | grep -v "^ok " ; done > $O/default.txt
for s in 6 7 8 9 10; do FUZZ_SEED=$s FUZZ_CASES=150 timeout -k 10 300 python tests/fuzz/fuzz_compat.py 2>&1 | grep -v "^ok " ; done > $O/compat.txt
for f in lane lane2 default compat; do echo "## $f"; cat $O/$f.txt; done
