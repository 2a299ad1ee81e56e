set -o pipefail
for s in 1 2 3 4 5; do FUZZ_SEED=$s FUZZ_CASES=120 timeout -k 10 300 python tests/fuzz/fuzz_newton.py 2>&1 | tail -1; done > gpurun_out/fuzz_r03b_newton.txt
for k in lane lane2; do for s in 11 12 13; do CATINT_NEWTON_KERNEL=$k FUZZ_SEED=$s FUZZ_CASES=120 timeout -k 10 300 python tests/fuzz/fuzz_newton.py 2>&1 | tail -1; done; done > gpurun_out/fuzz_r03b_newton_lane.txt
for s in 1 2 3 4 5; do FUZZ_SEED=$s FUZZ_CASES=150 timeout -k 10 300 python tests/fuzz/fuzz_compat.py 2>&1 | tail -1; done > gpurun_out/fuzz_r03b_compat.txt
cat gpurun_out/fuzz_r03b_newton.txt gpurun_out/fuzz_r03b_newton_lane.txt gpurun_out/fuzz_r03b_compat.txt
