#!/bin/bash
# round-4 validation pass (not a campaign): the lane kernels incl. the lane quad with reactions / convection / masks, GPU box
set -o pipefail
for k in lane4 lane2 lane ""; do for s in 21 22; do echo "kernel=${k:-auto} seed=$s: $(CATINT_NEWTON_KERNEL=$k FUZZ_NMIN=$([ "$k" = lane4 -o "$k" = lane2 ] && echo 5 || echo 1) FUZZ_SEED=$s FUZZ_CASES=100 timeout -k 10 300 python tests/fuzz/fuzz_newton.py 2>&1 | tail -1)"; done; done
for s in 31 32 33; do echo "batches seed=$s: $(FUZZ_SEED=$s FUZZ_CASES=60 timeout -k 10 400 python tests/fuzz/fuzz_lane_batches.py 2>&1 | grep -E 'BAD|EXC|cases' | tail -4)"; done
