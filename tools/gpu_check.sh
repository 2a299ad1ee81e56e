#!/bin/bash
# run on the GPU box: parity for every kernel variant, then the bench (JSON to gpurun_out/$1.json)
set -o pipefail
tag=${1:-bench}
timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
for w in 1 2 3 4; do
  CATINT_PNP_KERNEL=4 CATINT_PNP_WAVES_PER_GRID=$w timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -1 || { echo "FAILED for step_kernel_rr W=$w"; exit 1; }
done
for c in 1,1 1,2 1,3 2,1 2,2 3,1 4,1; do
  w=${c%,*}; g=${c#*,}
  CATINT_PNP_KERNEL=2 CATINT_PNP_WAVES_PER_GRID=$w CATINT_PNP_SPECIES_PER_WAVE=$g timeout -k 10 400 python -m pytest tests/test_gpu_parity_golden.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -1 || { echo "FAILED for step_kernel W=$w G=$g"; exit 1; }
done
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/$tag.json 2> gpurun_out/$tag.err || { tail -5 gpurun_out/$tag.err; exit 1; }
python - "$tag" <<'PY'
import json,sys
d=json.load(open('gpurun_out/%s.json'%sys.argv[1]))
print('value %.4g steps/s  %.2f us/step  frac %.3f | per-step-launch %.4g steps/s (%.2f us) | large batch %.4g steps/s frac %.3f | lanes_ok %d' % (d['value'], d['ms_per_step']*1e3, d['roofline']['frac'], d['per_step_launch']['timesteps_per_s'], d['per_step_launch']['ms_per_step']*1e3, d['large_batch']['timesteps_per_s'], d['large_batch']['frac'], d['lanes_ok']))
PY
