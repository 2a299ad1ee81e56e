#!/bin/bash
# run on the GPU box: parity for every waves-per-grid variant, then the bench (JSON to gpurun_out/$1.json)
set -o pipefail
tag=${1:-bench}
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
for w in 1 2 3 4; do CATINT_PNP_WAVES_PER_GRID=$w timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1 || exit 1; done
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/$tag.json 2> gpurun_out/$tag.err || { tail -5 gpurun_out/$tag.err; exit 1; }
python - "$tag" <<'PY'
import json,sys
d=json.load(open('gpurun_out/%s.json'%sys.argv[1]))
print('value %.4g steps/s  launch_us %.2f  frac %.3f  fused %.4g steps/s (%.2f us/step)  lanes_ok %d' % (d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d['fused']['timesteps_per_s'], d['fused']['ms_per_step']*1e3, d['lanes_ok']))
PY
