#!/bin/bash
# usage: tools/wsweep.sh "W,G W,G ..." [bench args]
cfgs=$1; shift
for c in $cfgs; do
w=${c%,*}; g=${c#*,}
CATINT_PNP_WAVES_PER_GRID=$w CATINT_PNP_SPECIES_PER_WAVE=$g python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('W=$w G=$g value %.4g launch_us %.2f frac %.3f per-step-launch %.4g (%.2f us/step) ok %d'%(d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d['per_step_launch']['timesteps_per_s'], d['per_step_launch']['ms_per_step']*1e3, d['lanes_ok']))"
done
