#!/bin/bash
# usage: tools/ksweep.sh "K:W,G K:W,G ..." [bench args]   (K = kernel 2|3)
cfgs=$1; shift
for c in $cfgs; do
k=${c%%:*}; wg=${c#*:}; w=${wg%,*}; g=${wg#*,}
CATINT_PNP_KERNEL=$k CATINT_PNP_WAVES_PER_GRID=$w CATINT_PNP_SPECIES_PER_WAVE=$g python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('K=$k W=$w G=$g value %.4g launch_us %.2f frac %.3f per-step-launch %.4g (%.2f us/step) ok %d'%(d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d['per_step_launch']['timesteps_per_s'], d['per_step_launch']['ms_per_step']*1e3, d['lanes_ok']))"
done
