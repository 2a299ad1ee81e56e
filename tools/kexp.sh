#!/bin/bash
# dev tool: compile ONE kernel instantiation and print resource usage + instruction mix
# usage: tools/kexp.sh "step_kernel_rr<8,3,true>" [extra hipcc flags]
INST=$1; shift
D=/tmp/asm; mkdir -p $D
SRC=/root/repo/catint_amd/csrc
python3 - "$INST" <<PY
import sys
inst=sys.argv[1]
s=open('$SRC/pnp_kernels.hip').read()
cut=s.index('// host-side launchers')
s=s[:cut]+"\n*/\ntemplate __global__ void %s(const DevArgs);\n}\n"%inst
s=s.replace('// ------------------------------------------------------------------------------------------------\n\n*/','/*\n*/')
s=s.replace('#include "pnp_internal.h"','#include "$SRC/pnp_internal.h"')
open('$D/kexp.hip','w').write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" -o $D/kexp.s $D/kexp.hip 2>&1 | grep -E "error|step_kernel|VGPRs:|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize|Occupancy" | sed 's/.*remark: //; s/\[-Rpass.*//' | grep -A6 "step_kernel" | paste -sd' '
python3 /root/repo/tools/asm_mix.py $D/kexp.s step_kernel | head -16
