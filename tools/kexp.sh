#!/bin/bash
# dev tool: compile ONE instantiation of step_kernel and print resource usage + instruction mix
# usage: tools/kexp.sh P W [extra hipcc flags]
P=$1; W=$2; G=$3; shift 3
D=/tmp/asm; mkdir -p $D
SRC=/root/repo/catint_amd/csrc
python3 - "$P" "$W" "$G" <<PY
import sys
P,W,G=sys.argv[1],sys.argv[2],sys.argv[3]
s=open('$SRC/pnp_kernels.hip').read()
cut=s.index('// host-side launchers')
s=s[:cut]+"\n*/\ntemplate __global__ void step_kernel<%s,%s,%s>(const DevArgs);\n}\n"%(P,W,G)
# the cut lands inside a comment banner: reopen it
s=s.replace('// ------------------------------------------------------------------------------------------------\n\n*/','/*\n*/')
s=s.replace('#include "pnp_internal.h"','#include "$SRC/pnp_internal.h"')
open('$D/kexp.hip','w').write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" -o $D/kexp.s $D/kexp.hip 2>&1 | grep -E "error|VGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize|Occupancy" | sed 's/.*remark: //; s/\[-Rpass.*//' | paste -sd' '
python3 /root/repo/tools/asm_mix.py $D/kexp.s step_kernel | head -14
