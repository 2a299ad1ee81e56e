#!/bin/bash
# dev tool: compile only one block size of pnp_newton.hip (default N+1 = 4) to /tmp/nw and print register / scratch /
# instruction-mix statistics of its kernels.   usage: tools/nwexp.sh [NB]
NB=${1:-4}
mkdir -p /tmp/nw && cd /tmp/nw || exit 1
sed -E "/^    case [0-9]: return launch_newton_nb/{/case ${NB}:/!d}" /root/repo/catint_amd/csrc/pnp_newton.hip > nw_only.hip
sed -i "s|#include \"pnp_internal.h\"|#include \"/root/repo/catint_amd/csrc/pnp_internal.h\"|; s|#include \"pnp_math.h\"|#include \"/root/repo/catint_amd/csrc/pnp_math.h\"|" nw_only.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --save-temps -Wall -Wno-unused-function nw_only.hip -o nw_only.o 2>&1 | grep -v "^$" | head -20
S=nw_only-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "\.vgpr_count|\.private_segment_fixed_size|\.name:|\.sgpr_spill_count|\.vgpr_spill_count" $S | paste - - - - - | awk '{print $2,$4,$6,$8,$10}'
for k in $(grep -oE "^_ZN3pnp18newton_pair_kernel\w+" $S | sort -u); do python /root/repo/tools/asm_mix.py $S ${k#_ZN3pnp} 2>/dev/null | head -14; done
