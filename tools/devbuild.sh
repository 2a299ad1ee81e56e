#!/bin/bash
# dev tool: build libcatint_pnp.so with only ONE block size of the Newton kernels instantiated (fast turn-around).
# usage: tools/devbuild.sh NB   (then run with CATINT_ALLOW_PARTIAL=1: the loader refuses a partial library otherwise)
#             (restore the full library afterwards with `python -c "import __graft_entry__ as g; g.build()"`)
NB=${1:-4}
R=/root/repo/catint_amd
mkdir -p /tmp/devb
sed -E "/^    case [0-9]: return launch_newton_nb/{/case ${NB}:/!d}" $R/csrc/pnp_newton.hip > /tmp/devb/pnp_newton.hip
sed -i "s|#include \"pnp_internal.h\"|#include \"$R/csrc/pnp_internal.h\"|; s|#include \"pnp_math.h\"|#include \"$R/csrc/pnp_math.h\"|" /tmp/devb/pnp_newton.hip
touch $R/lib/.partial   # build_library() rebuilds the full library when it sees this marker
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function $R/csrc/pnp_kernels.hip /tmp/devb/pnp_newton.hip $R/csrc/pnp_scf.hip $R/csrc/pnp_capi.hip -o $R/lib/libcatint_pnp.so 2>&1 | grep -v "^$" | head
