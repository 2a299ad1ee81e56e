#!/bin/bash
# Build variants of the library that differ in the number of T columns the lane kernel keeps in LDS (PNP_LANE_TL9 / TL8 / TL7: blocks of
# 9 / 8 / 7) into catint_amd/lib/variants/lib<NAME>.so.  Runs HERE (cross-compiles); the variants travel with the next gpurun push --
# delete catint_amd/lib/variants/ when the A/B is done.
# usage: bash tools/probe/lane_tl_variants.sh NAME "-DPNP_LANE_TL9=6" [NAME "-D..." ...]
R=/root/repo/catint_amd; mkdir -p $R/lib/variants /tmp/tlv
while [ $# -ge 2 ]; do
  NAME=$1; DEFS=$2; shift; shift
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $DEFS -c $R/csrc/pnp_lane.hip -o /tmp/tlv/pnp_lane_$NAME.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $R/lib/obj/*.o | grep -v "/pnp_lane.o") /tmp/tlv/pnp_lane_$NAME.o -o $R/lib/variants/lib$NAME.so && echo "built $NAME ($DEFS)" ) &
done
wait
