#!/bin/bash
# A/B of two builds of the library on one device in one call: catint_amd/lib/variants/libA.so (CATINT_PNP_LIB) against the library in
# place.  DELETE catint_amd/lib/variants/ afterwards (it travels with every gpurun push).
# usage: bash tools/probe/ab_lib.sh KERNEL "N NX B STEPS" ...
K=$1; shift
for spec in "$@"; do
  for rep in 1 2; do
    echo -n "A   "; CATINT_PNP_LIB=$PWD/catint_amd/lib/variants/libA.so timeout -k 10 200 python tools/probe/lane_rate.py $K "$spec" 2>/dev/null | cut -c1-175
    echo -n "new "; timeout -k 10 200 python tools/probe/lane_rate.py $K "$spec" 2>/dev/null | cut -c1-175
  done
done
