#!/bin/bash
# lane-quad kernel: what the forward row costs without one of its parts (elimination builds; results are wrong, cycles per row are not)
for X in "" "-DL4_NO_GJ" "-DL4_CHEAP_EDGE" "-DL4_NO_GJ -DL4_CHEAP_EDGE" "-DL4_NO_REC_STORE"; do
  echo "== $X"; EXTRA="$X" bash tools/probe/lane4_stamps.sh ${1:-8} ${2:-512} ${3:-8192} 2>&1 | tail -n 2 | head -n 1 | cut -c1-260
done
