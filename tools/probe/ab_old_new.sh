#!/bin/bash
# A/B of two trees on one device in one call: _ab_old (a copy of an older commit with its own library) against the current tree.
# (_ab_old: `git worktree add /tmp/old COMMIT`, build its library there, copy catint_amd/ tools/ bench.py include/ into _ab_old/; delete it
#  afterwards -- it is git-ignored but would travel with every gpurun push)
# usage: bash tools/probe/ab_old_new.sh KERNEL "N NX B STEPS" ...
K=$1; shift
for spec in "$@"; do
  for rep in 1 2; do
    echo -n "old "; (cd _ab_old && timeout -k 10 200 python tools/probe/lane_rate.py $K "$spec" 2>/dev/null | cut -c1-190)
    echo -n "new "; timeout -k 10 200 python tools/probe/lane_rate.py $K "$spec" 2>/dev/null | cut -c1-190
  done
done
