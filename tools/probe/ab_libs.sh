#!/bin/bash
# A/B/C... of several builds of the library on one device in one call: every catint_amd/lib/variants/lib*.so and the library in place
# ("cur"), REPS alternations (one process each), median of the per-process medians.  DELETE catint_amd/lib/variants/ afterwards.
# usage: bash tools/probe/ab_libs.sh KERNEL REPS "N NX B STEPS" ...
K=$1; REPS=$2; shift; shift
for spec in "$@"; do
  rm -f /tmp/ab_*.jsonl
  for rep in $(seq $REPS); do
    for lib in catint_amd/lib/variants/lib*.so; do
      n=$(basename $lib .so); CATINT_PNP_LIB=$PWD/$lib timeout -k 10 200 python tools/probe/lane_rate.py $K "$spec" 2>/dev/null >> /tmp/ab_$n.jsonl
    done
    timeout -k 10 200 python tools/probe/lane_rate.py $K "$spec" 2>/dev/null >> /tmp/ab_cur.jsonl
  done
  python - "$spec" <<'PY'
import glob, json, sys, statistics
for f in sorted(glob.glob('/tmp/ab_*.jsonl')):
    rows = [json.loads(l) for l in open(f)]
    med = [statistics.median(r['timesteps_per_s']) for r in rows]
    print(sys.argv[1], f[8:-6].ljust(8), 'per process:', [round(m / 1e6, 3) for m in med], 'median', round(statistics.median(med) / 1e6, 3))
PY
done
