#!/bin/bash
# like ab_lib.sh, REPS alternations, one summary line per library and shape (median of the per-process medians)
K=$1; REPS=$2; shift; shift
for spec in "$@"; do
  for rep in $(seq $REPS); do
    CATINT_PNP_LIB=$PWD/catint_amd/lib/variants/libA.so timeout -k 10 200 python tools/probe/lane_rate.py $K "$spec" 2>/dev/null >> /tmp/ab_A.jsonl
    timeout -k 10 200 python tools/probe/lane_rate.py $K "$spec" 2>/dev/null >> /tmp/ab_new.jsonl
  done
  python - "$spec" <<'PY'
import json, sys, statistics
for tag in ('A', 'new'):
    rows = [json.loads(l) for l in open('/tmp/ab_%s.jsonl' % tag)]
    med = [statistics.median(r['timesteps_per_s']) for r in rows]
    print(sys.argv[1], tag, 'per process:', [round(m / 1e6, 3) for m in med], 'median', round(statistics.median(med) / 1e6, 3))
PY
  rm -f /tmp/ab_A.jsonl /tmp/ab_new.jsonl
done
