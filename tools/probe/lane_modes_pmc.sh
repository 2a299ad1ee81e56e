#!/bin/bash
# on the GPU box: per-dispatch counters of the lane kernel over several workspace placements in ONE process (tools/probe/lane_modes2.py):
# do the slow placements miss the address-translation caches more often?
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; O=$R/gpurun_out/lane_modes_pmc; rm -rf $O; mkdir -p $O
rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum \
  --output-format csv -d $O/g1 -- python3 $R/tools/probe/lane_modes2.py ${1:-7} > $O/g1.log 2>&1 || tail -n 5 $O/g1.log
python3 - $O <<'PY'
import csv, collections, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/g1/*/*_counter_collection.csv')):
    rows = [r for r in csv.DictReader(open(f)) if 'newton_lane_kernel' in r['Kernel_Name']]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
    for d, c in by.items():
        act = c.get('GRBM_GUI_ACTIVE', 0)
        print(d, {k: ('%.4g' % v) for k, v in c.items()}, 'miss/active %.4g' % (c.get('TCP_UTCL1_TRANSLATION_MISS_sum', 0) / max(act, 1)))
PY
tail -n 2 $O/g1.log
