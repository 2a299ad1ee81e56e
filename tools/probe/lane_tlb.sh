#!/bin/bash
# on the GPU box: is the run-to-run bimodality of the lane kernel at B = 32768 a matter of address translation?  Six processes, each under
# rocprofv3 --pmc (UTCL1 translation misses / requests) + --kernel-trace: kernel duration against miss counts.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/lane_tlb; mkdir -p $O
export CATINT_NEWTON_KERNEL=lane
rocprofv3 --list-avail 2>/dev/null | grep -o "TCP_UTCL1_[A-Z_0-9]*\|UTCL2[A-Z_0-9]*\|TCP_UTCL[A-Z_0-9]*" | sort -u > $O/names.txt
for i in 1 2 3 4 5 6; do
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST --kernel-trace --output-format csv -d $O/r$i -- python3 $R/tools/newton_bench.py --nspecies 8 --nx 512 --batch 32768 --steps 6 --warmup 1 --stern --mpb > $O/r$i.log 2>&1 || echo "run $i failed: $(tail -2 $O/r$i.log)"
done
python3 - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
d = sys.argv[1]
for i in range(1, 7):
    dur = {}
    for f in glob.glob('%s/r%d/*/*_kernel_trace.csv' % (d, i)):
        for r in csv.DictReader(open(f)):
            if 'newton_lane_kernel' in r['Kernel_Name']:
                dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    cnt = collections.defaultdict(dict)
    for f in glob.glob('%s/r%d/*/*_counter_collection.csv' % (d, i)):
        for r in csv.DictReader(open(f)):
            if 'newton_lane_kernel' in r['Kernel_Name']:
                cnt[r['Dispatch_Id']][r['Counter_Name']] = cnt[r['Dispatch_Id']].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    for k in sorted(dur, key=int):
        print('run', i, 'dispatch', k, 'ms %.1f' % dur[k], {n: '%.4g' % v for n, v in cnt.get(k, {}).items()})
    print(open('%s/r%d.log' % (d, i)).read().strip().splitlines()[-1][:160])
PY
cat $O/names.txt | tr "\n" " "
