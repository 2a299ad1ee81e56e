#!/bin/bash
# On the GPU box: SQ counters + kernel stats of the row-per-thread Newton kernel (N = 6, nx = 1024, batch 512).
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/profile_newton_generic; mkdir -p $O
CMD="python3 $R/tools/newton_bench.py --nspecies 6 --nx 1024 --batch 512 --steps 6 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $CMD > $O/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/sq -- $CMD > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq2 -- $CMD > $O/sq2.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + '/kt/*/*_kernel_stats.csv'):
    print(open(f).read().strip())
for d in ('sq', 'sq2'):
    fs = glob.glob('%s/%s/*/*_counter_collection.csv' % (out, d))
    if not fs:
        print(d, 'no counters:', open('%s/%s.log' % (out, d)).read()[-400:]); continue
    agg = collections.defaultdict(list); grid = None
    for r in csv.DictReader(open(fs[0])):
        if 'newton' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value'])); grid = float(r['Grid_Size']); meta = (r['Kernel_Name'], r['VGPR_Count'], r['Scratch_Size'], r['LDS_Block_Size'])
    print(meta)
    print('per wave (mean over launches): ' + ', '.join('%s=%.0f' % (k, sum(v) / len(v) / (grid / 64)) for k, v in sorted(agg.items())))
PY
