#!/bin/bash
# SQ counters of the sweep kernel on a large batch (rocprofv3 --pmc, own run)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/sweep_counters; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq -- python3 $R/tools/newton_bench.py --nspecies 8 --nx 512 --batch 32768 --steps 2 --warmup 1 --mpb --stern > $O/sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/newton_bench.py --nspecies 8 --nx 512 --batch 32768 --steps 2 --warmup 1 --mpb --stern > $O/kt.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('$O/sq/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'sweep' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
        meta = (r['Kernel_Name'], r['Grid_Size'], r['VGPR_Count'], r['SGPR_Count'], r['Scratch_Size'], r['LDS_Block_Size'])
print(meta)
w = agg['SQ_WAVES'][-1]
for k, v in sorted(agg.items()):
    print(k, v[-1] / w)
for f in glob.glob('$O/kt/*/*_kernel_stats.csv'):
    print(open(f).read()[:600])
PY
