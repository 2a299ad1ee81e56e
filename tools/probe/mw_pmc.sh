#!/bin/bash
# on the GPU box: SQ counters of step_kernel_mw (one rocprofv3 --pmc pass per group) on 2048 x 8 x 4096, one launch per timestep
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/mw_pmc; mkdir -p $O
cat > $O/run.py <<PY
import sys; sys.path.insert(0, '$R')
import bench
s, inp = bench.compat_solver(2048, 8, 4096, 'Crank-Nicolson', 5)
s.set_batch(*inp[1:])
s.step(6, 1)
s.synchronize()
s.close()
PY
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVES_EQ_64"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 $O/run.py > $O/g$i.log 2>&1 || echo "group $i failed: $(tail -2 $O/g$i.log)"
done
python3 - $O <<'PY' | tee $O/summary.txt
import csv,collections,glob,sys
d=sys.argv[1]
for f in sorted(glob.glob(d+'/g*/*/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'step_kernel_mw' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        print(k, 'launches', len(v), 'median %.4g' % sorted(v)[len(v)//2])
PY
