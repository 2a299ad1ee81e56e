#!/bin/bash
# on the GPU box: PMC groups (each its own pass) for one streaming-probe configuration -> gpurun_out/spmc_<tag>/summary.txt
# usage: tools/probe/stream_pmc.sh tag N nx B variant spl nlaunch
tag=$1; shift
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/spmc_$tag; mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_WRITE_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/probe/stream_run.py "$@" > $O/g$i.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/probe/stream_run.py "$@" > $O/kt.log 2>&1
python3 - $O "$@" > $O/summary.txt <<'PY'
import csv,collections,glob,sys
O=sys.argv[1]
print('# stream_run.py '+' '.join(sys.argv[2:]))
for f in sorted(glob.glob(O+'/g*/*/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list); meta=None
    for r in rows:
        if 'step_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            meta=r
    if meta is None: continue
    waves=int(meta['Grid_Size'])//64
    for k,v in agg.items():
        v=sorted(v); med=v[len(v)//2]
        print('%-30s median=%.6g  per-wave=%.1f   (n=%d, grid %s wg %s vgpr %s scratch %s lds %s) %s'%(k,med,med/waves,len(v),meta['Grid_Size'],meta['Workgroup_Size'],meta['VGPR_Count'],meta['Scratch_Size'],meta['LDS_Block_Size'],meta['Kernel_Name'][:40]))
for f in glob.glob(O+'/kt/*/*_kernel_stats.csv'):
    print(open(f).read().strip())
for f in sorted(glob.glob(O+'/g*.log')):
    t=open(f).read().strip().splitlines()
    print(f.split('/')[-1], t[-1] if t else '')
PY
cat $O/summary.txt
