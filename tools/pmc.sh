#!/bin/bash
# on the GPU box: collect PMC groups for the step kernel (each group its own pass) -> gpurun_out/pmc_$1/
tag=${1:-x}; shift
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$tag/g$i -- python3 $R/bench.py --no-cpu-baseline --no-fused --large-batch 0 --steps 64 --warmup 64 "$@" > $R/gpurun_out/pmc_$tag/g$i.log 2>&1
done
python3 - $R/gpurun_out/pmc_$tag <<'PY'
import csv,collections,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/g*/*/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    waves=None
    for r in rows:
        if 'step_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            wg=int(r['Workgroup_Size']); grid=int(r['Grid_Size']); waves=grid//64
    for k,v in agg.items():
        print('%-24s mean=%.4g  per-wave=%.1f'%(k,sum(v)/len(v), sum(v)/len(v)/waves))
PY
