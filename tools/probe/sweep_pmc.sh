#!/bin/bash
# on the GPU box: SQ / LDS counters of the Newton sweep kernels (each group its own pass): sweep_pmc.sh N nx B kernel tag
N=$1; NX=$2; B=$3; K=$4; tag=$5
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
mkdir -p $R/gpurun_out/pmc_$tag
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$tag/g$i -- python3 $R/tools/probe/sweep_pmc_run.py $N $NX $B $K > $R/gpurun_out/pmc_$tag/g$i.log 2>&1 || exit 1
done
python3 - $R/gpurun_out/pmc_$tag <<'PY'
import csv,collections,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/g*/*/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    waves=None; name=None
    for r in rows:
        if 'newton' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            waves=int(r['Grid_Size'])//64; name=r['Kernel_Name'][:60]; regs=(r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'])
    for k,v in agg.items():
        print('%-24s last-launch=%.5g  per-wave=%.1f   (%s, %d waves, vgpr/agpr/lds/scratch %s)'%(k, v[-1], v[-1]/waves, name, waves, regs))
PY
