#!/bin/bash
# on the GPU box: SQ / TCC counters of the lane kernel (each group its own rocprofv3 pass, no tracing beside --pmc)
# usage: bash tools/probe/lane_pmc.sh TAG NSPECIES NX BATCH [extra newton_bench args]
tag=${1:-x}; N=${2:-8}; NX=${3:-512}; B=${4:-8192}; shift 4
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
export CATINT_NEWTON_KERNEL=${CATINT_NEWTON_KERNEL-lane}
mkdir -p $R/gpurun_out/lanepmc_$tag
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/lanepmc_$tag/g$i -- python3 $R/tools/newton_bench.py --nspecies $N --nx $NX --batch $B --steps 4 --warmup 1 --stern --mpb "$@" > $R/gpurun_out/lanepmc_$tag/g$i.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lanepmc_$tag/kt -- python3 $R/tools/newton_bench.py --nspecies $N --nx $NX --batch $B --steps 4 --warmup 1 --stern --mpb "$@" > $R/gpurun_out/lanepmc_$tag/kt.log 2>&1 || exit 1
python3 - $R/gpurun_out/lanepmc_$tag <<'PY' | tee $R/gpurun_out/lanepmc_$tag/summary.txt
import csv,collections,glob,sys
d=sys.argv[1]
print(open(d+'/kt.log').read().strip().splitlines()[-1])
for f in glob.glob(d+'/kt/*/*_kernel_stats.csv'):
    print(open(f).read().strip())
for f in sorted(glob.glob(d+'/g*/*/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list); meta=None
    for r in rows:
        if 'newton_' in r['Kernel_Name'] and 'transpose' not in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            waves=int(r['Grid_Size'])//64
            meta={k: r[k] for k in ('Kernel_Name','Grid_Size','Workgroup_Size','VGPR_Count','Accum_VGPR_Count','SGPR_Count','LDS_Block_Size','Scratch_Size')}
    print(meta)
    for k,v in agg.items():
        v=v[-1:]     # the timed launch (last)
        print('%-24s last=%.5g  per-wave=%.1f'%(k,v[0], v[0]/waves))
PY
