#!/bin/bash
# on the GPU box: instruction-cache counters of a lane kernel (is the ~70 KB loop body served by the 64 KB instruction cache two CUs share?)
# usage: bash tools/probe/lane_icache_pmc.sh KERNEL NSPECIES NX BATCH
K=${1:-lane}; N=${2:-8}; NX=${3:-512}; B=${4:-32768}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
export CATINT_NEWTON_KERNEL=$K
O=$R/gpurun_out/icache_$K; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" | sort -u | tr '\n' ' ' > $O/avail.txt; cat $O/avail.txt; echo
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/newton_bench.py --nspecies $N --nx $NX --batch $B --steps 4 --warmup 1 --stern --mpb > $O/g$i.log 2>&1 || { tail -3 $O/g$i.log; continue; }
done
python3 - $O <<'PY'
import csv, collections, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/g*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'newton_lane' in r['Kernel_Name'] and 'transpose' not in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    print({k: (v, n[k]) for k, v in agg.items()})
PY
