#!/bin/bash
# on the GPU box: instruction-cache and instruction-fetch counters of the lane / lane-pair kernels (one rocprofv3 --pmc pass per group)
# usage: bash tools/probe/lane_icache.sh TAG KERNEL NSPECIES NX BATCH
tag=${1:-x}; K=${2:-lane}; N=${3:-8}; NX=${4:-512}; B=${5:-8192}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
export CATINT_NEWTON_KERNEL=$K
O=$R/gpurun_out/icache_$tag; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*" $O/avail.txt | sort -u > $O/names.txt
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU" "SQ_IFETCH_LEVEL SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/newton_bench.py --nspecies $N --nx $NX --batch $B --steps 4 --warmup 1 --stern --mpb > $O/g$i.log 2>&1 || echo "group $i failed: $(tail -2 $O/g$i.log)"
done
python3 - $O <<'PY' | tee $O/summary.txt
import csv,collections,glob,sys
d=sys.argv[1]
for f in sorted(glob.glob(d+'/g*/*/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'newton_' in r['Kernel_Name'] and 'transpose' not in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        print(k[0], k[1], 'launches', len(v), 'last %.4g' % v[-1])
PY
