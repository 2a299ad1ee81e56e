#!/bin/bash
# On the GPU box: the driver's bench command, the default one, and the rocprofv3 --kernel-trace --stats pass of the driver's command
# (no nested counter collection under the profiler: --no-pmc).  Results -> gpurun_out/bench_round/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bench_round; mkdir -p $O
cd $R
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err || exit 1
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
export TMPDIR=/tmp; cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-pmc > $O/kt.log 2>&1 || exit 1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_steps20.csv; head -12 $O/kernel_stats_steps20.csv
