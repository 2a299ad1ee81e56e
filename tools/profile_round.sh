#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace stats + HBM traffic counters (separate --pmc passes, as the
# MI355X guide prescribes) for the default bench.py command.  Results -> gpurun_out/profile_$1/
tag=${1:-r01}; shift
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/profile_$tag; mkdir -p $O
ARGS="--no-cpu-baseline --no-fused --large-batch 0 --physical-steps 0 --steps 256 --warmup 256 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py $ARGS > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $ARGS > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq -- python3 $R/bench.py $ARGS > $O/sq.log 2>&1
python3 $R/tools/pmc_traffic.py $O "$ARGS" > $O/summary.txt
cat $O/summary.txt
# physical mode (implicit Newton kernel): kernel-trace stats + SQ counters of tools/newton_bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d $O/newton_kt -- python3 $R/tools/newton_bench.py --steps 40 > $O/newton_kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/newton_sq -- python3 $R/tools/newton_bench.py --steps 40 > $O/newton_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/newton_fetch -- python3 $R/tools/newton_bench.py --steps 40 > $O/newton_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/newton_write -- python3 $R/tools/newton_bench.py --steps 40 > $O/newton_write.log 2>&1
find $O -name "*kernel_stats.csv" | head; tail -2 $O/newton_kt.log
python3 $R/tools/pmc_newton.py $O > $O/newton_summary.txt; cat $O/newton_summary.txt
