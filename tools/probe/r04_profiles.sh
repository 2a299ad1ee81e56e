#!/bin/bash
# Round-4 profile set (GPU box).  usage: bash tools/probe/r04_profiles.sh part1|part2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04prof; mkdir -p $O; export TMPDIR=/tmp
if [ "$1" = part1 ]; then
  cd $R
  python bench.py --steps 20 --warmup 5 > $O/r04_bench_line_steps20$RUN.json 2> $O/steps20.err; echo "steps20 rc $?"
  [ -n "$RUN" ] && exit 0      # (RUN=_run2: only the driver's command again -- another box of the pool)
  python bench.py > $O/r04_bench_line_default.json 2> $O/default.err; echo "default rc $?"
  CATINT_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $O/r04_bench_line_gpus2_gloo_rehearsal.json 2> $O/gpus2.err; echo "gpus2 rc $?"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-pmc > $O/kt.log 2>&1; echo "kt rc $?"
  cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats_steps20.csv
else
  cd /tmp
  # the beyond-cache per-step launch ALONE: one GPU's share of configs[3], one launch per timestep (single row chunk)
  CATINT_PNP_STEP_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bc -- python3 $R/tools/probe/stream_run.py 6 1024 32768 default 1 12 > $O/bc.log 2>&1; echo "bc rc $?"
  (tail -1 $O/bc.log; cat $(find $O/bc -name "*kernel_stats.csv" | head -1)) > $O/r04_rocprofv3_beyond_cache_per_step_alone.txt
  cd $R
  CATINT_NEWTON_KERNEL=lane4 bash tools/probe/lane_pmc.sh r04l4 8 512 8192 > $O/r04_rocprofv3_lane4_kernel_n8_nx512_b8192.txt 2>&1
  CATINT_NEWTON_KERNEL=lane4 bash tools/probe/lane_pmc.sh r04l4c4 8 4096 8192 > $O/r04_rocprofv3_lane4_kernel_config4_share.txt 2>&1
  CATINT_NEWTON_KERNEL=lane bash tools/probe/lane_pmc.sh r04l32k 8 512 32768 > $O/r04_rocprofv3_lane_kernel_fused_n8_nx512_b32768.txt 2>&1
  python tools/probe/lane4_probe.py --out $O/r04_lane4_probe.jsonl "8 512 1024" "8 512 2048" "8 512 4096" "8 512 8192" "8 512 12288" "8 512 16384" "6 1024 2048" "6 1024 8192" "6 1024 16384" "8 4096 2048" "8 4096 8192" > /dev/null 2>&1; echo "probe rc $?"
fi
