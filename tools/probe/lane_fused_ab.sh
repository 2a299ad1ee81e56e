# lane kernel, update fused into the back-substitution or not, 20-step launches (the bench's workload), one device, one call
for spec in "8 512 8192 20" "8 512 12288 20" "8 512 14336 20" "8 512 16384 20" "8 512 20480 20" "6 1024 16384 10" "6 1024 24576 10" "7 512 16384 20" "5 512 16384 20"; do
  for f in 0 1 0 1; do
    echo -n "fused=$f "; CATINT_LANE_FUSED=$f timeout -k 10 120 python tools/probe/lane_rate.py lane "$spec" 2>/dev/null | cut -c1-200
  done
done
echo "lane2 for comparison"
for spec in "8 512 12288 20" "8 512 14336 20" "8 512 16384 20"; do timeout -k 10 120 python tools/probe/lane_rate.py lane2 "$spec" 2>/dev/null | cut -c1-200; done
