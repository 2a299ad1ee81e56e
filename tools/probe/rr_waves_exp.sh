for cfg in "0 0" "4 1" "4 3" "2 3"; do set -- $cfg
  CATINT_PNP_KERNEL=$1 CATINT_PNP_WAVES_PER_GRID=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --large-batch 0 --physical-steps 0 --steps 1024 --warmup 256 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel=$1 waves=$2', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
