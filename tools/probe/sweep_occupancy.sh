export CATINT_NEWTON_KERNEL=sweep
for cfg in "8 4096 8192 2" "6 1024 32768 3" "8 512 65536 4" "6 512 16384 4"; do set -- $cfg
  timeout -k 10 400 python tools/newton_bench.py --nspecies $1 --nx $2 --batch $3 --steps $4 --warmup 1 --mpb --stern > /tmp/o.json 2>/tmp/o.err && python -c "
import json; d=json.loads(open('/tmp/o.json').read()); print('N=$1 nx=$2 B=$3 sweep(64,2): its/s %.3g ok %d' % (d['newton_iterations_per_s'], d['lanes_ok']))" || tail -3 /tmp/o.err
  if grep -q HSA_STATUS /tmp/o.err; then exit 1; fi
done
