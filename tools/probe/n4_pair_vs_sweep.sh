for extra in "" "--mpb --stern" "--mpb --stern --reactions"; do
 for B in 16384 32768; do
  for k in default sweep; do
    if [ $k = sweep ]; then export CATINT_NEWTON_KERNEL=sweep; else unset CATINT_NEWTON_KERNEL; fi
    timeout -k 10 300 python tools/newton_bench.py --nspecies 4 --nx 512 --batch $B --steps 6 --warmup 1 $extra > /tmp/o.json 2>/tmp/o.err && python -c "
import json; d=json.loads(open('/tmp/o.json').read()); print('N=4 nx=512 B=$B $k [$extra] its/s %.3g ok %d' % (d['newton_iterations_per_s'], d['lanes_ok']))" || tail -2 /tmp/o.err
    if grep -q HSA_STATUS /tmp/o.err; then exit 1; fi
  done
 done
done
