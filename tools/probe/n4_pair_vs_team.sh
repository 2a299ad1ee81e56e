for extra in "" "--mpb" "--mpb --reactions --stern"; do
 for B in 1024 8192; do
  for k in pair team; do
    if [ $k = team ]; then export CATINT_NEWTON_KERNEL=team; else unset CATINT_NEWTON_KERNEL; fi
    python tools/newton_bench.py --nspecies 4 --nx 512 --batch $B --steps 20 $extra | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('N=4 nx=512 B=$B $k [$extra] its/s %.3g  ok %d' % (d['newton_iterations_per_s'], d['lanes_ok']))"
  done
 done
done
unset CATINT_NEWTON_KERNEL
for k in pair team; do
    if [ $k = team ]; then export CATINT_NEWTON_KERNEL=team; else unset CATINT_NEWTON_KERNEL; fi
    python tools/newton_bench.py --nspecies 4 --nx 256 --batch 4096 --steps 20 --mpb | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('N=4 nx=256 B=4096 $k mpb its/s %.3g  ok %d' % (d['newton_iterations_per_s'], d['lanes_ok']))"
    python tools/newton_bench.py --nspecies 4 --nx 1024 --batch 1024 --steps 10 --mpb | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('N=4 nx=1024 B=1024 $k mpb its/s %.3g  ok %d' % (d['newton_iterations_per_s'], d['lanes_ok']))"
done
