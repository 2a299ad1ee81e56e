python - <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from catint_amd import _capi
from catint_amd.synthetic import make_batch
B = 1024
prob, c0, pb, vz, fl = make_batch(B, 3, 512, seed=0, dt_factor=1e-5)
s = _capi.PnpSolver(3, 512, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Crank-Nicolson', batch_capacity=B)
def burst(tag, n=5, spl=256):
    out = []
    for _ in range(n):
        s.timer_start(); s.step(spl, spl); out.append(s.timer_stop() / spl * 1e3)
    print('%-40s %s' % (tag, ' '.join('%.2f' % x for x in out)), flush=True)
for rep in range(10):
    s.set_batch(c0, pb, vz, fl)
    burst('after set_batch(c0)')
PY
python bench.py --no-cpu-baseline --physical-steps 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['large_batch'])"
python bench.py --no-cpu-baseline --physical-steps 0 --large-batch 0| python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'])"
