"""dev probe: per-phase s_memtime stamps of step_kernel4 (diagnostic build, never shipped)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from catint_amd import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libcatint_pnp_diag.so')
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nfuse = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
os.environ['CATINT_PNP_KERNEL'] = '4'; os.environ['CATINT_PNP_WAVES_PER_GRID'] = str(W)
p, c0, pb, vz, fl = make_batch(B, 3, 512, seed=1, phi_max=0.025, dt_factor=1e-4)
s = solver_from_problem(p, 'Crank-Nicolson', batch_capacity=B)
s.set_batch(c0, pb, vz, fl)
s.step(nfuse, nfuse)
s.set_batch(c0, pb, vz, fl)
s.step(nfuse, nfuse)
buf = (C.c_ulonglong * (4096 + 4 * 65536))()
_capi.load_library().pnp_debug_dump(s._h, buf)
t = np.array(buf, dtype=np.uint64)[:4096].reshape(4, 64, 16).astype(np.int64)
names = ['loads+poisson', 'bc+assembly', 'tridiag', 'results', 'exchange', 'charge row']
for w in range(W):
    for st in range(2, nfuse):
        r = t[w, st]
        d = np.diff(r[:7])
        print('wave %d step %d: total %6d | ' % (w, st, r[6] - r[0]) + ' '.join('%s %d' % (n, x) for n, x in zip(names, d)))
# shader clock: delta s_memtime / delta s_memrealtime * 100 MHz (MI355X_MICROARCH.md, DVFS item 6)
r = t[0]
dt_cyc = r[nfuse - 1][0] - r[1][0]
dt_rt = r[nfuse - 1][8] - r[1][8]
if dt_rt > 0:
    print('in-kernel clock ~ %.2f GHz  (%d cycles over %d ticks of 100 MHz)' % (dt_cyc / dt_rt * 0.1, dt_cyc, dt_rt))
