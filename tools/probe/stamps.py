"""dev probe: per-phase s_memtime stamps of step_kernel3 (diagnostic build, never shipped)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from catint_amd import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libcatint_pnp_diag.so')
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem
G = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nfuse = int(sys.argv[2]) if len(sys.argv) > 2 else 8
os.environ['CATINT_PNP_KERNEL'] = '3'; os.environ['CATINT_PNP_SPECIES_PER_WAVE'] = str(G)
p, c0, pb, vz, fl = make_batch(1024, 3, 512, seed=1, phi_max=0.025, dt_factor=1e-4)
s = solver_from_problem(p, 'Crank-Nicolson', batch_capacity=1024)
s.set_batch(c0, pb, vz, fl)
s.step(nfuse, nfuse)
s.set_batch(c0, pb, vz, fl)
s.step(nfuse, nfuse)
buf = (C.c_ulonglong * (16 * 256))()
_capi.load_library().pnp_debug_dump(s._h, buf)
t = np.array(buf, dtype=np.uint64).reshape(256, 16).astype(np.int64)
names = ['loads+poisson', 'boundary values', 'assembly', 'tridiag', 'results/stores', 'charge row']
for st in range(nfuse):
    r = t[st]
    d = np.diff(r[:7])
    nxt = t[st + 1][0] - r[6] if st + 1 < nfuse else 0
    print('step %d: total %6d cycles | ' % (st, r[6] - r[0]) + ' '.join('%s %d' % (n, x) for n, x in zip(names, d)) + ' | to next %d' % nxt)
