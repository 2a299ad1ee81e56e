"""dev probe: where and when do the workgroups of step_kernel4 run (diagnostic build)"""
import ctypes as C, os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from catint_amd import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libcatint_pnp_diag.so')
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem
W = int(sys.argv[1]); nfuse = int(sys.argv[2]); B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
os.environ['CATINT_PNP_KERNEL'] = '4'; os.environ['CATINT_PNP_WAVES_PER_GRID'] = str(W)
p, c0, pb, vz, fl = make_batch(B, 3, 512, seed=1, phi_max=0.025, dt_factor=1e-4)
s = solver_from_problem(p, 'Crank-Nicolson', batch_capacity=B)
s.set_batch(c0, pb, vz, fl); s.step(nfuse, nfuse)
s.set_batch(c0, pb, vz, fl); s.step(nfuse, nfuse)
n = 4096 + 4 * 65536
buf = (C.c_ulonglong * n)()
_capi.load_library().pnp_debug_dump(s._h, buf)
t = np.array(buf, dtype=np.uint64)[4096:4096 + 4 * B].reshape(B, 4).astype(np.int64)
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0; end = (t[:, 1] - t0) / 100.0     # us (100 MHz)
hw = t[:, 2] & 0xffffffff; xcc = t[:, 2] >> 32
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
key = [(int(x), int(e), int(h), int(c)) for x, e, h, c in zip(xcc, se, sh, cu)]
cnt = collections.Counter(key)
print('W=%d steps=%d B=%d: kernel span %.1f us; WG duration min/median/max %.1f/%.1f/%.1f us; late starters (>1us): %d'
      % (W, nfuse, B, end.max(), (end - start).min(), np.median(end - start), (end - start).max(), (start > 1.0).sum()))
print('distinct CUs used: %d; WGs per CU histogram: %s' % (len(cnt), sorted(collections.Counter(cnt.values()).items())))
print('per-step time of WGs: median %.2f us, p90 %.2f us' % (np.median(end - start) / nfuse, np.percentile(end - start, 90) / nfuse))
