"""ctypes wrapper of the C oracle (oracle/pnp_oracle.c).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by catint_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, '_build', 'libpnp_oracle.so')
MAX_REACTANTS = 4
_lib = None


def build():
    subprocess.run(['make', '-s', '-C', _HERE], check=True)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(_HERE, 'pnp_oracle.c')):
            build()
        lib = C.CDLL(LIB)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        lib.pnp_oracle_steps.argtypes = [C.c_int32] * 6 + [C.c_double] * 4 + [dp, dp, C.c_int32, ip, ip, ip, ip, dp, dp,
                                                                             C.c_int64, dp, dp, dp, dp, dp, C.c_int32,
                                                                             C.c_int32, dp, dp, dp, C.c_int32]
        lib.pnp_oracle_steps.restype = C.c_int
        lib.pnp_oracle_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def max_threads():
    return int(load().pnp_oracle_max_threads())


def steps(p, method, c, pb, vzeta, flux, nsteps, first_pass=True, cbulk=None, want_potential=True, nthreads=0):
    """Advance B lanes by nsteps loop passes. p: oracle.pnp_ref.Problem-like (problem-wide fields).
    c [B][N][nx] is advanced IN PLACE (must be C-contiguous float64). Returns (v, grad_v, lapl_v) or None."""
    from .pnp_ref import pb_mode_from_bound
    lib = load()
    N, nx = len(p.D), p.nx
    assert c.dtype == np.float64 and c.flags['C_CONTIGUOUS']
    B = c.shape[0]
    c3 = c.reshape(B, N, nx)
    if cbulk is None:
        cbulk = c3[:, :, -1].copy()
    cbulk = np.ascontiguousarray(cbulk, dtype=np.float64)
    pb = np.ascontiguousarray(np.broadcast_to(np.asarray(pb, dtype=np.float64), (B, 4)))
    mode = pb_mode_from_bound(pb[0])
    pb = np.nan_to_num(pb, nan=0.0)
    vzeta = np.ascontiguousarray(np.broadcast_to(np.asarray(vzeta, dtype=np.float64), (B,)))
    flux = np.ascontiguousarray(np.broadcast_to(np.asarray(flux, dtype=np.float64), (B, N)))
    D = np.ascontiguousarray(p.D, dtype=np.float64)
    q = np.ascontiguousarray(p.charges, dtype=np.float64)
    reactions = list(getattr(p, 'reactions', []) or [])
    nr = len(reactions)
    n_lhs = np.zeros(max(nr, 1), np.int32); n_rhs = np.zeros(max(nr, 1), np.int32)
    lhs = np.zeros((max(nr, 1), MAX_REACTANTS), np.int32); rhs = np.zeros((max(nr, 1), MAX_REACTANTS), np.int32)
    kf = np.zeros(max(nr, 1)); kr = np.zeros(max(nr, 1))
    for r, (l, rr, f, b_) in enumerate(reactions):
        n_lhs[r], n_rhs[r] = len(l), len(rr)
        lhs[r, :len(l)] = l; rhs[r, :len(rr)] = rr
        kf[r], kr[r] = f, b_
    if want_potential:
        v = np.zeros((B, nx)); g = np.zeros((B, nx)); l = np.zeros((B, nx))
    else:
        v = g = l = None
    m = {'Crank-Nicolson': 0, 'FTCS': 1}[method]
    rc = lib.pnp_oracle_steps(N, nx, m, mode, int(bool(p.lax_friedrich)), int(bool(p.use_migration)), p.dx, p.dt, p.beta,
                              p.eps, _d(D), _d(q), nr, _i(n_lhs), _i(lhs), _i(n_rhs), _i(rhs), _d(kf), _d(kr), B,
                              _d(c3), _d(cbulk), _d(pb), _d(vzeta), _d(flux), int(nsteps), int(bool(first_pass)),
                              _d(v), _d(g), _d(l), int(nthreads))
    if rc != 0:
        raise RuntimeError('pnp_oracle_steps failed (%d)' % rc)
    return (v, g, l) if want_potential else None


def integrate(p, c0, nt, itout, method, nthreads=0):
    """integrate_pnp for B lanes: c0 [B][N*nx] -> cout [n_out][B][N*nx], (v, grad_v, lapl_v) of the last pass."""
    c0 = np.ascontiguousarray(c0, dtype=np.float64)
    B = c0.shape[0]
    N, nx = len(p.D), p.nx
    c = c0.reshape(B, N, nx).copy()
    cbulk = c[:, :, -1].copy()
    n = 1 if method == 'Crank-Nicolson' else 0
    first = True
    cout, pot = [], None
    for target in list(itout) + [nt - 1]:
        todo = target - n + 1
        if todo > 0:
            pot = steps(p, method, c, p.pb, p.vzeta, p.flux_bound, todo, first_pass=first, cbulk=cbulk, nthreads=nthreads)
            first = False
            n = target + 1
        if len(cout) < len(itout):
            cout.append(c.reshape(B, -1).copy())
    return np.array(cout), pot
