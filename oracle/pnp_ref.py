"""CPU ORACLE (test infrastructure, NOT product code) -- NumPy restatement of the reference's
legacy finite-difference PNP integrators, one operating point at a time.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product path (``catint_amd``) never does.

Pinned: every function below is checked against golden vectors produced by running the
reference itself (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) in
``tests/test_oracle_golden.py``.

Reference (sringe/CatINT, paths relative to /root/reference):
  catint/calculator_old.py:680-819   get_potential_and_gradient  -> :func:`poisson`
  catint/calculator_old.py:457-564   integrate_Crank_Nicolson    -> :func:`cn_step`, :func:`integrate`
  catint/calculator_old.py:976-1029  integrate_FTCS              -> :func:`ftcs_step`
  catint/calculator_old.py:827-935   ode_func (method of lines)  -> :func:`mol_rhs`
  catint/calculator_old.py:159-208   get_rates                   -> :func:`get_rates`
  catint/calculator_old.py:140-152   itout selection             -> :func:`make_itout`

The arithmetic deliberately reproduces the reference's quirks: the
row-vector x matrix RHS product, the interior-index offset of grad_v / lapl_v in the
Crank-Nicolson stencil, lagged potential, the FTCS migration sign, get_rates' overwrite.
``solver='dense'`` uses np.linalg.solve on (nx-2)^2 arrays exactly like the reference;
``solver='banded'`` uses a Thomas sweep (same equations, O(nx)).
"""
from dataclasses import dataclass, field
import numpy as np

# Poisson boundary modes (which two of pb_bound's four slots are set; transport.py:1296-1311)
PB_DD = 0            # potential wall + potential bulk        calculator_old.py:780-786
PB_VWALL_GBULK = 1   # potential wall + gradient bulk (default, transport.py:207-210)
PB_GWALL_VBULK = 2   # gradient wall + potential bulk
PB_VWALL_GWALL = 3   # potential wall + gradient wall
PB_VBULK_GBULK = 4   # potential bulk + gradient bulk


def pb_mode_from_bound(pb):
    """pb = [pot_wall, pot_bulk, grad_wall, grad_bulk] with NaN for 'None'."""
    vw, vb, gw, gb = [not np.isnan(x) for x in pb]
    if gw and gb:
        raise ValueError('Cannot use two boundary conditions for gradient')  # calculator_old.py:712-714
    if vw and vb:
        return PB_DD
    if vw and gb:
        return PB_VWALL_GBULK
    if gw and vb:
        return PB_GWALL_VBULK
    if vw and gw:
        return PB_VWALL_GWALL
    if vb and gb:
        return PB_VBULK_GBULK
    raise ValueError('unsupported pb_bound combination')


@dataclass
class Problem:
    """Everything the legacy integrators read from ``tp`` for one operating point."""
    D: np.ndarray            # [N]  m^2/s            transport.py:423-434
    charges: np.ndarray      # [N]  z*F  C/mol       transport.py:1271
    beta: float              # 1/(RT)                transport.py:312
    eps: float               # eps_r*eps_0           transport.py:311
    dx: float
    nx: int
    dt: float
    pb: np.ndarray           # [4] pot wall, pot bulk, grad wall, grad bulk (NaN = None)
    vzeta: float             # tp.system['vzeta']
    flux_bound: np.ndarray   # [N] left (wall) flux  tp.flux_bound[:,0]
    lax_friedrich: bool = False
    use_migration: bool = True
    reactions: list = field(default_factory=list)   # [(lhs_idx[], rhs_idx[], kf, kr)], see encode_reactions

    @property
    def N(self):
        return len(self.D)

    @property
    def mu(self):            # transport.py:436
        return self.D * self.charges * self.beta


def encode_reactions(reactions, species_names):
    """tp.reactions dict -> ordered list of (lhs species indices, rhs indices, kf, kr).
    Reactions without 'rates' are skipped, species not in the species list are dropped
    (calculator_old.py:166-171)."""
    out = []
    names = list(species_names)
    for r in reactions:
        rx = reactions[r]
        if 'rates' not in rx:
            continue
        lhs = [names.index(s) for s in rx['reactants'][0] if s in names]
        rhs = [names.index(s) for s in rx['reactants'][1] if s in names]
        out.append((lhs, rhs, float(rx['rates'][0]), float(rx['rates'][1])))
    return out


def make_itout(nt, ntout):
    """calculator_old.py:140-152 (== calculator.py:126-138)."""
    itout = []
    for it in range(nt):
        if it == nt - 1:
            itout.append(it)
        elif it > 1 and it % int(nt / float(ntout)) == 0:
            itout.append(it)
    return itout


def _thomas(a, b, c, d):
    """Tridiagonal solve, a=sub (a[0] unused), b=diag, c=super (c[-1] unused)."""
    n = len(b)
    cp = np.empty(n); dp = np.empty(n); x = np.empty(n)
    cp[0] = c[0] / b[0]; dp[0] = d[0] / b[0]
    for i in range(1, n):
        m = b[i] - a[i] * cp[i - 1]
        cp[i] = c[i] / m if i < n - 1 else 0.0
        dp[i] = (d[i] - a[i] * dp[i - 1]) / m
    x[-1] = dp[-1]
    for i in range(n - 2, -1, -1):
        x[i] = dp[i] - cp[i] * x[i + 1]
    return x


def poisson(C, p, solver='dense'):
    """calculator_old.py:680-819.  Returns v, grad_v, lapl_v (each [nx])."""
    nx, dx = p.nx, p.dx
    lapl_v = np.zeros(nx)
    for k in range(p.N):                      # :767-771
        lapl_v -= p.charges[k] * C[k, :] / p.eps
    v = np.zeros(nx); grad_v = np.zeros(nx)
    vw, vb, gw, gb = p.pb
    mode = pb_mode_from_bound(p.pb)
    if not np.isnan(vw):
        v[0] = vw
    if not np.isnan(vb):
        v[-1] = vb
    if mode == PB_DD:                         # :780-786, solve_poisson :716-730
        m = nx - 2
        b = lapl_v[1:nx - 1] * dx ** 2
        b = b.copy()
        b[0] -= v[0]
        b[-1] -= v[-1]
        if solver == 'dense':
            A = (np.diag(-2.0 * np.ones(m)) + np.diag(np.ones(m - 1), 1) + np.diag(np.ones(m - 1), -1))
            x = np.linalg.solve(A, b)
        else:
            x = _thomas(np.ones(m), -2.0 * np.ones(m), np.ones(m), b)
        v[1:nx - 1] = x
        grad_v[1:nx - 1] = 1. / (2 * dx) * (v[2:] - v[:-2])
        grad_v[0] = grad_v[1] + (grad_v[1] - grad_v[2])
        grad_v[-1] = grad_v[-2] + (grad_v[-2] - grad_v[-3])
    else:                                     # :787-803  (integrate_1d_func n=1, :752-761)
        if not np.isnan(gw):
            grad_v[0] = gw
            for i in range(1, nx - 1):
                grad_v[i] = grad_v[i - 1] + lapl_v[i] * dx
            grad_v[-1] = grad_v[-2] + (grad_v[-2] - grad_v[-3])
        if not np.isnan(gb):
            grad_v[-1] = gb
            for i in range(nx - 2, 0, -1):
                grad_v[i] = grad_v[i + 1] - lapl_v[i] * dx
            grad_v[0] = grad_v[1] + (grad_v[1] - grad_v[2])
        if not np.isnan(vw):
            for i in range(1, nx - 1):
                v[i] = v[i - 1] + grad_v[i] * dx
            v[-1] = v[-2] + (v[-2] - v[-3])
        if not np.isnan(vb):
            for i in range(nx - 2, 0, -1):
                v[i] = v[i + 1] - grad_v[i] * dx
            v[0] = v[1] + (v[1] - v[2])
    return v, grad_v, lapl_v


def get_rates(C, p):
    """calculator_old.py:159-208 including the 'rates[k,i]=0.0' overwrite (:173,:193)."""
    rates = np.zeros_like(C)
    for lhs, rhs, kf, kr in p.reactions:
        pl = np.ones(C.shape[1]); pr = np.ones(C.shape[1])
        for k2 in lhs:
            pl = pl * C[k2]
        for k2 in rhs:
            pr = pr * C[k2]
        for k in lhs:
            rates[k] = 0.0
            rates[k] -= pl * kf
            rates[k] += pr * kr
        for k in rhs:
            rates[k] = 0.0
            rates[k] += pl * kf
            rates[k] -= pr * kr
    return rates


def cn_step(C, COLD, C0, p, first, solver='dense'):
    """One pass of the time loop body calculator_old.py:512-558 (in place on C, COLD).
    Returns (v, grad_v, lapl_v) used by this step (lagged potential)."""
    nx, dx, dt = p.nx, p.dx, p.dt
    m = nx - 2
    if not p.use_migration:
        raise NameError("reference reads 'v' unbound when use_migration is False (calculator_old.py:514,529)")
    v, grad_v, lapl_v = poisson(C, p, solver)
    mu = p.mu
    for k in range(p.N):
        if first:
            COLD[k, :] = C[k, :]
        a = mu[k] * (v[1] - p.vzeta)
        C[k, 0] = (-2 * p.D[k] - a) / (-2 * p.D[k] + a) * C[k, 1] \
            - 2 * p.flux_bound[k] * dx / (-2 * p.D[k] + a)                     # :528-532
        C[k, -1] = C0[k, -1]                                                    # :540
        s = p.D[k] * dt / dx ** 2
        if p.lax_friedrich:
            s += 0.5
        ee = p.charges[k] * p.beta * dt * p.D[k]
        r = np.arange(m)
        g = ee * grad_v[:m] / 4. / dx                       # NB index r, not r+1 (:483-490)
        # A = tridiag(-s/2 + g_r, 1+s, -s/2 - g_r);  B1 = tridiag(s/2 - g_r, 1-s+ee*lapl_r, s/2 + g_r)
        A_lo = -0.5 * s + g; A_up = -0.5 * s - g; A_di = (1 + s) * np.ones(m)
        B_lo = 0.5 * s - g; B_up = 0.5 * s + g; B_di = 1 - s + ee * lapl_v[:m]
        ci = C[k, 1:-1]
        if solver == 'dense':
            A = np.diag(A_di) + np.diag(A_up[:-1], 1) + np.diag(A_lo[1:], -1)
            B1 = np.diag(B_di) + np.diag(B_up[:-1], 1) + np.diag(B_lo[1:], -1)
            B = np.dot(ci, B1)                                                   # :553 row-vector x matrix
        else:
            B = ci * B_di
            B[1:] += ci[:-1] * B_up[:-1]      # B[j] += c[j-1]*B1[j-1,j]
            B[:-1] += ci[1:] * B_lo[1:]       # B[j] += c[j+1]*B1[j+1,j]
        B[0] += (0.5 * s + ee * grad_v[0] / 4. / dx) * (C[k, 0] + COLD[k, 0])    # :496-499
        B[-1] += (0.5 * s - ee * grad_v[-1] / 4. / dx) * (C[k, -1] + COLD[k, -1])
        if solver == 'dense':
            C[k, 1:-1] = np.linalg.solve(A, B)
        else:
            C[k, 1:-1] = _thomas(A_lo, A_di, A_up, B)
        COLD[k, :] = C[k, :]
    return v, grad_v, lapl_v


def ftcs_step(C, C0, p, solver='dense'):
    """One pass of calculator_old.py:990-1023 (in place on C)."""
    nx, dx, dt = p.nx, p.dx, p.dt
    if p.use_migration:
        v, grad_v, lapl_v = poisson(C, p, solver)
    else:
        v = np.zeros(nx); grad_v = np.zeros(nx); lapl_v = np.zeros(nx)
    rates = get_rates(C, p)
    mu = p.mu
    for k in range(p.N):
        flux = p.flux_bound[k]
        divisor = 2 * p.D[k] - mu[k] * (v[1] - p.vzeta)
        C[k, 0] = ((2 * p.D[k] + mu[k] * (v[1] - p.vzeta)) * C[k, 1] + flux * 2. * dx) / divisor   # :1003-1006
        C[k, -1] = C0[k, -1]
        temp = np.zeros(nx)
        temp[0] = C[k, 0]; temp[-1] = C[k, -1]
        i = np.arange(1, nx - 1)
        W = p.D[k] * dt / dx ** 2 - dt / (2. * dx) * mu[k] * grad_v[i + 1] + 0.5
        M = -2. * p.D[k] * dt / dx ** 2 * np.ones(nx - 2)
        E = p.D[k] * dt / dx ** 2 + dt / (2. * dx) * mu[k] * grad_v[i - 1] + 0.5
        if not p.lax_friedrich:
            W = W - 0.5; E = E - 0.5; M = M + 1
        temp[i] = E * C[k, i - 1] + M * C[k, i] + W * C[k, i + 1] + rates[k, i] * dt
        C[k, :] = temp
    return v, grad_v, lapl_v


def mol_rhs(c, p, use_reactions=False, solver='dense'):
    """ode_func, calculator_old.py:827-935: dc/dt for flat c[N*nx]."""
    nx, dx, dt = p.nx, p.dx, p.dt
    C = c.reshape(p.N, nx).copy()
    if p.use_migration:
        v, grad_v, lapl_v = poisson(C, p, solver)
    else:
        grad_v = np.zeros(nx)
    rates = get_rates(C, p) if use_reactions else np.zeros_like(C)
    out = np.zeros_like(C)
    for k in range(p.N):
        # i = 0 (wall cell), :897-915 -- no rate term here
        corr = (C[k, 1] - C[k, 0]) / dt if p.lax_friedrich else 0.0
        out[k, 0] = corr + (p.D[k] * ((C[k, 2] - C[k, 0]) / (2. * dx)
                                      + p.beta * p.charges[k] * C[k, 1] * grad_v[1]) - p.flux_bound[k]) / dx
        i = np.arange(1, nx - 1)
        d2 = (C[k, i + 1] - 2 * C[k, i] + C[k, i - 1]) / (dx ** 2)
        dcg = (C[k, i + 1] * grad_v[i + 1] - C[k, i - 1] * grad_v[i - 1]) / (2. * dx) if p.use_migration else 0.0
        corr = d2 * dx ** 2 / dt / 2. if p.lax_friedrich else 0.0
        out[k, i] = corr + p.D[k] * (d2 + p.beta * p.charges[k] * dcg) + rates[k, i]
        out[k, -1] = 0.0
    return out.reshape(-1)


def integrate(p, c0, nt, itout, method='Crank-Nicolson', solver='dense'):
    """integrate_pnp for the two hand-written integrators (calculator_old.py:1121-1140).
    c0 is the flat species-major [N*nx] state; returns (cout[list of flat arrays], v, grad_v, lapl_v
    of the last Poisson solve == tp.potential, -tp.efield, -tp.total_charge/eps)."""
    C0 = np.array(c0, dtype=np.float64).reshape(p.N, p.nx)
    C = C0.copy()
    cout = []
    last = (None, None, None)
    itout = set(int(i) for i in itout)
    if method == 'Crank-Nicolson':
        COLD = np.zeros_like(C)
        for n in range(1, nt):                     # :512
            last = cn_step(C, COLD, C0, p, first=(n == 1), solver=solver)
            if n in itout:
                cout.append(C.reshape(-1).copy())
    elif method == 'FTCS':
        for n in range(0, nt):                     # :990
            last = ftcs_step(C, C0, p, solver=solver)
            if n in itout:
                cout.append(C.reshape(-1).copy())
    else:
        raise ValueError(method)
    return cout, last


def problem_from_golden(d):
    """Build a Problem (+ c0, nt, itout, method) from a tests/golden/*.npz fixture."""
    method = str(d['method'])
    names = [str(s) for s in d['species']]
    import json
    reactions = encode_reactions(json.loads(str(d['reactions_json'])), names)
    p = Problem(D=np.array(d['D']), charges=np.array(d['charges']), beta=float(d['beta']), eps=float(d['eps']),
                dx=float(d['dx']), nx=int(d['nx']), dt=float(d['dt']), pb=np.array(d['pb_bound']),
                vzeta=float(d['vzeta']), flux_bound=np.array(d['flux_bound'])[:, 0].copy(),
                lax_friedrich=bool(d['lax_friedrich']), use_migration=bool(d['use_migration']),
                reactions=reactions)
    return p, np.array(d['c0']), int(d['nt']), [int(i) for i in d['itout']], method.split('--')[0]
