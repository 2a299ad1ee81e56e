"""CPU ORACLE (test infrastructure, NOT product code) for the *physical* mode: fully implicit
(backward-Euler or stationary) 1D Poisson-Nernst-Planck solved as ONE coupled nonlinear system per timestep by
Newton's method with a block-tridiagonal Jacobian ((N+1)x(N+1) blocks: N species + potential per grid point).

PARITY UNPINNED against the reference: sringe/CatINT delegates this solve to the closed-source COMSOL
binary (catint/comsol_wrapper.py:145,158); the model it generates (catint/comsol_model.py, SURVEY.md
App. C) is the physics restated here -- Poisson `-d/dx(eps dphi/dx) = F sum z_i c_i` (:609-612, :1003-1009),
Nernst-Planck `d/dx(-D_i dc_i/dx - z_i u_i F c_i dphi/dx + c_i u_i) = R_i` (:682-919) with the size-modified
(MPB) drift `u_i = -D_i grad(phi0)/(1-phi0)`, `phi0 = N_A sum a_i^3 c_i` (:1041-1063), mass-action homogeneous
reactions in activities `a_i = c_i/(1-phi0)` (:781-867, :1064-1084), wall flux `N0_i = j_i` (:770), bulk
`c_i = c_bulk` (:771-772), wall potential Dirichlet or the Stern Robin condition
`rho_s = ((phiM-phiPZC)-phi)*C_S` (:613, :982), bulk potential Dirichlet (:662-663).
What pins it instead are analytic known answers the reference itself carries: the Gouy-Chapman potential
(catint/transport.py:1373-1383), Boltzmann profiles (:1325-1346), the Debye length (:439-443), plus mass
conservation, the MPB saturation limit and Newton's quadratic convergence (tests/test_physical_oracle.py).

Discretisation (stated for the uniform grid x_i = i*dx, i = 0..nx-1; on a non-uniform grid every edge flux carries the
weight dx/h_e and every storage/source term the control volume V_i/dx, see PhysicalProblem; conservative;
exponentially fitted = Scharfetter-Gummel
fluxes, which reproduce the Boltzmann distribution exactly on any grid and stay monotone at any bias):
  generalised drift potential of species k:  psi_k = q_k beta phi + w,   w = -ln(1 - phi0)   (w = 0 without MPB)
  edge i+1/2:  u = psi_k[i+1] - psi_k[i],   Jhat_k = J_k dx / D_k = -( B(-u) c_k[i+1] - B(u) c_k[i] ),
               B(u) = u / (exp(u) - 1)
  rows are scaled to O(1): species rows by dx^2/D_k, the Poisson row by dx^2/eps
  interior 0<i<nx-1:  sigma_k (c_i - c_i^old) + Jhat_{i+1/2} - Jhat_{i-1/2} - (dx^2/D_k) R_k(c_i) = 0,
                      sigma_k = dx^2/(D_k dt)   (0 for the stationary problem)
                      phi_{i+1} - 2 phi_i + phi_{i-1} + (dx^2/eps) sum_k q_k c_{k,i} = 0
  wall i=0:           sigma_k/2 (c_0 - c_0^old) + Jhat_{1/2} - j_k dx/D_k - (dx^2/(2 D_k)) R_k(c_0) = 0
                      (half cell; j_k = flux INTO the domain)
                      phi_0 = phiM                                         (Dirichlet)   or
                      (phi_1 - phi_0) + (dx C_S/eps) (phiM - phiPZC - phi_0) = 0   (Stern Robin)
  bulk i=nx-1:        c = c_bulk, phi = phi_bulk

The device kernel (catint_amd/csrc/pnp_newton.hip) evaluates the same formulas in the same order; its linear
solver is block parallel cyclic reduction, mirrored here by `solve_block_pcr` (the default is LAPACK's banded LU).
"""
import numpy as np
from scipy.linalg import solve_banded

N_AVOGADRO = 6.022140857e23      # catint/units.py (unit_NA)
FREE_MIN = 1e-12                 # floor of the free volume fraction 1-phi0 the damped update can reach
SERIES_U = 0.05                  # |u| below which B(u) and B'(u) use their Taylor series


class PhysicalProblem(object):
    """One operating point.  reactions: list of dicts {'lhs': [species idx...], 'rhs': [...], 'kf':, 'kr':}.
    wall_kinetics: list of dicts {'species': s (or -1 for zeroth order), 'k': K, 'nu': [nu_k], 'alpha': a (optional, 1/V),
    'saturation': K_sat (optional, m^3/mol)}: surface reactions whose flux INTO the domain,
    nu_k K c_s/(1 + K_sat c_s) exp(a (phiM - phi(x=0))), is part of the nonlinear system (what the SCF loop of
    catint/calculator.py:294-406 converges to when the kinetics are rate = K(phiM) c_surface; alpha / saturation: the
    Butler-Volmer / Langmuir forms of the user-defined flux equations, docs/source/topics/flux_definition.rst:90-160)."""

    def __init__(self, D, charges, beta, eps, dx, nx, c_bulk, phiM, flux=None, phi_bulk=0.0, stern_capacitance=None,
                 phi_pzc=0.0, mpb_radius=None, reactions=None, wall_kinetics=None, x=None, velocity=0.0):
        self.D = np.asarray(D, float)
        self.q = np.asarray(charges, float)        # z*F
        self.beta, self.eps, self.dx, self.nx = float(beta), float(eps), float(dx), int(nx)
        self.c_bulk = np.asarray(c_bulk, float)
        self.phiM, self.phi_bulk = float(phiM), float(phi_bulk)
        self.N = len(self.D)
        self.flux = np.zeros(self.N) if flux is None else np.asarray(flux, float)
        self.CS = None if stern_capacitance is None else float(stern_capacitance)
        self.phi_pzc = float(phi_pzc)
        a = np.zeros(self.N) if mpb_radius is None else np.asarray(mpb_radius, float)
        self.vol = N_AVOGADRO * a ** 3                     # m^3/mol, phi0 = sum vol_k c_k
        self.mpb = bool(np.any(self.vol != 0.0))
        self.reactions = list(reactions or [])
        self.wall_kinetics = list(wall_kinetics or [])
        self.velocity = float(velocity)                    # constant convection velocity along x (tp.system['flow rate'])
        # grid: uniform x_i = i*dx, or any increasing x[nx] (then dx is only the reference length of the row scaling).
        # w[e] = dx/h_e weights the flux across edge e (between points e and e+1), v[i] = V_i/dx is the control volume
        # (half cells at the ends)
        self.x = np.arange(self.nx) * self.dx if x is None else np.asarray(x, float)
        h = np.diff(self.x)
        self.w = self.dx / h
        self.v = np.empty(self.nx)
        self.v[1:-1] = 0.5 * (h[1:] + h[:-1]) / self.dx
        self.v[0] = 0.5 * h[0] / self.dx
        self.v[-1] = 0.5 * h[-1] / self.dx


def bernoulli(u):
    """B(u) = u/(exp(u)-1) and dB/du, series for |u| < SERIES_U."""
    u = np.asarray(u, float)
    small = np.abs(u) < SERIES_U
    us = np.where(small, u, 0.0)
    u2 = us * us
    Bs = 1.0 - 0.5 * us + u2 * (1.0 / 12.0 + u2 * (-1.0 / 720.0 + u2 * (1.0 / 30240.0)))
    dBs = -0.5 + us * (1.0 / 6.0 + u2 * (-1.0 / 180.0 + u2 * (1.0 / 5040.0)))
    ul = np.where(small, 1.0, u)
    with np.errstate(over='ignore', invalid='ignore'):
        E = np.expm1(ul)
        Bl = ul / E
        dBl = (1.0 - Bl - ul) / E
    return np.where(small, Bs, Bl), np.where(small, dBs, dBl)


def _steric(p, c):
    """w = -ln(1-phi0) [nx] and g[k, i] = d w_i / d c_{k,i} = vol_k/(1-phi0_i)."""
    if not p.mpb:
        return np.zeros(c.shape[1]), np.zeros_like(c)
    phi0 = (p.vol[:, None] * c).sum(axis=0)
    return -np.log1p(-phi0), p.vol[:, None] / (1.0 - phi0)[None, :]


def reaction_rates(p, c):
    """R[k, i] and dR[k, j, i] = dR_k/dc_j at the same point (mass action in activities a = c/(1-phi0))."""
    N, nx = c.shape
    R = np.zeros((N, nx))
    dR = np.zeros((N, N, nx))
    if not p.reactions:
        return R, dR
    if p.mpb:
        phi0 = (p.vol[:, None] * c).sum(axis=0)
        gam = 1.0 / (1.0 - phi0)
        dgam = p.vol[:, None] * (gam * gam)[None, :]         # d gamma / d c_j
    else:
        gam = np.ones(nx)
        dgam = np.zeros((N, nx))
    for r in p.reactions:
        for side, kk, sign in ((r['lhs'], r['kf'], 1.0), (r['rhs'], r['kr'], -1.0)):
            if kk == 0.0:                       # an empty side is a constant rate (excluded species such as H2O)
                continue
            m = len(side)
            prod = kk * gam ** m
            for j in side:
                prod = prod * c[j]
            # d prod / d c_j = mult_j * (product with one occurrence of c_j removed)  [+ gamma dependence below]
            dprod = np.zeros((N, nx))
            for j in set(side):
                rest = kk * gam ** m
                skipped = False
                for jj in side:
                    if jj == j and not skipped:
                        skipped = True
                        continue
                    rest = rest * c[jj]
                dprod[j] += side.count(j) * rest
            if p.mpb:
                dprod += prod[None, :] * m * dgam / gam[None, :]
            rate, drate = sign * prod, sign * dprod           # contribution to (forward - backward)
            for j in r['lhs']:
                R[j] -= rate
                dR[j] -= drate
            for j in r['rhs']:
                R[j] += rate
                dR[j] += drate
    return R, dR


def wall_rate_law(p, wk, c, phi):
    """g, dg/dc_s, alpha of one surface reaction at the wall state: rate = K g, g = c_s/(1 + K_sat c_s) exp(alpha (phiM - phi_0))
    (so d rate/d phi_0 = -alpha K g)."""
    cs = c[wk['species'], 0] if wk['species'] >= 0 else 1.0
    al, ks = float(wk.get('alpha', 0.0)), float(wk.get('saturation', 0.0))
    den = 1.0 / (1.0 + ks * cs)
    E = np.exp(al * (p.phiM - phi[0])) if al != 0.0 else 1.0
    return cs * den * E, den * den * E, al


def residual_and_jacobian(p, c, phi, c_old, dt, want_jacobian=True):
    """Scaled residual F[(N+1), nx] and the block-tridiagonal Jacobian (L, M, U)[nx, N+1, N+1]."""
    N, nx, dx = p.N, p.nx, p.dx
    nb = N + 1
    F = np.zeros((nb, nx))
    L = np.zeros((nx, nb, nb)); M = np.zeros((nx, nb, nb)); U = np.zeros((nx, nb, nb))
    w, g = _steric(p, c)
    R, dR = reaction_rates(p, c)
    dphi = phi[1:] - phi[:-1]
    dw = w[1:] - w[:-1]
    ii = np.arange(1, nx - 1)
    for k in range(N):
        qb = p.q[k] * p.beta
        sig = 0.0 if np.isinf(dt) else dx * dx / (p.D[k] * dt)
        rs = dx * dx / p.D[k]
        u = qb * dphi + dw
        if p.velocity != 0.0:                                  # + c v in the flux (comsol_model.py:901-903: tds "u" = system['flow rate'])
            u = u - p.velocity * dx / (p.D[k] * p.w)
        Bp, dBp = bernoulli(u)
        Bm = Bp + u
        cl, cr = c[k, :-1], c[k, 1:]
        Bp, Bm = p.w * Bp, p.w * Bm                            # edge weights dx/h_e of a non-uniform grid (1 if uniform)
        J = -(Bm * cr - Bp * cl)                               # Jhat at edge e = i+1/2, index i
        Ju = -p.w * ((dBp + 1.0) * cr - dBp * cl)              # dJhat/du
        v = p.v
        F[k, 1:-1] = sig * v[1:-1] * (c[k, 1:-1] - c_old[k, 1:-1]) + J[1:] - J[:-1] - rs * v[1:-1] * R[k, 1:-1]
        jw = p.flux[k]
        for wk in p.wall_kinetics:
            jw = jw + wk['nu'][k] * wk['k'] * wall_rate_law(p, wk, c, phi)[0]
        F[k, 0] = sig * v[0] * (c[k, 0] - c_old[k, 0]) + J[0] - jw * dx / p.D[k] - rs * v[0] * R[k, 0]
        F[k, -1] = c[k, -1] - p.c_bulk[k]
        if not want_jacobian:
            continue
        # interior row i: +J[i] (left point i, right point i+1)  -J[i-1] (left point i-1, right point i)
        M[ii, k, k] += sig * v[ii] + Bp[ii] + Bm[ii - 1]       # dJ[i]/dc_l = +Bp ; -dJ[i-1]/dc_r = +Bm
        U[ii, k, k] += -Bm[ii]
        L[ii, k, k] += -Bp[ii - 1]
        M[ii, k, N] += Ju[ii] * (-qb) - Ju[ii - 1] * qb
        U[ii, k, N] += Ju[ii] * qb
        L[ii, k, N] += Ju[ii - 1] * qb
        M[0, k, k] += sig * v[0] + Bp[0]
        U[0, k, k] += -Bm[0]
        M[0, k, N] += -Ju[0] * qb
        U[0, k, N] += Ju[0] * qb
        for wk in p.wall_kinetics:
            gw, dgw, al = wall_rate_law(p, wk, c, phi)
            if wk['species'] >= 0:
                M[0, k, wk['species']] += -wk['nu'][k] * wk['k'] * dgw * dx / p.D[k]
            M[0, k, N] += wk['nu'][k] * wk['k'] * al * gw * dx / p.D[k]
        for j in range(N):                                    # steric coupling through u and reactions
            if p.mpb:
                M[ii, k, j] += -Ju[ii] * g[j, ii] - Ju[ii - 1] * g[j, ii]
                U[ii, k, j] += Ju[ii] * g[j, ii + 1]
                L[ii, k, j] += Ju[ii - 1] * g[j, ii - 1]
                M[0, k, j] += -Ju[0] * g[j, 0]
                U[0, k, j] += Ju[0] * g[j, 1]
            M[ii, k, j] += -rs * v[ii] * dR[k, j, ii]
            M[0, k, j] += -rs * v[0] * dR[k, j, 0]
        M[nx - 1, k, k] = 1.0
    pe = dx * dx / p.eps
    rho = (p.q[:, None] * c).sum(axis=0)
    w, v = p.w, p.v
    F[N, 1:-1] = w[1:] * (phi[2:] - phi[1:-1]) - w[:-1] * (phi[1:-1] - phi[:-2]) + pe * v[1:-1] * rho[1:-1]
    if p.CS is None:
        F[N, 0] = phi[0] - p.phiM
    else:
        F[N, 0] = w[0] * (phi[1] - phi[0]) + (dx * p.CS / p.eps) * (p.phiM - p.phi_pzc - phi[0])
    F[N, -1] = phi[-1] - p.phi_bulk
    if want_jacobian:
        for k in range(N):
            M[ii, N, k] = pe * v[ii] * p.q[k]
        M[ii, N, N] = -(w[ii] + w[ii - 1])
        L[ii, N, N] = w[ii - 1]
        U[ii, N, N] = w[ii]
        if p.CS is None:
            M[0, N, N] = 1.0
        else:
            M[0, N, N] = -w[0] - dx * p.CS / p.eps
            U[0, N, N] = w[0]
        M[nx - 1, N, N] = 1.0
    return F, L, M, U


def residual(p, c, phi, c_old, dt):
    return residual_and_jacobian(p, c, phi, c_old, dt, want_jacobian=False)[0]


def solve_block_tridiagonal(L, M, U, rhs):
    """LAPACK banded LU (partial pivoting) on the point-major unknown ordering.  rhs, result: [(N+1), nx]."""
    nx, nb, _ = M.shape
    n = nx * nb
    kl = ku = 2 * nb - 1
    ab = np.zeros((kl + ku + 1, n))
    rr, ss = np.meshgrid(np.arange(nb), np.arange(nb), indexing='ij')
    for (blk, joff) in ((L, -1), (M, 0), (U, 1)):
        i0, i1 = max(0, -joff), min(nx, nx - joff)
        for i in range(i0, i1):
            row = i * nb + rr
            col = (i + joff) * nb + ss
            ab[ku + row - col, col] = blk[i]
    x = solve_banded((kl, ku), ab, rhs.T.reshape(-1))
    return x.reshape(nx, nb).T


def solve_block_pcr(L, M, U, rhs):
    """Mirror of the device solver: rows normalised to a unit diagonal block, then parallel cyclic reduction
    over all nx block rows (out-of-range neighbours are empty rows)."""
    nx, nb, _ = M.shape
    r = rhs.T.copy()                                           # [nx, nb]
    aug = np.concatenate([L, U, r[:, :, None]], axis=2)        # [nx, nb, 2nb+1]
    aug = np.linalg.solve(M, aug)
    Lt, Ut, rt = aug[:, :, :nb].copy(), aug[:, :, nb:2 * nb].copy(), aug[:, :, 2 * nb].copy()
    s = 1
    eye = np.eye(nb)[None]
    while s < nx:
        def shifted(A, d):
            out = np.zeros_like(A)
            if d > 0:
                out[:-d] = A[d:]
            else:
                out[-d:] = A[:d]
            return out
        Lm, Um, rm = shifted(Lt, -s), shifted(Ut, -s), shifted(rt, -s)     # row i-s
        Lp, Up, rp = shifted(Lt, s), shifted(Ut, s), shifted(rt, s)        # row i+s
        Dm = eye - Lt @ Um - Ut @ Lp
        nL = -(Lt @ Lm)
        nU = -(Ut @ Up)
        nr = rt - np.einsum('irs,is->ir', Lt, rm) - np.einsum('irs,is->ir', Ut, rp)
        aug = np.linalg.solve(Dm, np.concatenate([nL, nU, nr[:, :, None]], axis=2))
        Lt, Ut, rt = aug[:, :, :nb], aug[:, :, nb:2 * nb], aug[:, :, 2 * nb]
        s *= 2
    return rt.T.copy()


def at_rounding_floor(upd, upd_prev, tol):
    """The third exit of the Newton loop (newton_at_rounding_floor in catint_amd/csrc/pnp_math.h): two consecutive full steps within
    100 tol, the second NOT SMALLER than the first -- near the solution Newton contracts quadratically, so an update that stops
    shrinking is the noise of an ill-conditioned Jacobian, not progress.  A linearly converging iteration (every update a fixed
    fraction of the one before) shrinks at every step and never leaves here."""
    w = 100.0 * tol
    return upd_prev < w and upd < w and upd >= upd_prev


def newton_step(p, c, phi, c_old, dt, tol=1e-10, maxit=50, dphi_max=0.05, solver=solve_block_tridiagonal,
                verbose=False, estimate=False, jacobian_once=False):
    """Solve one backward-Euler step (or, with dt=inf, the stationary problem) by damped Newton.
    Damping (identical on the device): the whole update is scaled so that |d phi| <= dphi_max, a concentration
    never drops below 10 % of its previous iterate, and neither does the free volume fraction 1-phi0 (MPB).
    Converged when the scaled update max(|dc_k,i|/(c_k,i + c_bulk_k), |dphi| beta max|q|) < tol on a full step, or when the
    iteration sits on its rounding floor (see below).
    jacobian_once=True: the Jacobian of the first iterate serves every iteration of the step (the chord iteration COMSOL is told
    to run, comsol_model.py:526,530 jtech "once") -- a measuring option of the oracle only (tools/probe/jacobian_once_oracle.py: how many
    iterations the chord needs on the bench workloads); the library factorises every iteration.
    Returns (c, phi, iterations, history of update norms); iterations = maxit+1 if not converged."""
    c = c.copy(); phi = phi.copy()
    N = p.N
    vt = 1.0 / (p.beta * max(np.abs(p.q).max(), 1.0))            # thermal voltage of the highest valence
    hist = []
    upd_prev = np.inf
    for it in range(1, maxit + 1):
        if jacobian_once and it > 1:
            F = residual(p, c, phi, c_old, dt)
        else:
            F, L, M, U = residual_and_jacobian(p, c, phi, c_old, dt)
        if not (np.all(np.isfinite(F)) and np.all(np.isfinite(M))):      # diverged: the device keeps iterating on NaNs and reports
            c = np.full_like(c, np.nan)                                   # maxit+1 iterations and the NaN status
            phi = np.full_like(phi, np.nan)
            break
        du = solver(L, M, U, -F)
        lam = 1.0
        if dphi_max is not None:
            m = np.abs(du[N]).max()
            if m > dphi_max:
                lam = dphi_max / m
        c_prev = c
        cn = c + lam * du[:N]
        cn = np.where(cn < 0.1 * c, 0.1 * c, cn)
        if p.mpb:                                              # same rule for the free volume fraction 1-phi0
            phi0_old = (p.vol[:, None] * c).sum(axis=0)
            phi0_new = (p.vol[:, None] * cn).sum(axis=0)
            free = 1.0 - phi0_old
            target = np.maximum(0.1 * free, FREE_MIN)
            over = (1.0 - phi0_new) < target
            if np.any(over):
                theta = np.where(over, (free - target) / np.where(over, phi0_new - phi0_old, 1.0), 1.0)
                cn = c + theta[None, :] * (cn - c)
        c = cn
        phi = phi + lam * du[N]
        upd = max((np.abs(du[:N]) / (np.abs(c_prev) + np.abs(p.c_bulk)[:, None] + 1e-300)).max(), np.abs(du[N]).max() / vt)
        hist.append(upd)
        if verbose:
            print(it, lam, upd)
        if lam == 1.0:
            # estimate=True: also accept when two consecutive undamped iterations contract and the quadratic estimate
            # upd^2/upd_prev of the error of the state just computed is below tol
            if upd < tol or (estimate and np.isfinite(upd_prev) and upd < 0.1 * upd_prev and upd * (upd / upd_prev) < tol):
                return c, phi, it, hist
            if at_rounding_floor(upd, upd_prev, tol):
                return c, phi, it, hist
            upd_prev = upd
        else:
            upd_prev = np.inf
    return c, phi, maxit + 1, hist


def integrate(p, c0, phi0, dt, nsteps, bdf2=False, predictor=False, **kw):
    """nsteps implicit timesteps.  Backward Euler, or (bdf2=True) the second-order backward differentiation formula the reference's
    transient study asks COMSOL for (comsol_model.py:518-531: BDF, maxorder 2): (3 c_n+1 - 4 c_n + c_n-1) / (2 dt) -- a backward-Euler
    step of length dt / 1.5 against the combination c* = (4 c_n - c_n-1) / 3; the first step is backward Euler.
    predictor=True: from the second step on Newton starts from the linear extrapolation 2 u_n - u_n-1 (concentrations not below a tenth
    of their value; a point whose extrapolated ions would fill more than 90 % of the volume keeps u_n) -- step_prepare_kernel in
    catint_amd/csrc/pnp_capi.hip."""
    c, phi = c0.copy(), phi0.copy()
    c_prev = phi_prev = None
    its = []
    for _ in range(nsteps):
        if (bdf2 or predictor) and c_prev is not None:
            c_old = (4.0 * c - c_prev) / 3.0 if bdf2 else c
            cs, ps = c, phi
            if predictor:
                g = 2.0 * c - c_prev
                g = np.where(g < 0.1 * c, 0.1 * c, g)
                fill = (p.vol[:, None] * g).sum(axis=0) if p.mpb else np.zeros(c.shape[1])
                ext = fill < 0.9
                cs = np.where(ext[None, :], g, c)
                ps = np.where(ext, 2.0 * phi - phi_prev, phi)
            c_prev, phi_prev = c, phi
            c, phi, it, _ = newton_step(p, cs, ps, c_old, dt / 1.5 if bdf2 else dt, **kw)
        else:
            c_prev, phi_prev = c, phi
            c, phi, it, _ = newton_step(p, c, phi, c, dt, **kw)
        its.append(it)
    return c, phi, its


def gouy_chapman(x, phiM, beta, q_abs, debye_length):
    """catint/transport.py:1373-1383 (z:z electrolyte)."""
    g = np.tanh(phiM * beta * q_abs / 4.)
    return 2. / (beta * q_abs) * np.log((1. + g * np.exp(-x / debye_length)) / (1. - g * np.exp(-x / debye_length)))
