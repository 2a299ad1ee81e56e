"""Host-side glue between problem descriptions and the C-ABI solver."""
import numpy as np

from ._capi import PnpSolver, PB_DD, PB_VWALL_GBULK, PB_GWALL_VBULK, PB_VWALL_GWALL, PB_VBULK_GBULK


def pb_mode_from_bound(pb):
    """pb = [potential wall, potential bulk, gradient wall, gradient bulk], NaN/None = unset
    (reference tp.pb_bound, catint/transport.py:1296-1311; branch selection
    catint/calculator_old.py:712-714, :776-803)."""
    vw, vb, gw, gb = [x is not None and not np.isnan(x) for x in pb]
    if gw and gb:
        raise ValueError('Cannot use two boundary conditions for gradient')
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
    raise ValueError('unsupported pb_bound combination: %r' % (pb,))


def solver_from_problem(p, method, batch_capacity=1, device=0):
    """Build a PnpSolver from any object with the reference's per-problem fields
    (D, charges, beta, eps, dx, nx, dt, pb, lax_friedrich, use_migration, reactions)."""
    s = PnpSolver(nspecies=len(p.D), nx=p.nx, dx=p.dx, dt=p.dt, beta=p.beta, eps=p.eps, D=p.D, charges=p.charges,
                  method=method, pb_mode=pb_mode_from_bound(p.pb), lax_friedrich=p.lax_friedrich,
                  use_migration=p.use_migration, batch_capacity=batch_capacity, device=device)
    if getattr(p, 'reactions', None):
        s.set_reactions(p.reactions)
    return s


def graded_mesh(length, h0, nx):
    """Geometric grid x[nx] on [0, length] with first spacing h0 (the electrode end) and constant growth ratio: resolves a
    nanometre double layer inside a micrometre diffusion layer, like the reference's COMSOL mesh
    (hmax = L/grid_factor_domain in the domain, lambda_D/grid_factor_bound at the boundaries, comsol_model.py:588,593)."""
    import numpy as np
    n = int(nx) - 1
    if n * h0 >= length:
        return np.linspace(0.0, length, nx)
    lo, hi = 1.0 + 1e-12, 2.0
    f = lambda r: h0 * (r ** n - 1.0) / (r - 1.0) - length
    while f(hi) < 0:
        hi *= 2.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if f(mid) > 0:
            hi = mid
        else:
            lo = mid
    r = 0.5 * (lo + hi)
    x = np.concatenate([[0.0], np.cumsum(h0 * r ** np.arange(n))])
    x[-1] = length
    return x


def booth_permittivity(E, eps_r, n=1.33, beta=1.41e-8):
    """Field-dependent relative permittivity of water (Booth), as the reference's reader uses it for the Stern layer
    (comsol_reader.py:102-114): n^2 + (eps_r - n^2) * 3/(beta E) * (coth(beta E) - 1/(beta E)) for |E| >= 1e7 V/m, eps_r below."""
    import numpy as np
    E = abs(float(E))
    if E < 1e7:
        return float(eps_r)
    x = beta * E
    return n ** 2 + (eps_r - n ** 2) * 3.0 / x * (1.0 / np.tanh(x) - 1.0 / x)


def booth_stern_field(E_out, eps_r):
    """Field inside the Stern layer from the field at its outer plane: the root of E = E_out * eps_r / eps_Booth(|E|)
    (comsol_reader.py:115-119, :262-273; the reference minimises the squared mismatch by basin hopping).  The right-hand side
    grows with |E| (the permittivity drops), bounded by eps_r/n^2: bisection on a bracket that always contains the root."""
    E_out = float(E_out)
    if E_out == 0.0:
        return 0.0, float(eps_r)
    sgn = 1.0 if E_out > 0 else -1.0
    a = abs(E_out)
    f = lambda E: E - a * eps_r / booth_permittivity(E, eps_r)
    lo, hi = a, a * eps_r / 1.33 ** 2 * 1.0000001
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if f(mid) > 0:
            hi = mid
        else:
            lo = mid
    E = 0.5 * (lo + hi)
    return sgn * E, booth_permittivity(E, eps_r)
