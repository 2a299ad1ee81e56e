"""GPU parity of the physical mode (PNP_METHOD_NEWTON, catint_amd/csrc/pnp_newton.hip) against the CPU oracle
oracle/pnp_physical.py, through the C-ABI.  The oracle is pinned by analytic known answers (tests/test_physical_oracle.py);
parity with the reference's COMSOL path itself is unpinned (no fixtures exist).

Tolerance: both sides iterate Newton to a scaled update < 1e-10 with the same damping rules, so the converged states agree
to ~1e-9 relative whatever the linear solver (LAPACK banded LU on the CPU, block cyclic reduction on the GPU).
"""
import os

import numpy as np
import pytest

from catint_amd import _capi
from oracle import pnp_physical as PH

pytestmark = pytest.mark.gpu

F = 96485.33289
BETA = 1.0 / (8.3144598 * 298.14)
EPS = 78.36 * 8.854187817e-12

SPECIES = [  # (D, z)
    (1.957e-9, 1), (1.185e-9, -1), (2.032e-9, -1), (1.334e-9, 1), (0.923e-9, -2), (5.273e-9, -1), (2.06e-9, 1), (1.792e-9, -1),
]


def make_lanes(N, nx, B, seed, phi_lo=-0.15, phi_hi=0.15, points_per_debye=6.0, cref=10.0):
    rng = np.random.default_rng(seed)
    D = np.array([SPECIES[k][0] for k in range(N)])
    q = np.array([SPECIES[k][1] * F for k in range(N)])
    cb = np.exp(rng.uniform(np.log(0.3 * cref), np.log(3 * cref), (B, N)))
    if N >= 2:  # first species closes charge neutrality (transport.py:757-765); keep it positive by flipping signs if needed
        rest = (cb[:, 1:] * q[None, 1:]).sum(axis=1)
        need = -rest / q[0]
        bad = need <= 0
        cb[:, 0] = np.where(bad, cb[:, 0], need)
        # lanes that cannot be neutralised by species 0 get a compensating amount of the first opposite species
        if np.any(bad):
            opp = [k for k in range(1, N) if q[k] * q[0] < 0][0]
            excess = (cb[bad] * q[None, :]).sum(axis=1)
            cb[bad, opp] += -excess / q[opp]
            assert np.all(cb[bad, opp] > 0)
    lam = np.sqrt(EPS / BETA / max((q ** 2 * cref).sum(), 1.0))
    dx = lam / points_per_debye
    phiM = rng.uniform(phi_lo, phi_hi, B)
    return D, q, cb, dx, phiM


def run_both(N, nx, B, seed, dt=None, nsteps=1, stationary=True, newton_kw=None, flux=None, reactions=None, wall_kinetics=None, x=None,
             velocity=0.0, **lane_kw):
    newton_kw = dict(newton_kw or {})
    D, q, cb, dx, phiM = make_lanes(N, nx, B, seed, **lane_kw)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    fl = np.zeros((B, N)) if flux is None else flux
    s = _capi.PnpSolver(N, nx, dx, dt if dt else 1.0, BETA, EPS, D, q, method='Newton', pb_mode=_capi.PB_DD, batch_capacity=B)
    s.set_newton(**newton_kw)
    if x is not None:
        x = x * dx            # given in units of dx
        s.set_grid(x)
    if reactions:
        s.set_reactions([(r['lhs'], r['rhs'], r['kf'], r['kr']) for r in reactions])
    if velocity:
        s.set_convection(velocity)
    s.set_batch(c0, pb, np.zeros(B), fl)
    if wall_kinetics:
        law = any('alpha' in w or 'saturation' in w for w in wall_kinetics)
        s.set_wall_kinetics([w['species'] for w in wall_kinetics], [w['nu'] for w in wall_kinetics],
                            np.array([w['k'] for w in wall_kinetics]).T,
                            [w.get('alpha', 0.0) for w in wall_kinetics] if law else None,
                            [w.get('saturation', 0.0) for w in wall_kinetics] if law else None)
    if stationary:
        st = s.solve_stationary()
    else:
        s.step(nsteps)
        st = s.get_status()
    c, phi, _, _ = s.get_state()
    its = s.newton_iterations()
    s.close()
    okw = dict(tol=newton_kw.get('tol', 1e-10), maxit=newton_kw.get('maxit', 50), dphi_max=newton_kw.get('dphi_max', 0.05),
               estimate=bool(newton_kw.get('error_estimate', False)))
    if okw['dphi_max'] <= 0:
        okw['dphi_max'] = None
    ref_c = np.zeros_like(c); ref_phi = np.zeros_like(phi); ref_it = np.zeros(B, int)
    for b in range(B):
        p = PH.PhysicalProblem(D=D, charges=q, beta=BETA, eps=EPS, dx=dx, nx=nx, c_bulk=cb[b], phiM=phiM[b], flux=fl[b],
                               stern_capacitance=newton_kw.get('stern_capacitance') if newton_kw.get('wall_bc') == 'stern' else None,
                               phi_pzc=newton_kw.get('phi_pzc', 0.0), mpb_radius=newton_kw.get('mpb_radius'), reactions=reactions,
                               wall_kinetics=[dict(w, k=w['k'][b]) for w in (wall_kinetics or [])], x=x, velocity=velocity)
        cc, ph = c0[b].copy(), np.zeros(nx)
        if stationary:
            cc, ph, it, _ = PH.newton_step(p, cc, ph, cc, np.inf, **okw)
            ref_it[b] = it
        else:
            cc, ph, its_b = PH.integrate(p, cc, ph, dt, nsteps, bdf2=newton_kw.get('time_order', 1) == 2,
                                         predictor=bool(newton_kw.get('predictor', False)), **okw)
            ref_it[b] = sum(its_b)
        ref_c[b], ref_phi[b] = cc, ph
    return (c, phi, its, st), (ref_c, ref_phi, ref_it)


def run_gpu_only(N, nx, B, seed):
    D, q, cb, dx, phiM = make_lanes(N, nx, B, seed)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    with _capi.PnpSolver(N, nx, dx, 1.0, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
        s.set_newton()
        s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
        s.solve_stationary()
        c, phi, _, _ = s.get_state()
        return c, phi, s.newton_iterations()


def assert_close(got, ref, rtol=2e-9):
    c, phi, its, st = got
    rc, rphi, rit = ref
    assert np.all(st == 0), st
    cscale = np.abs(rc).max(axis=2, keepdims=True)
    assert np.abs(c - rc).max() <= rtol * cscale.max() and (np.abs(c - rc) / (np.abs(rc) + 1e-3 * cscale)).max() < 1e-6
    assert np.abs(phi - rphi).max() <= rtol * max(np.abs(rphi).max(), 0.025)
    assert np.array_equal(its, rit), (its, rit)


@pytest.mark.parametrize("N,nx", [(2, 64), (2, 201), (3, 128), (3, 512), (1, 33), (4, 100), (5, 70), (6, 96), (7, 50), (8, 40)])
def test_stationary_matches_oracle(N, nx):
    got, ref = run_both(N, nx, B=5, seed=N * 1000 + nx)
    assert_close(got, ref)


@pytest.mark.parametrize("N,nx", [(2, 64), (3, 130), (4, 77), (6, 40)])
def test_row_per_thread_kernel(N, nx, monkeypatch):
    # the general kernel (rows exchanged through device memory or LDS ping-pong buffers) on shapes the pair kernel
    # would otherwise take
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', 'generic')
    got, ref = run_both(N, nx, B=4, seed=N * 77 + nx)
    assert_close(got, ref)


@pytest.mark.parametrize("N,nx,blocks", [(2, 64, 0), (3, 130, 0), (4, 77, 1), (5, 70, 0), (6, 96, 1), (7, 50, 2), (8, 40, 1)])
def test_sweep_kernel(N, nx, blocks, monkeypatch):
    # block Thomas with one lane team per operating point (the large-batch kernel for large blocks), forced onto small batches;
    # blocks > 0 caps its workgroups so that every team walks several operating points one after the other, with teams of a
    # wave running out of lanes at different times
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', 'sweep')
    if blocks:
        monkeypatch.setenv('CATINT_NEWTON_SWEEP_BLOCKS', str(blocks))
    kw = {'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * N} if N >= 6 else {}
    got, ref = run_both(N, nx, B=23, seed=N * 31 + nx, newton_kw=kw)
    assert_close(got, ref)


@pytest.mark.parametrize("N,nx,blocks", [(5, 70, 0), (5, 71, 2), (6, 96, 1), (6, 9, 0), (7, 50, 2), (7, 51, 0), (8, 40, 1), (8, 129, 0)])
def test_two_sided_sweep_kernel(N, nx, blocks, monkeypatch):
    # elimination from both ends (two lane teams per operating point, the middle row joins them): even and odd row counts (the upper
    # team has one row less when nx is even), very short grids, several operating points per pair one after the other
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', 'both')
    if blocks:
        monkeypatch.setenv('CATINT_NEWTON_SWEEP_BLOCKS', str(blocks))
    kw = {'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * N} if N >= 6 else {}
    got, ref = run_both(N, nx, B=23, seed=N * 31 + nx, newton_kw=kw)
    assert_close(got, ref)


def test_two_sided_sweep_with_reactions_wall_kinetics_and_transient_steps(monkeypatch):
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', 'both')
    rx = [{'lhs': [1], 'rhs': [2], 'kf': 4e5, 'kr': 9e5}, {'lhs': [0, 2, 2], 'rhs': [4, 5], 'kf': 5.0, 'kr': 1e2}]
    got, ref = run_both(6, 64, B=11, seed=41, reactions=rx, dt=1e-7, nsteps=3, stationary=False)
    assert_close(got, ref)
    B = 9
    rng = np.random.default_rng(3)
    wk = [{'species': 2, 'k': rng.uniform(0.05, 1.0, B), 'nu': [0.0, 0.0, -1.0, 1.0, 0.0, 0.0], 'alpha': -6.0, 'saturation': 0.05},
          {'species': -1, 'k': rng.uniform(1e-6, 1e-5, B), 'nu': [0.0, 1.0, 0.0, 0.0, 0.0, 0.0], 'alpha': -4.0}]
    got, ref = run_both(6, 96, B=B, seed=29, wall_kinetics=wk,
                        newton_kw=dict(wall_bc='stern', stern_capacitance=0.2, phi_pzc=0.05, mpb_radius=[4.1e-10] + [0.0] * 5))
    assert_close(got, ref)


def test_two_sided_sweep_equals_the_one_sided_sweep(monkeypatch):
    """Same linear systems, different elimination order: states to rounding, identical iteration counts, lane by lane."""
    outs = []
    for kern in ('sweep', 'both'):
        monkeypatch.setenv('CATINT_NEWTON_KERNEL', kern)
        got, _ = run_both(8, 128, B=40, seed=5, newton_kw={'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * 8})
        outs.append(got)
    (c1, p1, it1, st1), (c2, p2, it2, st2) = outs
    assert np.array_equal(it1, it2) and np.array_equal(st1, st2)
    assert np.abs(c1 - c2).max() <= 1e-10 * np.abs(c1).max() and np.abs(p1 - p2).max() <= 1e-11


def test_kernel_choice_for_large_batches_of_large_blocks(monkeypatch):
    # N >= 5 species: 896 <= B < 10 240 operating points take the lane-quad kernel (pnp_lane4.hip), up to 14 336 the lane pair, larger batches the lane kernel
    # (pnp_lane.hip; measured crossovers, profiles/r03_lane_sweep.jsonl, profiles/r04_lane4_probe.jsonl): the default equals the forced
    # kernel bit for bit; the sweep kernels (one lane team per operating point, one- and two-sided) stay selectable and give the same
    # answers and iteration counts as the lane-team kernel
    N, nx, B = 5, 24, 16384
    monkeypatch.delenv('CATINT_NEWTON_KERNEL', raising=False)
    q0 = run_gpu_only(N, nx, 2048, 4)
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', 'lane4')
    q1 = run_gpu_only(N, nx, 2048, 4)
    assert np.array_equal(q0[0], q1[0]) and np.array_equal(q0[1], q1[1]) and np.array_equal(q0[2], q1[2])
    monkeypatch.delenv('CATINT_NEWTON_KERNEL', raising=False)
    a = run_gpu_only(N, nx, B, 5)
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', 'lane')
    forced = run_gpu_only(N, nx, B, 5)
    assert np.array_equal(a[0], forced[0]) and np.array_equal(a[1], forced[1]) and np.array_equal(a[2], forced[2])
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', 'team')
    b = run_gpu_only(N, nx, B, 5)
    assert np.abs(a[0] - b[0]).max() <= 1e-9 * np.abs(b[0]).max() and np.abs(a[1] - b[1]).max() <= 1e-10
    assert np.array_equal(a[2], b[2]) and (a[2] <= 50).all()
    assert not np.array_equal(a[0], b[0])            # (two different linear solvers: not the same bits)
    for kern in ('sweep', 'both'):
        monkeypatch.setenv('CATINT_NEWTON_KERNEL', kern)
        c = run_gpu_only(N, nx, B // 2, 6)
        monkeypatch.setenv('CATINT_NEWTON_KERNEL', 'team')
        d = run_gpu_only(N, nx, B // 2, 6)
        assert np.abs(c[0] - d[0]).max() <= 1e-9 * np.abs(d[0]).max() and np.array_equal(c[2], d[2])
        assert not np.array_equal(c[0], d[0])
    # below the crossover the lane-team kernel stays the choice
    monkeypatch.delenv('CATINT_NEWTON_KERNEL')
    e = run_gpu_only(N, nx, 640, 7)
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', 'team')
    f = run_gpu_only(N, nx, 640, 7)
    assert np.array_equal(e[0], f[0]) and np.array_equal(e[2], f[2])


@pytest.mark.parametrize("kernel,B", [('lane4', 1536), ('lane', 2100), ('', 40)])
def test_tune_placement_keeps_the_trajectory_and_the_memory(kernel, B, monkeypatch):
    """pnp_tune_placement: the lane kernels' workspace is allocated several times, a few timesteps are timed on each placement from the
    handle's state, the fastest placement stays and the others are freed.  Where the workspace lies changes no number: state, history,
    status and iteration counts are as before the call, the steps after it continue the trajectory to the bit, and the handle holds as
    much device memory as before.  A batch that no lane kernel takes is left alone."""
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', kernel)
    N, nx = 6, 64
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 13)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    dt = 0.2 * (6 * dx) * (nx * dx) / D.max()
    kw = dict(wall_bc='stern', stern_capacitance=0.25, mpb_radius=[3.5e-10] * N, time_order=2)

    def run(tune):
        with _capi.PnpSolver(N, nx, dx, dt, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
            s.set_newton(**kw)
            s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
            s.step(2)
            if tune:
                before = (s.get_state()[0].copy(), s.get_state()[1].copy(), s.newton_iterations().copy(), s.device_bytes)
                ms = s.tune_placement(2, 3)
                assert len(ms) == (3 if kernel else 0) and all(v > 0 for v in ms)
                assert np.array_equal(before[0], s.get_state()[0]) and np.array_equal(before[1], s.get_state()[1])
                assert np.array_equal(before[2], s.newton_iterations()) and s.device_bytes <= before[3] + 16 * B
            s.step(3)
            assert (s.get_status() == 0).all()
            return s.get_state()[0], s.get_state()[1], s.newton_iterations()
    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("N,nx,B,family", [
    (8, 16, 512, 'workgroup'), (8, 16, 1024, 'lane4'), (8, 16, 8192, 'lane4'), (8, 16, 10240, 'lane2'), (8, 16, 16384, 'lane2'),
    (8, 16, 16385, 'lane+fused'), (6, 16, 13311, 'lane2'), (6, 16, 13312, 'lane+fused'), (7, 16, 13311, 'lane2'), (7, 16, 13312, 'lane+fused'), (5, 16, 10239, 'lane4'), (5, 16, 10240, 'lane+fused'),
    (4, 16, 6143, 'workgroup'), (4, 16, 6144, 'lane+fused'), (3, 16, 15359, 'workgroup'), (3, 16, 15360, 'lane+fused'),
    (2, 16, 28671, 'workgroup'), (2, 16, 28672, 'lane+fused'), (2, 2100, 4096, 'lane+fused'), (2, 2100, 4095, 'workgroup'),
])
def test_default_family_follows_the_measured_thresholds(N, nx, B, family, monkeypatch):
    """pnp_autotune_default reports the kernel family the library's thresholds choose for a batch -- the table of
    profiles/r04_family_rates*.jsonl as the selection functions state it (pnp_lane.hip, pnp_lane4.hip): lane quad from 896 to 10 239
    points, lane pair up to 16 384 at N = 8 (13 311 at N = 6, 7; none at N = 5), then the fused lane kernel; small blocks go to the lane
    kernel where it overtakes the pair kernel / the lane teams."""
    monkeypatch.delenv('CATINT_NEWTON_KERNEL', raising=False)
    D, q, cb, dx, phiM = make_lanes(N, nx, 2, 5)
    with _capi.PnpSolver(N, nx, dx, 1e-9, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
        assert s.default_family() is None                      # no batch yet
        s.set_newton(**({'mpb_radius': [3.5e-10] * N} if N >= 5 else {}))
        s.set_batch(np.repeat(np.repeat(cb[:1], B, axis=0)[:, :, None], nx, axis=2), np.zeros((B, 4)), np.zeros(B), np.zeros((B, N)))
        assert s.default_family() == family


@pytest.mark.parametrize("time_order", [1, 2])
def test_autotune_measures_every_family_and_leaves_the_trajectory_untouched(time_order, monkeypatch):
    """pnp_autotune: every kernel family that supports the shape is timed on the handle's own batch (one warm-up step + nsteps from the
    current state), the fastest becomes the handle's option, and state, BDF2 history, status and iteration counts are put back after
    every trial -- steps taken after the call continue the trajectory exactly as if the chosen family had been forced from the start."""
    monkeypatch.delenv('CATINT_NEWTON_KERNEL', raising=False)
    N, nx, B = 6, 96, 1536
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 11)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    dt = 0.2 * (6 * dx) * (nx * dx) / D.max()
    kw = dict(wall_bc='stern', stern_capacitance=0.25, mpb_radius=[3.5e-10] * N, time_order=time_order, tol=1e-10)

    def solver(option=None):
        s = _capi.PnpSolver(N, nx, dx, dt, BETA, EPS, D, q, method='Newton', batch_capacity=B)
        s.set_newton(**kw)
        for k, v in (option or {}).items():
            s.set_option(k, v)
        s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
        return s
    with solver() as s:
        s.step(2)
        before = (s.get_state()[0].copy(), s.get_state()[1].copy(), s.newton_iterations().copy(), s.get_status().copy())
        name, ms = s.autotune(2)
        after = (s.get_state()[0], s.get_state()[1], s.newton_iterations(), s.get_status())
        for a, b in zip(before, after):
            assert np.array_equal(a, b)
        assert set(ms) == {'lane4', 'lane2', 'lane', 'lane+fused', 'workgroup', 'team', 'sweep', 'both'} and name in ms
        assert all(v > 0 for v in ms.values()) and ms[name] == min(ms.values())
        held = s.device_bytes
        s.step(3)
        tuned = (s.get_state()[0], s.get_state()[1], s.newton_iterations())
        assert (s.get_status() == 0).all()
    forced = {'lane+fused': {'NEWTON_KERNEL': 'lane', 'LANE_FUSED': '1'}, 'lane': {'NEWTON_KERNEL': 'lane', 'LANE_FUSED': '0'}}.get(name, {'NEWTON_KERNEL': name})
    with solver() as s:                       # the same trajectory with the family forced for its second part
        s.step(2)
        for k, v in forced.items():
            s.set_option(k, v)
        s.step(3)
        ref = (s.get_state()[0], s.get_state()[1], s.newton_iterations())
        assert held <= s.device_bytes + 16 * B        # the losers' workspaces went back to the device (a per-lane counter array may stay)
    for a, b in zip(tuned, ref):
        assert np.array_equal(a, b)
    with _capi.PnpSolver(3, 64, 1e-10, 1e-9, BETA, EPS, D[:3], q[:3], batch_capacity=4) as s:       # compat handle: refused
        with pytest.raises(_capi.PnpError):
            s.autotune()


@pytest.mark.parametrize("N,nx", [(3, 513), (2, 1030), (3, 1200)])
def test_more_rows_than_threads(N, nx):
    got, ref = run_both(N, nx, B=3, seed=nx, points_per_debye=20.0)
    assert_close(got, ref)


@pytest.mark.parametrize("N,nx", [(3, 512), (4, 100), (6, 96), (7, 50), (8, 40)])
def test_bitwise_reproducible(N, nx):
    a = run_gpu_only(N, nx, 4, 99)
    b = run_gpu_only(N, nx, 4, 99)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_transient_steps_match_oracle():
    N, nx = 3, 160
    D, q, cb, dx, phiM = make_lanes(N, nx, 4, 7)
    dt = 0.2 * (6 * dx) * (nx * dx) / D.max()
    got, ref = run_both(N, nx, B=4, seed=7, dt=dt, nsteps=6, stationary=False)
    assert_close(got, ref)
    assert got[2].min() >= 6 * 2          # at least two Newton iterations per step


def test_stern_layer_and_steric_ions():
    a = [4.1e-10, 3.1e-10, 3.5e-10]
    got, ref = run_both(3, 240, B=6, seed=11, phi_lo=-1.5, phi_hi=1.0, cref=100.0,
                        newton_kw=dict(wall_bc='stern', stern_capacitance=0.2, phi_pzc=0.05, mpb_radius=a, maxit=60))
    assert_close(got, ref)
    c = got[0]
    assert (c[:, 0, 0] * PH.N_AVOGADRO * a[0] ** 3).max() < 1.0


def test_wall_fluxes():
    rng = np.random.default_rng(5)
    B, N = 4, 3
    flux = rng.uniform(-2e-4, 2e-4, (B, N))
    got, ref = run_both(N, 96, B=B, seed=13, flux=flux)
    assert_close(got, ref)


def test_homogeneous_reactions_point_ions():
    # K+ / HCO3- / CO3^2- style buffer: species 1 <-> species 4 exchange plus a dimerisation, bulk out of equilibrium
    rx = [{'lhs': [1], 'rhs': [2], 'kf': 4e5, 'kr': 9e5}, {'lhs': [2, 2], 'rhs': [3], 'kf': 3e3, 'kr': 0.0},
          {'lhs': [0, 1], 'rhs': [3], 'kf': 1e3, 'kr': 2e4}, {'lhs': [], 'rhs': [0, 1], 'kf': 2e3, 'kr': 1.5e2}]   # last: H2O <-> A+ + B-
    got, ref = run_both(4, 120, B=4, seed=31, reactions=rx, points_per_debye=2.0)
    assert_close(got, ref)


def test_homogeneous_reactions_with_steric_ions_and_stern_layer():
    rx = [{'lhs': [1], 'rhs': [2], 'kf': 4e5, 'kr': 9e5}, {'lhs': [0, 2], 'rhs': [1], 'kf': 2e3, 'kr': 1e4}]
    got, ref = run_both(3, 200, B=5, seed=37, reactions=rx, phi_lo=-1.0, phi_hi=0.8, cref=100.0,
                        newton_kw=dict(wall_bc='stern', stern_capacitance=0.2, mpb_radius=[4.1e-10, 3e-10, 0.0], maxit=60))
    assert_close(got, ref)


def test_reactions_in_the_row_per_thread_kernel(monkeypatch):
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', 'generic')
    rx = [{'lhs': [1], 'rhs': [2], 'kf': 4e5, 'kr': 9e5}, {'lhs': [0, 2, 2], 'rhs': [4, 5], 'kf': 5.0, 'kr': 1e2}]
    got, ref = run_both(6, 64, B=3, seed=41, reactions=rx, dt=1e-7, nsteps=3, stationary=False)
    assert_close(got, ref)


def test_implicit_wall_kinetics():
    B, N = 5, 4
    rng = np.random.default_rng(9)
    wk = [{'species': 2, 'k': rng.uniform(0.05, 5.0, B), 'nu': [0.0, 0.0, -1.0, 1.0]},
          {'species': 0, 'k': rng.uniform(1e-3, 1e-2, B), 'nu': [-1.0, 0.0, 0.5, 0.0]},
          {'species': -1, 'k': rng.uniform(1e-6, 1e-5, B), 'nu': [0.0, 1.0, 0.0, 0.0]}]
    got, ref = run_both(N, 150, B=B, seed=17, wall_kinetics=wk,
                        newton_kw=dict(wall_bc='stern', stern_capacitance=0.2, mpb_radius=[4.1e-10, 0.0, 0.0, 0.0]))
    assert_close(got, ref)


@pytest.mark.parametrize("N,nx,B,kernel", [(4, 150, 5, None), (3, 128, 4, None), (6, 96, 4, None), (8, 64, 3, None), (4, 150, 5, 'generic'),
                                           (6, 96, 70, 'sweep')])
def test_butler_volmer_and_langmuir_wall_kinetics(N, nx, B, kernel, monkeypatch):
    """pnp_set_wall_rate_law: rate = K c/(1 + K_sat c) exp(alpha (phiM - phi(0))) inside the Newton system (both derivatives), Stern
    wall so that phi(0) is an unknown -- against the oracle, in every kernel family (pair/row-per-thread N <= 4, lane teams N >= 5,
    sweep)."""
    if kernel:
        monkeypatch.setenv('CATINT_NEWTON_KERNEL', kernel)
    rng = np.random.default_rng(N * 100 + nx)
    nu = lambda **kw: [kw.get(k, 0.0) for k in range(N)]      # noqa: E731
    wk = [{'species': 2, 'k': rng.uniform(0.05, 1.0, B), 'nu': [0.0, 0.0, -1.0] + [1.0] * (N > 3) + [0.0] * max(N - 4, 0), 'alpha': -6.0,
           'saturation': 0.05},
          {'species': 0, 'k': rng.uniform(1e-4, 1e-3, B), 'nu': [-1.0, 0.0, 0.5] + [0.0] * (N - 3), 'alpha': 3.0},
          {'species': -1, 'k': rng.uniform(1e-6, 1e-5, B), 'nu': [0.0, 1.0, 0.0] + [0.0] * (N - 3), 'alpha': -4.0},
          {'species': 1, 'k': rng.uniform(1e-4, 1e-3, B), 'nu': [0.0, -1.0, 0.0] + [0.0] * (N - 3), 'saturation': 0.2}]
    got, ref = run_both(N, nx, B=B, seed=23 + N, wall_kinetics=wk,
                        newton_kw=dict(wall_bc='stern', stern_capacitance=0.2, phi_pzc=0.05, mpb_radius=[4.1e-10] + [0.0] * (N - 1)))
    assert_close(got, ref)
    # the potential dependence is live: the same table without alpha gives a different surface state
    got0, _ = run_both(N, nx, B=B, seed=23 + N, wall_kinetics=[dict(w, alpha=0.0) for w in wk],
                       newton_kw=dict(wall_bc='stern', stern_capacitance=0.2, phi_pzc=0.05, mpb_radius=[4.1e-10] + [0.0] * (N - 1)))
    assert np.abs(got[0][:, 2, 0] - got0[0][:, 2, 0]).max() > 1e-3 * np.abs(got0[0][:, 2, 0]).max()


def test_rate_law_argument_checks():
    D, q, cb, dx, phiM = make_lanes(3, 32, 2, 1)
    with _capi.PnpSolver(3, 32, dx, 1.0, BETA, EPS, D, q, method='Newton', batch_capacity=2) as s:
        s.set_newton()
        s.set_batch(np.repeat(cb[:, :, None], 32, axis=2), np.zeros((2, 4)), np.zeros(2), np.zeros((2, 3)))
        with pytest.raises(_capi.PnpError, match='n differs'):
            s._check(s._lib.pnp_set_wall_rate_law(s._h, 1, None, None))
        with pytest.raises(_capi.PnpError, match='saturation'):
            s.set_wall_kinetics([-1], [[0.0, 1.0, 0.0]], np.ones((2, 1)), None, [0.5])
        with pytest.raises(_capi.PnpError, match='saturation'):
            s.set_wall_kinetics([1], [[0.0, 1.0, 0.0]], np.ones((2, 1)), None, [-0.5])


@pytest.mark.parametrize("N,nx,kw", [(3, 96, {}), (3, 257, dict(wall_bc='stern', stern_capacitance=0.2, mpb_radius=[4.1e-10, 3e-10, 0.0])),
                                     (6, 80, {})])
def test_graded_grid(N, nx, kw, monkeypatch):
    from catint_amd.host import graded_mesh
    x = graded_mesh(4000.0, 1.0, nx)              # first cell = dx, domain 4000 dx, growth ratio ~1.03-1.1
    rx = [{'lhs': [1], 'rhs': [2], 'kf': 4e5, 'kr': 9e5}]
    wk = [{'species': 2, 'k': np.full(3, 0.05), 'nu': [0.0, 0.0, -1.0] + [0.0] * (N - 3)}]
    got, ref = run_both(N, nx, B=3, seed=nx, x=x, reactions=rx, wall_kinetics=wk, newton_kw=kw, points_per_debye=8.0)
    assert_close(got, ref)
    # transient on the same grid
    got, ref = run_both(N, nx, B=3, seed=nx + 1, x=x, newton_kw=kw, points_per_debye=8.0, dt=1e-7, nsteps=3, stationary=False)
    assert_close(got, ref)


@pytest.mark.parametrize("N", [2, 3, 4, 5, 6, 7, 8])
def test_zero_rate_reaction_table_changes_nothing(N, monkeypatch):
    """The reaction-enabled kernel variant with an all-zero rate table must reproduce the plain variant bit for bit, in both
    kernel families (guards the instantiation-specific miscompile seen for N = 7)."""
    for kern in ('', 'generic'):
        monkeypatch.setenv('CATINT_NEWTON_KERNEL', kern)
        D, q, cb, dx, phiM = make_lanes(N, 72, 3, 5)
        c0 = np.repeat(cb[:, :, None], 72, axis=2)
        pb = np.zeros((3, 4)); pb[:, 0] = phiM
        out = []
        for table in (None, [([0, min(1, N - 1)], [0], 0.0, 0.0)]):
            with _capi.PnpSolver(N, 72, dx, 1.0, BETA, EPS, D, q, method='Newton', batch_capacity=3) as s:
                s.set_newton(mpb_radius=[4.1e-10] + [0.0] * (N - 1))
                if table:
                    s.set_reactions(table)
                s.set_batch(c0, pb, np.zeros(3), np.zeros((3, N)))
                st = s.solve_stationary()
                out.append((s.get_state()[0], s.newton_iterations(), st))
        assert np.all(out[0][2] == 0) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])


def test_integrate_entry_point_writes_outputs_at_itout():
    N, nx, B = 3, 96, 3
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 23)
    dt = 5e-9
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4)); pb[:, 0] = phiM
    itout = [2, 5, 7]
    with _capi.PnpSolver(N, nx, dx, dt, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
        s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
        cout, st = s.integrate(8, itout)              # loop n = 1..7: the state after n backward-Euler steps is t_n
    assert np.all(st == 0) and cout.shape == (3, B, N * nx)
    for b in range(B):
        p = PH.PhysicalProblem(D=D, charges=q, beta=BETA, eps=EPS, dx=dx, nx=nx, c_bulk=cb[b], phiM=phiM[b])
        cc, ph = c0[b].copy(), np.zeros(nx)
        k = 0
        for n in range(1, 8):
            cc, ph, _, _ = PH.newton_step(p, cc, ph, cc, dt)
            if n in itout:
                assert np.abs(cout[k, b].reshape(N, nx) - cc).max() <= 2e-9 * np.abs(cc).max()
                k += 1


def test_config2_batch_physical_mode_full_size_properties():
    """BASELINE configs[1] sizes (batch 1024 x 3 species x 512 points) in the physical mode: every lane converges, sampled
    lanes equal the oracle, and a lane's result does not depend on where in the batch (or in which batch) it is solved."""
    from catint_amd.synthetic import make_batch
    B, N, nx = 1024, 3, 512
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=5, phi_max=0.2, dt_factor=0.1)
    pb = np.nan_to_num(pb)

    def solve(idx, steps):
        with _capi.PnpSolver(N, nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton', batch_capacity=len(idx)) as s:
            s.set_newton(tol=1e-9)
            s.set_batch(c0[idx], pb[idx], vz[idx], fl[idx])
            s.step(steps)
            c, phi, _, _ = s.get_state()
            return c, phi, s.newton_iterations(), s.get_status()
    c, phi, its, st = solve(np.arange(B), 4)
    assert np.all(st == 0) and np.all(np.isfinite(c)) and c.min() > 0 and its.min() >= 8
    perm = np.random.default_rng(0).permutation(B)
    c2, phi2, its2, _ = solve(perm, 4)
    assert np.array_equal(c2, c[perm]) and np.array_equal(phi2, phi[perm]) and np.array_equal(its2, its[perm])
    sub = np.array([3, 500, 1023])
    c3, phi3, its3, _ = solve(sub, 4)
    assert np.array_equal(c3, c[sub]) and np.array_equal(its3, its[sub])
    for b in sub:
        p = PH.PhysicalProblem(D=prob.D, charges=prob.charges, beta=prob.beta, eps=prob.eps, dx=prob.dx, nx=nx,
                               c_bulk=c0[b].reshape(N, nx)[:, -1], phiM=pb[b, 0])
        rc, rphi, rit = PH.integrate(p, c0[b].reshape(N, nx), np.zeros(nx), prob.dt, 4, tol=1e-9)
        assert sum(rit) == its[b]
        assert np.abs(c[b] - rc).max() <= 2e-9 * np.abs(rc).max() and np.abs(phi[b] - rphi).max() <= 2e-9 * 0.2


def test_config4_per_gpu_share_through_the_sweep_kernel():
    """BASELINE configs[3] (6 species x 1024 points, 262144 lanes over 8 GPUs): one GPU's 32768 lanes, size-modified with a Stern
    wall, two implicit timesteps -- the batch takes the sweep kernel.  Every lane converges, sampled lanes equal the oracle, and
    a lane does not depend on its position in the batch (teams pick lanes up in a different order)."""
    from catint_amd.synthetic import make_batch
    B, N, nx = 32768, 6, 1024
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=6, phi_max=0.2, dt_factor=0.1)
    pb = np.nan_to_num(pb)
    radii = [4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 0.0, 3e-10]
    kw = dict(wall_bc='stern', stern_capacitance=0.2, tol=1e-9, mpb_radius=radii)

    def solve(idx, steps):
        with _capi.PnpSolver(N, nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton', batch_capacity=len(idx)) as s:
            s.set_newton(**kw)
            s.set_batch(c0[idx], pb[idx], vz[idx], fl[idx])
            s.step(steps)
            cs, vs, es = s.get_surface()
            return cs, vs, s.newton_iterations(), s.get_status(), (s.get_state() if len(idx) <= 64 else None)
    cs, vs, its, st, _ = solve(np.arange(B), 2)
    assert np.all(st == 0) and np.all(np.isfinite(cs)) and cs.min() > 0 and its.min() >= 4 and its.max() <= 2 * 50
    sub = np.array([0, 7, 12345, 32767])
    cs3, vs3, its3, st3, full = solve(np.concatenate([sub, np.arange(100, 160)]), 2)      # 64 lanes: the lane-team kernel
    assert np.array_equal(its3[:4], its[sub])
    assert np.abs(cs3[:4] - cs[sub]).max() <= 1e-9 * np.abs(cs).max() and np.abs(vs3[:4] - vs[sub]).max() <= 1e-10
    c, phi = full[0], full[1]
    for j, b in enumerate(sub[:2]):
        p = PH.PhysicalProblem(D=prob.D, charges=prob.charges, beta=prob.beta, eps=prob.eps, dx=prob.dx, nx=nx,
                               c_bulk=c0[b].reshape(N, nx)[:, -1], phiM=pb[b, 0], stern_capacitance=0.2, mpb_radius=radii)
        rc, rphi, rit = PH.integrate(p, c0[b].reshape(N, nx), np.zeros(nx), prob.dt, 2, tol=1e-9)
        assert sum(rit) == its[b]
        assert np.abs(c[j] - rc).max() <= 2e-9 * np.abs(rc).max() and np.abs(phi[j] - rphi).max() <= 2e-9 * 0.2
        assert np.abs(cs[b] - rc[:, 0]).max() <= 2e-9 * np.abs(rc).max()


def test_config5_shape_size_modified_eight_species_4096_points():
    """BASELINE configs[4]: 8-species size-modified PNP on 4096 grid points (lane-team kernel), two lanes against the oracle."""
    a = [4.1e-10, 3.0e-10, 3.3e-10, 3.6e-10, 0.0, 3.5e-10, 3.2e-10, 0.0]
    got, ref = run_both(8, 4096, B=2, seed=4096, points_per_debye=40.0, phi_lo=-0.9, phi_hi=-0.5, cref=50.0,
                        newton_kw=dict(wall_bc='stern', stern_capacitance=0.2, mpb_radius=a, maxit=60))
    assert_close(got, ref)


@pytest.mark.parametrize("N,nx", [(3, 200), (6, 90), (3, 1100)])
def test_error_estimate_termination(N, nx):
    """pnp_newton_params.error_estimate: same acceptance rule on both sides -> same iteration counts, fewer than strict."""
    D, q, cb, dx, phiM = make_lanes(N, nx, 4, 77)
    dt = 0.2 * (6 * dx) * (nx * dx) / D.max()
    strict, _ = run_both(N, nx, B=4, seed=77, dt=dt, nsteps=8, stationary=False, newton_kw=dict(tol=1e-8))
    got, ref = run_both(N, nx, B=4, seed=77, dt=dt, nsteps=8, stationary=False, newton_kw=dict(tol=1e-8, error_estimate=True))
    assert np.array_equal(got[2], ref[2]) and np.all(got[3] == 0)
    assert got[2].sum() <= strict[2].sum()
    assert np.abs(got[0] - ref[0]).max() <= 1e-7 * np.abs(ref[0]).max()
    assert np.abs(got[0] - strict[0]).max() <= 1e-6 * np.abs(strict[0]).max()


def test_not_converged_is_reported():
    got, ref = run_both(2, 64, B=3, seed=3, newton_kw=dict(maxit=2))
    assert np.all(got[3] == _capi.STATUS_MAXIT)
    assert np.array_equal(got[2], ref[2]) and np.all(got[2] == 3)


def test_surface_observables():
    N, nx, B = 2, 128, 3
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 21)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4)); pb[:, 0] = phiM
    with _capi.PnpSolver(N, nx, dx, 1.0, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
        s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
        assert np.all(s.solve_stationary() == 0)
        c, phi, g, l = s.get_state()
        cs, vs, es = s.get_surface()
    assert np.allclose(cs, c[:, :, 0]) and np.allclose(vs, phiM) and np.allclose(es, -(phi[:, 1] - phi[:, 0]) / dx)
    assert np.allclose(l, -(q[None, :, None] * c).sum(axis=1) / EPS)
    assert np.allclose(g[:, 1:-1], (phi[:, 2:] - phi[:, :-2]) / (2 * dx))


def test_randomised_configurations_follow_the_oracle():
    """tests/fuzz/fuzz_newton.py at test size: random species counts, grids, boundary models, reactions, kinetics, fluxes,
    stationary or transient -- converged lanes equal the oracle, every lane takes the oracle's number of Newton iterations."""
    import subprocess
    import sys
    env = dict(os.environ, FUZZ_SEED='3', FUZZ_CASES='30')
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), 'fuzz', 'fuzz_newton.py')], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert '30 cases, 0 bad' in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("N,nx,B", [(5, 70, 5), (6, 96, 5), (6, 300, 300), (5, 200, 64)])
def test_register_budget_does_not_change_results(N, nx, B, monkeypatch):
    """The row-per-thread kernel built for 512 registers per thread (launch bound 256: 256 VGPRs + accumulator registers as spill
    space) against the 256-register build the library uses (scratch instead): same bits, same iteration counts, run to run.  (An early
    version of the kernel was not reproducible at 512 registers -- DESIGN.md section 7; the register budget is an occupancy choice.)"""
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', 'generic')
    monkeypatch.delenv('CATINT_NEWTON_REGS', raising=False)
    a = run_gpu_only(N, nx, B, N * 1000 + nx)
    monkeypatch.setenv('CATINT_NEWTON_REGS', '512')
    b1 = run_gpu_only(N, nx, B, N * 1000 + nx)
    b2 = run_gpu_only(N, nx, B, N * 1000 + nx)
    for b in (b1, b2):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert (a[2] <= 50).all()


@pytest.mark.parametrize("kernel", ['', 'generic', 'team'])
def test_fuzz_case_117_stops_at_the_rounding_floor(kernel, monkeypatch):
    """The outlier of the randomised runs (tests/fuzz/fuzz_newton.py, seed 101, case 117: 3 species, 3 stiff homogeneous reactions, wall
    fluxes, graded 400-point grid, stationary; frozen as tests/golden/fuzz/newton_case117.json).  Newton converges quadratically by
    iteration 17; from there the scaled update floats at ~3e-9 -- cond(J) eps of this Jacobian -- ABOVE the tolerance of 1e-10, and every
    linear solver left the loop when ITS rounding noise happened to dip below it: 20 iterations (pair kernel), 31 (oracle, LAPACK),
    37 (row-per-thread kernel), all with the same state to 4e-9.  With the rounding-floor exit (pnp_math.h: newton_at_rounding_floor;
    oracle/pnp_physical.py) they all stop within an iteration of each other, on the same state."""
    import json
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', kernel)
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fuzz', 'newton_case117.json')))
    got, ref = run_both(d['N'], d['nx'], B=d['B'], seed=d['seed'], newton_kw=d['newton_kw'], reactions=d['reactions'],
                        flux=np.array(d['flux']), x=np.array(d['x']), stationary=True, phi_lo=-0.3, phi_hi=0.3,
                        points_per_debye=d['points_per_debye'])
    c, phi, its, st = got
    rc, rphi, rit = ref
    # (round 4: the exit asks for an update that did NOT shrink -- round 3 accepted any pair whose second update was more than half the
    # first, which also lets a linearly converging lane out; same iteration here)
    assert (st == 0).all() and rit[0] == 20
    assert abs(int(its[0]) - int(rit[0])) <= 1, (its, rit)
    assert np.abs(c - rc).max() <= 1e-8 * np.abs(rc).max() and np.abs(phi - rphi).max() <= 1e-9


# ---- BDF2 timesteps (the reference's transient study: BDF, maxorder 2, comsol_model.py:518-531) -----------------------------------------
@pytest.mark.parametrize("N,nx,B,kernel,kw", [
    (3, 128, 5, '', {}),                                                                                 # pair kernel
    (2, 1100, 3, '', {}),                                                                                # lane teams
    (3, 130, 4, 'generic', {}),                                                                          # row-per-thread kernel
    (6, 96, 9, 'sweep', dict(mpb_radius=[3.5e-10] * 6)),                                                 # one-sided sweep
    (7, 80, 9, 'both', {}),                                                                              # two-sided sweep
    (6, 96, 40, 'team', dict(wall_bc='stern', stern_capacitance=0.25, mpb_radius=[3.5e-10] * 6)),
    (8, 64, 70, 'lane', dict(wall_bc='stern', stern_capacitance=0.25, mpb_radius=[3.5e-10] * 8)),
    (6, 80, 37, 'lane2', {}),
    (8, 64, 37, 'lane4', dict(mpb_radius=[3.5e-10] * 8)),
    (3, 96, 70, 'lane', {}),
])
def test_bdf2_timesteps_match_oracle(N, nx, B, kernel, kw, monkeypatch):
    """pnp_newton_params.time_order = 2: (3 c_n+1 - 4 c_n + c_n-1) / (2 dt), first step backward Euler; every kernel family, states and
    summed iteration counts against the oracle; the result differs from backward Euler's (the option is not a no-op)."""
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', kernel)
    D, q, cb, dx, phiM = make_lanes(N, nx, 4, 7)
    dt = 0.2 * (6 * dx) * (nx * dx) / D.max()
    got, ref = run_both(N, nx, B=B, seed=7, dt=dt, nsteps=5, stationary=False, newton_kw=dict(kw, time_order=2))
    assert_close(got, ref)
    be, _ = run_both(N, nx, B=B, seed=7, dt=dt, nsteps=5, stationary=False, newton_kw=kw)
    assert np.abs(got[0] - be[0]).max() > 1e-6 * np.abs(be[0]).max()


@pytest.mark.parametrize("N,nx,B,kernel,kw", [
    (3, 128, 5, '', {}),                                                                                 # pair kernel
    (2, 1100, 3, '', dict(time_order=2)),                                                                # lane teams, with BDF2
    (3, 130, 4, 'generic', {}),
    (6, 96, 9, 'sweep', dict(mpb_radius=[3.5e-10] * 6)),
    (7, 80, 9, 'both', dict(time_order=2)),
    (6, 96, 40, 'team', dict(wall_bc='stern', stern_capacitance=0.25, mpb_radius=[3.5e-10] * 6)),
    (8, 64, 70, 'lane', dict(wall_bc='stern', stern_capacitance=0.25, mpb_radius=[3.5e-10] * 8, time_order=2)),
    (6, 80, 37, 'lane2', {}),
    (8, 64, 37, 'lane4', dict(mpb_radius=[3.5e-10] * 8)),
])
def test_predictor_matches_oracle_and_saves_iterations(N, nx, B, kernel, kw, monkeypatch):
    """pnp_newton_params.predictor: from the second step on Newton starts from 2 u_n - u_n-1 (the start of a BDF stepper's corrector).
    Same equations and stopping rule: GPU and oracle walk the same iterates (identical counts), the trajectory agrees with the one
    without predictor to the Newton tolerance, and it needs fewer iterations."""
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', kernel)
    D, q, cb, dx, phiM = make_lanes(N, nx, 4, 7)
    dt = 0.05 * (6 * dx) * (nx * dx) / D.max()
    got, ref = run_both(N, nx, B=B, seed=7, dt=dt, nsteps=8, stationary=False, newton_kw=dict(kw, predictor=True, tol=1e-10))
    assert_close(got, ref)
    plain, _ = run_both(N, nx, B=B, seed=7, dt=dt, nsteps=8, stationary=False, newton_kw=dict(kw, tol=1e-10))
    assert np.abs(got[0] - plain[0]).max() <= 1e-7 * np.abs(plain[0]).max() and np.abs(got[1] - plain[1]).max() <= 1e-8
    assert got[2].sum() < plain[2].sum(), (got[2].sum(), plain[2].sum())


@pytest.mark.parametrize("kernel,N,B", [('', 3, 6), ('lane', 6, 70), ('lane2', 6, 37), ('lane4', 6, 21)])
def test_bdf2_steps_split_over_calls_and_second_order_in_time(kernel, N, B, monkeypatch):
    """The BDF2 history (c_n-1) lives on the handle: pnp_step(2) + pnp_step(3) == pnp_step(5) to the bit, iteration counts of a call are
    the sums over its steps.  And the point of the option: halving dt cuts the error of a diffusion-dominated relaxation four-fold
    (backward Euler: two-fold).  (The lane kernels take all steps of a call in ONE launch and keep the history in their own layout;
    between calls it goes home to the handle -- the split must not show there either.)"""
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', kernel)
    nx = 96
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 3)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    dt = 0.2 * (6 * dx) * (nx * dx) / D.max()

    def run(order, dt_, splits):
        with _capi.PnpSolver(N, nx, dx, dt_, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
            s.set_newton(time_order=order, tol=1e-12)
            s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
            its = []
            for n in splits:
                s.step(n)
                its.append(s.newton_iterations())
            assert (s.get_status() == 0).all()
            return s.get_state()[0], its
    a, ia = run(2, dt, [5])
    b, ib = run(2, dt, [2, 3])
    assert np.array_equal(a, b) and np.array_equal(ia[0], ib[0] + ib[1])
    ref, _ = run(2, dt / 16, [8 * 16])
    err = {}
    for order in (1, 2):
        e1 = np.abs(run(order, dt, [8])[0] - ref).max()
        e2 = np.abs(run(order, dt / 2, [16])[0] - ref).max()
        err[order] = e1 / e2
    assert 1.6 < err[1] < 2.6 and 3.0 < err[2] < 5.5, err


@pytest.mark.parametrize("kernel,B", [('lane', 133), ('lane2', 70), ('lane4', 37)])
def test_bdf2_history_inside_the_lane_kernels_with_several_chunks_and_a_mask(kernel, B, monkeypatch):
    """The lane kernels keep the BDF2 history in their transposed workspace and bring it home between launches.  With a workspace of
    ONE group (the launcher walks the batch in chunks, every chunk transposes its own history in and out) and split calls the
    trajectory is the one-call, whole-workspace trajectory to the bit; and operating points masked out of a call keep state AND
    history, so that they continue from where they were."""
    N, nx = 6, 64
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 17)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    dt = 0.2 * (6 * dx) * (nx * dx) / D.max()
    kw = dict(time_order=2, mpb_radius=[3.5e-10] * N, wall_bc='stern', stern_capacitance=0.25)

    def run(groups, calls, mask_call=None):
        monkeypatch.setenv('CATINT_NEWTON_KERNEL', kernel)
        if groups:
            monkeypatch.setenv('CATINT_NEWTON_LANE_GROUPS', str(groups))
        else:
            monkeypatch.delenv('CATINT_NEWTON_LANE_GROUPS', raising=False)
        with _capi.PnpSolver(N, nx, dx, dt, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
            s.set_newton(**kw)
            s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
            for i, n in enumerate(calls):
                if mask_call is not None and i == mask_call[0]:
                    s.set_lane_mask(mask_call[1])
                s.step(n)
                if mask_call is not None and i == mask_call[0]:
                    s.set_lane_mask(None)
            assert (s.get_status() == 0).all()
            return s.get_state()[0], s.get_state()[1]
    whole = run(0, [5])
    chunks = run(1, [2, 1, 2])
    assert np.array_equal(whole[0], chunks[0]) and np.array_equal(whole[1], chunks[1])
    # the odd points sit out the middle call: they end on the 4-step trajectory, the even ones on the 5-step trajectory
    mask = (np.arange(B) % 2 == 0).astype(np.int32)
    mixed = run(1, [2, 1, 2], mask_call=(1, mask))
    four = run(0, [4])
    even, odd = mask == 1, mask == 0
    assert np.array_equal(mixed[0][even], whole[0][even]) and np.array_equal(mixed[0][odd], four[0][odd])


# ---- constant convection velocity (tp.system['flow rate'], comsol_model.py:901-903) ------------------------------------------------------
@pytest.mark.parametrize("N,nx,B,kernel,kw,graded", [
    (3, 128, 5, None, {}, False),                                                                       # pair kernel
    (4, 150, 4, None, dict(wall_bc='stern', stern_capacitance=0.2, mpb_radius=[4e-10, 3e-10, 0.0, 3.5e-10]), True),
    (2, 1100, 3, None, {}, False),                                                                      # lane teams (grid too long for the pair kernel)
    (6, 96, 4, None, dict(wall_bc='stern', stern_capacitance=0.25, mpb_radius=[3.5e-10] * 6), True),    # lane teams
    (6, 96, 9, 'sweep', dict(mpb_radius=[3.5e-10] * 6), False),                                         # one-sided sweep
    (7, 80, 9, 'both', {}, False),                                                                      # two-sided sweep
    (3, 130, 4, 'generic', {}, False),                                                                  # row-per-thread kernel
    (8, 64, 70, 'team', dict(mpb_radius=[3.5e-10] * 8), False),                                         # lane teams at a batch the lane kernels would take
])
def test_convection_velocity_matches_oracle(N, nx, B, kernel, kw, graded, monkeypatch):
    """+ c v in every Nernst-Planck flux: the drift argument of an edge loses v h_e / D_k.  Velocities of both signs, of the order of
    D / L so that the profiles change visibly; stationary and transient; every workgroup-per-point / lane-team kernel family (the lane
    kernels: tests/test_gpu_lane.py, tests/test_gpu_lane2.py)."""
    # (velocities of D / L: the profiles change by e^3 across the cell -- a velocity of D / dx would pile up e^(nx) at one end)
    if kernel:
        monkeypatch.setenv('CATINT_NEWTON_KERNEL', kernel)
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 21)
    x = None
    if graded:
        x = np.cumsum(np.concatenate([[0.0], np.geomspace(0.4, 2.5, nx - 1)]))
    Lx = (x[-1] if graded else nx - 1) * dx
    for v in (3.0 * D.max() / Lx, -2.0 * D.max() / Lx):              # cell Peclet numbers of a few over the whole grid
        got, ref = run_both(N, nx, B=B, seed=21, newton_kw=kw, x=x, velocity=v)
        assert_close(got, ref)
        base, _ = run_both(N, nx, B=B, seed=21, newton_kw=kw, x=x)
        assert np.abs(got[0] - base[0]).max() > 1e-3 * np.abs(base[0]).max()           # the term is not a no-op
    dt = 0.2 * (6 * dx) * (nx * dx) / D.max()
    got, ref = run_both(N, nx, B=B, seed=22, newton_kw=kw, x=x, velocity=3.0 * D.max() / Lx, dt=dt, nsteps=3, stationary=False)
    assert_close(got, ref)
