"""GPU parity of the lane kernel (catint_amd/csrc/pnp_lane.hip: one operating point per lane, block Thomas from both ends in
registers, batch-innermost state and records) against the CPU oracle oracle/pnp_physical.py and against the lane-team kernels,
through the C-ABI.  Same bar as tests/test_gpu_newton.py: states to 2e-9 of the profile's scale and IDENTICAL Newton
iteration counts per operating point (both sides run the same damping and stopping rules).

The reference hands this solve to COMSOL (catint/comsol_model.py:465-516: fully coupled Newton, direct linear solver); parity
with COMSOL itself is unpinned, the oracle is pinned by analytic answers (tests/test_physical_oracle.py).
"""
import numpy as np
import pytest

from catint_amd import _capi
from tests.test_gpu_newton import BETA, EPS, assert_close, make_lanes, run_both, run_gpu_only

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def lane_kernel(monkeypatch):
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', 'lane')


@pytest.mark.parametrize("N,nx", [(1, 33), (2, 64), (2, 201), (3, 128), (3, 513), (4, 100), (5, 70), (5, 71), (6, 96), (6, 9), (7, 50),
                                  (7, 51), (8, 40), (8, 129)])
def test_stationary_matches_oracle(N, nx):
    # even and odd row counts (the upward half rests for one row when nx is even), very short grids, a batch that is not a
    # multiple of the 32 operating points of a wave
    kw = {'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * N} if N >= 6 else {}
    got, ref = run_both(N, nx, B=37, seed=N * 31 + nx, newton_kw=kw)
    assert_close(got, ref)


@pytest.mark.parametrize("nx", [5, 6, 7, 8])
def test_shortest_grids(nx):
    got, ref = run_both(3, nx, B=5, seed=nx)
    assert_close(got, ref)


def test_several_chunks(monkeypatch):
    # the workspace holds one group of 32 operating points: the launcher walks the batch in five chunks
    monkeypatch.setenv('CATINT_NEWTON_LANE_GROUPS', '1')
    got, ref = run_both(4, 48, B=133, seed=3)
    assert_close(got, ref)


def test_transient_steps_match_oracle():
    N, nx = 3, 160
    D, q, cb, dx, phiM = make_lanes(N, nx, 4, 7)
    dt = 0.2 * (6 * dx) * (nx * dx) / D.max()
    got, ref = run_both(N, nx, B=40, seed=7, dt=dt, nsteps=6, stationary=False)
    assert_close(got, ref)
    assert got[2].min() >= 6 * 2          # at least two Newton iterations per step


def test_transient_steric_eight_species():
    got, ref = run_both(8, 96, B=35, seed=5, dt=2e-8, nsteps=4, stationary=False,
                        newton_kw={'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * 8})
    assert_close(got, ref)


def test_stern_layer_and_steric_ions():
    a = [4.1e-10, 3.1e-10, 3.5e-10]
    got, ref = run_both(3, 240, B=6, seed=11, phi_lo=-1.5, phi_hi=1.0, cref=100.0,
                        newton_kw=dict(wall_bc='stern', stern_capacitance=0.2, phi_pzc=0.05, mpb_radius=a, maxit=60))
    assert_close(got, ref)


def test_wall_fluxes():
    rng = np.random.default_rng(5)
    B, N = 4, 3
    flux = rng.uniform(-2e-4, 2e-4, (B, N))
    got, ref = run_both(N, 96, B=B, seed=13, flux=flux)
    assert_close(got, ref)


@pytest.mark.parametrize("N,nx,B", [(4, 150, 5), (3, 128, 4), (6, 96, 4), (8, 64, 3)])
def test_butler_volmer_and_langmuir_wall_kinetics(N, nx, B):
    rng = np.random.default_rng(N * 100 + nx)
    wk = [{'species': 2, 'k': rng.uniform(0.05, 1.0, B), 'nu': [0.0, 0.0, -1.0] + [1.0] * (N > 3) + [0.0] * max(N - 4, 0), 'alpha': -6.0,
           'saturation': 0.05},
          {'species': 0, 'k': rng.uniform(1e-4, 1e-3, B), 'nu': [-1.0, 0.0, 0.5] + [0.0] * (N - 3), 'alpha': 3.0},
          {'species': -1, 'k': rng.uniform(1e-6, 1e-5, B), 'nu': [0.0, 1.0, 0.0] + [0.0] * (N - 3), 'alpha': -4.0},
          {'species': 1, 'k': rng.uniform(1e-4, 1e-3, B), 'nu': [0.0, -1.0, 0.0] + [0.0] * (N - 3), 'saturation': 0.2}]
    got, ref = run_both(N, nx, B=B, seed=23 + N, wall_kinetics=wk,
                        newton_kw=dict(wall_bc='stern', stern_capacitance=0.2, phi_pzc=0.05, mpb_radius=[4.1e-10] + [0.0] * (N - 1)))
    assert_close(got, ref)


@pytest.mark.parametrize("N,nx,kw", [(3, 96, {}), (3, 257, dict(wall_bc='stern', stern_capacitance=0.2, mpb_radius=[4.1e-10, 3e-10, 0.0])),
                                     (6, 80, {})])
def test_graded_grid(N, nx, kw):
    from catint_amd.host import graded_mesh
    x = graded_mesh(4000.0, 1.0, nx)
    got, ref = run_both(N, nx, B=3, seed=nx, x=x, newton_kw=kw, points_per_debye=8.0)
    assert_close(got, ref)
    got, ref = run_both(N, nx, B=3, seed=nx + 1, x=x, newton_kw=kw, points_per_debye=8.0, dt=1e-7, nsteps=3, stationary=False)
    assert_close(got, ref)


# ---- homogeneous reactions and the convection term in the lane kernels (MODE 2 instances; comsol_model.py:781-867, :901-903) ------------
RX4 = [{'lhs': [1], 'rhs': [2], 'kf': 4e5, 'kr': 9e5}, {'lhs': [2, 2], 'rhs': [3], 'kf': 3e3, 'kr': 0.0},
       {'lhs': [0, 1], 'rhs': [3], 'kf': 1e3, 'kr': 2e4}, {'lhs': [], 'rhs': [0, 1], 'kf': 2e3, 'kr': 1.5e2}]      # last: H2O <-> A+ + B-
RX6 = [{'lhs': [1], 'rhs': [2], 'kf': 4e5, 'kr': 9e5}, {'lhs': [0, 2, 2], 'rhs': [4, 5], 'kf': 5.0, 'kr': 1e2}]


@pytest.mark.parametrize("N,nx,B,rx,kw", [
    (4, 120, 37, RX4, {}),
    (3, 200, 9, [{'lhs': [1], 'rhs': [2], 'kf': 4e5, 'kr': 9e5}, {'lhs': [0, 2], 'rhs': [1], 'kf': 2e3, 'kr': 1e4}],
     dict(wall_bc='stern', stern_capacitance=0.2, mpb_radius=[4.1e-10, 3e-10, 0.0], maxit=60)),
    (6, 64, 35, RX6, {}),
    (7, 96, 40, RX6 + [{'lhs': [3, 6], 'rhs': [0], 'kf': 2e2, 'kr': 7e3}], dict(wall_bc='stern', stern_capacitance=0.25, mpb_radius=[3.5e-10] * 7)),
    (8, 51, 33, RX6 + [{'lhs': [6], 'rhs': [7], 'kf': 1e6, 'kr': 3e6}], dict(mpb_radius=[3.5e-10] * 8)),
])
def test_homogeneous_reactions_match_oracle(N, nx, B, rx, kw):
    """Mass-action reactions in activities: source and Jacobian (a rank-one update of the species block per reaction side) in the lane's
    block row; LU without row exchanges under the pivot monitor.  Stationary from a bulk out of equilibrium, and transient steps."""
    got, ref = run_both(N, nx, B=B, seed=31 + N, reactions=rx, newton_kw=kw, points_per_debye=2.0 if N == 4 else 6.0,
                        **(dict(phi_lo=-1.0, phi_hi=0.8, cref=100.0) if N == 3 else {}))
    assert_close(got, ref)
    got, ref = run_both(N, nx, B=B, seed=41 + N, reactions=rx, newton_kw=kw, dt=1e-7, nsteps=3, stationary=False)
    assert_close(got, ref)


@pytest.mark.parametrize("N,nx,B,kw,graded", [(3, 128, 37, {}, False), (6, 96, 34, dict(wall_bc='stern', stern_capacitance=0.25, mpb_radius=[3.5e-10] * 6), True),
                                              (8, 64, 70, dict(mpb_radius=[3.5e-10] * 8), False)])
def test_convection_velocity_matches_oracle(N, nx, B, kw, graded):
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 21)
    x = np.cumsum(np.concatenate([[0.0], np.geomspace(0.4, 2.5, nx - 1)])) if graded else None
    Lx = (x[-1] if graded else nx - 1) * dx
    for v in (3.0 * D.max() / Lx, -2.0 * D.max() / Lx):
        got, ref = run_both(N, nx, B=B, seed=21, newton_kw=kw, x=x, velocity=v)
        assert_close(got, ref)
    base, _ = run_both(N, nx, B=B, seed=21, newton_kw=kw, x=x)
    assert np.abs(got[0] - base[0]).max() > 1e-3 * np.abs(base[0]).max()           # the term is not a no-op
    dt = 0.2 * (6 * dx) * (nx * dx) / D.max()
    got, ref = run_both(N, nx, B=B, seed=22, newton_kw=kw, x=x, velocity=3.0 * D.max() / Lx, dt=dt, nsteps=3, stationary=False, reactions=RX6 if N >= 6 else None)
    assert_close(got, ref)


@pytest.mark.parametrize("N,nx", [(3, 512), (6, 96), (8, 40)])
def test_bitwise_reproducible(N, nx):
    a = run_gpu_only(N, nx, 70, 99)
    b = run_gpu_only(N, nx, 70, 99)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_lane_kernel_equals_the_lane_team_kernels(monkeypatch):
    """Same Newton systems, another elimination (LU without row exchanges in one lane against Gauss-Jordan across a team):
    states to rounding and identical iteration counts on a batch of 300 operating points, steric ions, Stern wall."""
    outs = []
    for kern in ('lane', 'team'):
        monkeypatch.setenv('CATINT_NEWTON_KERNEL', kern)
        N, nx, B = 8, 128, 300
        D, q, cb, dx, phiM = make_lanes(N, nx, B, 5)
        c0 = np.repeat(cb[:, :, None], nx, axis=2)
        pb = np.zeros((B, 4))
        pb[:, 0] = phiM
        with _capi.PnpSolver(N, nx, dx, 1.0, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
            s.set_newton(stern_capacitance=0.25, wall_bc='stern', mpb_radius=[3.5e-10] * N)
            s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
            st = s.solve_stationary()
            c, phi, _, _ = s.get_state()
            outs.append((c, phi, s.newton_iterations(), st))
    (c1, p1, it1, st1), (c2, p2, it2, st2) = outs
    assert np.array_equal(it1, it2) and np.array_equal(st1, st2) and (st1 == 0).all()
    assert np.abs(c1 - c2).max() <= 1e-9 * np.abs(c1).max() and np.abs(p1 - p2).max() <= 1e-10


def test_not_converged_and_nan_are_reported_per_lane():
    N, nx, B = 3, 64, 40
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 3, phi_lo=-0.6, phi_hi=0.6)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    c0[7, 1, 5] = np.nan
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    with _capi.PnpSolver(N, nx, dx, 1.0, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
        s.set_newton(maxit=3)
        s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
        st = s.solve_stationary()
        its = s.newton_iterations()
    assert st[7] == 2                                    # NaN lane
    others = np.delete(np.arange(B), 7)
    assert set(np.unique(st[others])) <= {0, 1} and (st[others] == 1).any()
    assert (its[st == 1] == 4).all()                     # maxit + 1 marks a failed solve


@pytest.mark.parametrize("kernel", ['lane', 'lane2', 'lane4', 'team'])
def test_lane_mask_and_set_lanes(kernel, monkeypatch):
    """pnp_set_lane_mask / pnp_set_lanes, the two entry points behind the rerun ladder (reference catint/calculator.py:466-531): a solve
    restricted to some lanes leaves the others bit for bit alone; lanes recovered elsewhere are patched in without touching the rest
    of the batch and the confirming solve needs one iteration.  (A masked solve of the lane kernels deals only the lanes it solves to
    slots and launches the groups they fill -- pnp_capi.hip: lane_order.)"""
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', kernel)
    N, nx, B = 6, 64, 70
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 11)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    kw = dict(stern_capacitance=0.25, wall_bc='stern', mpb_radius=[3.5e-10] * N)
    with _capi.PnpSolver(N, nx, dx, 1.0, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
        s.set_newton(**kw)
        s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
        st = s.solve_stationary()
        cref, pref = s.get_state()[:2]
        itref = s.newton_iterations()
        assert (st == 0).all()
    even = np.arange(B) % 2 == 0
    with _capi.PnpSolver(N, nx, dx, 1.0, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
        s.set_newton(**kw)
        s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
        s.set_lane_mask(even)
        st = s.solve_stationary()
        c1, p1 = s.get_state()[:2]
        it1 = s.newton_iterations()
        assert np.array_equal(c1[even], cref[even]) and np.array_equal(p1[even], pref[even]) and np.array_equal(it1[even], itref[even])
        assert np.array_equal(c1[~even].reshape(-1, N, nx), c0[~even]) and (p1[~even] == 0).all() and (it1[~even] == 0).all()
        # the odd lanes arrive from elsewhere (here: the reference run, perturbed in the last digits); only they are solved again
        lanes = np.flatnonzero(~even)
        s.set_lanes(lanes, cref[lanes] * (1 + 1e-9), pref[lanes])
        s.set_lane_mask(~even)
        st = s.solve_stationary()
        s.set_lane_mask(None)
        c2, p2 = s.get_state()[:2]
        it2 = s.newton_iterations()
        assert (st == 0).all()
        assert np.array_equal(c2[even], cref[even]) and np.array_equal(it2[even], itref[even])          # untouched by patch and solve
        assert (it2[~even] <= 2).all() and np.abs(c2 - cref).max() <= 1e-8 * np.abs(cref).max()


@pytest.mark.parametrize("kernel,N", [('lane', 3), ('lane', 8), ('lane2', 6), ('lane2', 8), ('lane4', 5), ('lane4', 8)])
def test_pivot_monitor_reports_a_lane_as_not_converged(kernel, N, monkeypatch):
    """The lane kernels eliminate without row exchanges; a multiplier beyond 1e8 marks the lane and a marked lane is never reported
    converged (status 1: the rerun ladder then solves it with the pivoting kernels).  On well-posed systems the monitor stays silent
    (every other test of this file); with the limit forced below one every lane trips it -- the states are the same, the flags are not."""
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', kernel)
    a = run_gpu_only(N, 48, 40, 7)
    monkeypatch.setenv('CATINT_LANE_PIVOT_LIMIT', '1e-6')
    D, q, cb, dx, phiM = make_lanes(N, 48, 40, 7)
    c0 = np.repeat(cb[:, :, None], 48, axis=2)
    pb = np.zeros((40, 4))
    pb[:, 0] = phiM
    with _capi.PnpSolver(N, 48, dx, 1.0, BETA, EPS, D, q, method='Newton', batch_capacity=40) as s:
        s.set_newton()
        s.set_batch(c0, pb, np.zeros(40), np.zeros((40, N)))
        st = s.solve_stationary()
        c = s.get_state()[0]
    assert (st == 1).all() and np.array_equal(c, a[0])


@pytest.mark.parametrize("kernel,N,nx,B,groups", [('lane', 3, 96, 203, None), ('lane', 8, 64, 150, '2'), ('lane2', 6, 80, 133, None),
                                                  ('lane2', 8, 48, 97, '3'), ('lane4', 7, 80, 133, None), ('lane4', 8, 48, 97, '5')])
def test_points_ordered_by_expected_iterations_give_the_same_bits(kernel, N, nx, B, groups, monkeypatch):
    """The host deals the operating points to the slots (group, lane) in the order of the Newton iterations they are expected to need
    (pnp_capi.hip: lane_order -- first call: wall-to-bulk potential difference; later calls: the previous call's iteration counts), so
    that the lanes of a wave finish together.  A point's arithmetic does not depend on its slot: state, iteration counts and flags
    equal those of the identity order to the bit, over a first solve, transient steps after it (ordered by counts), a lane mask and
    several workspace chunks."""
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', kernel)
    if groups:
        monkeypatch.setenv('CATINT_NEWTON_LANE_GROUPS', groups)
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 11)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    dt = 0.2 * (6 * dx) * (nx * dx) / D.max()
    mask = (np.arange(B) % 3 != 1).astype(np.int32)
    outs = []
    for order in ('0', '1'):
        monkeypatch.setenv('CATINT_LANE_ORDER', order)
        with _capi.PnpSolver(N, nx, dx, dt, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
            s.set_newton(stern_capacitance=0.25, wall_bc='stern', mpb_radius=[3.5e-10] * N if N >= 6 else None)
            s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
            s.step(2)
            it_a = s.newton_iterations()
            s.step(3)
            it_b = s.newton_iterations()
            s.set_lane_mask(mask)
            s.step(1)
            s.set_lane_mask(None)
            c, phi, _, _ = s.get_state()
            outs.append((c, phi, it_a, it_b, s.newton_iterations(), s.get_status()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert len(set(outs[0][2])) > 1 and (outs[0][5] == 0).all()       # different counts within the batch: the order is not trivial


def test_first_solve_after_an_upload_is_ordered_by_the_potential_difference_and_options_are_per_handle(monkeypatch):
    """pnp_set_batch returned before it reset the lane order of a Newton handle, so the first solve after an upload ran in
    batch order.  pnp_get_lane_order shows the order of the last launch: after an upload the points are dealt by |phiM - phi_bulk|,
    largest first; after a solve by the iteration counts of that solve; after the next upload by the potential difference again.
    And pnp_set_option: two handles of one process run different kernel families (the environment is only read by pnp_create), an
    unknown key is refused."""
    monkeypatch.delenv('CATINT_NEWTON_KERNEL', raising=False)
    N, nx, B = 6, 64, 200
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 19)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    by_potential = np.argsort(-np.abs(phiM).astype(np.float32), kind='stable')
    with _capi.PnpSolver(N, nx, dx, 1.0, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s, \
            _capi.PnpSolver(N, nx, dx, 1.0, BETA, EPS, D, q, method='Newton', batch_capacity=B) as t:
        s.set_option('NEWTON_KERNEL', 'lane')
        t.set_option('newton_kernel', 'team')
        with pytest.raises(_capi.PnpError):
            s.set_option('NO_SUCH_SWITCH', '1')
        for solver in (s, t):
            solver.set_newton()
            solver.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
        assert len(s.lane_order()) == 0                          # nothing launched yet
        st = s.solve_stationary()
        assert (st == 0).all() and np.array_equal(s.lane_order(), by_potential)
        its = s.newton_iterations()
        t.solve_stationary()
        assert len(t.lane_order()) == 0 and np.array_equal(t.newton_iterations(), its)      # lane teams: no order; same counts
        assert not np.array_equal(s.get_state()[0], t.get_state()[0])                       # ... from another linear solver
        s.solve_stationary()                                     # warm: ordered by the counts of the solve before
        o2 = s.lane_order()
        assert np.array_equal(np.sort(o2), np.arange(B)) and (np.diff(its[o2]) <= 0).all()
        s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))      # a new upload: the counts are gone, the key is the potential again
        s.solve_stationary()
        assert np.array_equal(s.lane_order(), by_potential)


@pytest.mark.parametrize("N,nx,B,kw", [
    (2, 64, 70, {}), (3, 97, 37, {}), (4, 80, 40, {}),
    (6, 96, 70, {'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * 6}),
    (8, 65, 37, {'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * 8}),
    (8, 64, 35, {'time_order': 2, 'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * 8}),
])
def test_update_fused_into_the_back_substitution_and_separate_passes_are_the_same_iteration(N, nx, B, kw):
    """The lane kernel's two forms (option LANE_FUSED; the fused one runs by default since round 4): the update applied inside the back-substitution with two state copies, or written out and applied by a third
    pass.  Same arithmetic in the same order -- timesteps (first steps from the bulk state: damped iterations walk the back-substitution
    twice) and a stationary solve give the same bits, iteration counts and status either way; and both match the oracle."""
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 23, phi_lo=-0.25, phi_hi=0.25)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    dt = 0.3 * (6 * dx) * (nx * dx) / D.max()
    outs = []
    for fused in ('0', '1'):
        with _capi.PnpSolver(N, nx, dx, dt, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
            s.set_option('LANE_FUSED', fused)
            s.set_newton(**kw)
            s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
            s.step(3)
            a = (s.get_state()[0], s.get_state()[1], s.newton_iterations(), s.get_status())
            st = s.solve_stationary()
            outs.append(a + (s.get_state()[0], s.get_state()[1], s.newton_iterations(), st))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert (outs[0][3] == 0).all() and (outs[0][7] == 0).all()
    got, ref = run_both(N, nx, B=B, seed=23, dt=dt, nsteps=3, stationary=False, newton_kw=kw, phi_lo=-0.25, phi_hi=0.25)
    assert_close(got, ref)
    assert np.array_equal(got[0], outs[0][0])


@pytest.mark.parametrize("N,nx,B,kw", [
    (2, 64, 70, {}), (3, 97, 37, {}), (4, 80, 40, {'stern_capacitance': 0.25, 'wall_bc': 'stern'}),
    (5, 70, 37, {}),
    (6, 96, 70, {'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * 6}),
    (7, 51, 35, {'mpb_radius': [3.5e-10] * 7}),
    (8, 65, 37, {'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * 8}),
    (8, 64, 35, {'time_order': 2, 'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * 8}),
])
def test_record_columns_in_single_precision_give_the_same_iteration(N, nx, B, kw, monkeypatch):
    """Option LANE_RECORDS = f32: the columns T of the block-Thomas records travel as floats (t, the elimination and the residual stay
    double).  The back-substitution then puts a relative error of ~1e-7 into the Newton UPDATE; the next residual is exact, so the
    iteration ends on the same state: against the oracle -- which solves its linear systems in double throughout -- states within the
    suite's bound and the oracle's iteration counts (one more iteration in a few operating points whose last update lands a hair above
    the tolerance), on timesteps from the bulk state (damped first iterations) and on a stationary solve; and the states of the two
    record formats agree below the Newton tolerance."""
    D, q, cb, dx, phiM = make_lanes(N, nx, 4, 29)
    dt = 0.3 * (6 * dx) * (nx * dx) / D.max()
    monkeypatch.setenv('CATINT_LANE_RECORDS', 'f64')
    ref64, _ = run_both(N, nx, B=B, seed=29, dt=dt, nsteps=4, stationary=False, newton_kw=kw)
    def close_with_counts_at_most_one_apart(got, ref):
        # (an update that lands a hair above the tolerance costs the inexact solve one more iteration now and then: the counts are the
        #  oracle's in all but a few operating points, never more than one apart; the states are held to the suite's bound as always)
        c, phi, its, st = got
        rc, rphi, rit = ref
        assert np.all(st == 0), st
        cscale = np.abs(rc).max(axis=2, keepdims=True)
        assert np.abs(c - rc).max() <= 2e-9 * cscale.max() and (np.abs(c - rc) / (np.abs(rc) + 1e-3 * cscale)).max() < 1e-6
        assert np.abs(phi - rphi).max() <= 2e-9 * max(np.abs(rphi).max(), 0.025)
        d = np.abs(its.astype(int) - np.asarray(rit, int))
        assert d.max() <= 1 and (d != 0).mean() <= 0.1, (its, rit)
    monkeypatch.setenv('CATINT_LANE_RECORDS', 'f32')
    got, ref = run_both(N, nx, B=B, seed=29, dt=dt, nsteps=4, stationary=False, newton_kw=kw)
    close_with_counts_at_most_one_apart(got, ref)
    assert not np.array_equal(got[0], ref64[0])                      # (another linear solve: not the same bits)
    assert np.abs(got[0] - ref64[0]).max() <= 1e-9 * np.abs(ref64[0]).max() and np.abs(got[1] - ref64[1]).max() <= 1e-10
    if 'time_order' not in kw:
        got, ref = run_both(N, nx, B=B, seed=31, newton_kw=kw)       # stationary
        close_with_counts_at_most_one_apart(got, ref)
    if N == 6:                                                        # homogeneous reactions + convection: the MODE 2 instances
        got, ref = run_both(N, nx, B=B, seed=41, reactions=RX6, newton_kw=kw, dt=1e-7, nsteps=3, stationary=False, velocity=1e-4)
        close_with_counts_at_most_one_apart(got, ref)


@pytest.mark.parametrize("kernel,N,nx", [('lane', 3, 96), ('lane', 8, 64), ('lane2', 6, 80), ('lane2', 8, 48), ('lane4', 5, 80), ('lane4', 8, 48)])
def test_error_estimate_stopping_rule(kernel, N, nx, monkeypatch):
    """pnp_newton_params.error_estimate: accept an iterate whose quadratic error estimate upd^2 / upd_prev is below the tolerance (saves
    the iteration that only confirms convergence) -- same rule in the oracle, same iteration counts; fewer than without."""
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', kernel)
    kw = {'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * N} if N >= 6 else {}
    D, q, cb, dx, phiM = make_lanes(N, nx, 4, 9)
    dt = 0.2 * (6 * dx) * (nx * dx) / D.max()
    got, ref = run_both(N, nx, B=41, seed=9, dt=dt, nsteps=4, stationary=False, newton_kw=dict(kw, error_estimate=True, tol=1e-8))
    assert_close(got, ref, rtol=2e-7)          # (the accepted iterate is within the tolerance of the fixed point, not at it)
    plain, _ = run_both(N, nx, B=41, seed=9, dt=dt, nsteps=4, stationary=False, newton_kw=dict(kw, tol=1e-8))
    assert got[2].sum() < plain[2].sum()
