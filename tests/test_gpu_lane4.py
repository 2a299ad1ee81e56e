"""GPU parity of the lane-quad kernel (catint_amd/csrc/pnp_lane4.hip: eight lanes per operating point -- two sweep directions x four
column lanes of the block row, columns of [D' | Ah | r] distributed modulo four, Gauss-Jordan with the pivot column broadcast by DPP)
against the CPU oracle and the lane / lane-pair / lane-team kernels, through the C-ABI.  Same bar as tests/test_gpu_lane.py: states to
2e-9 of the profile's scale and IDENTICAL Newton iteration counts.  (The solve the reference hands to COMSOL,
catint/comsol_model.py:465-516; reactions :781-867; convection :901-903.)"""
import numpy as np
import pytest

from catint_amd import _capi
from tests.test_gpu_newton import BETA, EPS, assert_close, make_lanes, run_both, run_gpu_only

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def lane4_kernel(monkeypatch):
    monkeypatch.setenv('CATINT_NEWTON_KERNEL', 'lane4')


@pytest.mark.parametrize("N,nx", [(5, 70), (5, 71), (6, 96), (6, 9), (7, 50), (7, 51), (8, 40), (8, 129), (8, 5), (7, 6)])
@pytest.mark.parametrize("steric", [False, True])
def test_stationary_matches_oracle(N, nx, steric):
    kw = {'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * N} if steric else {}
    got, ref = run_both(N, nx, B=37, seed=N * 31 + nx, newton_kw=kw)
    assert_close(got, ref)


def test_several_chunks(monkeypatch):
    monkeypatch.setenv('CATINT_NEWTON_LANE_GROUPS', '1')
    got, ref = run_both(6, 48, B=70, seed=3)
    assert_close(got, ref)


@pytest.mark.parametrize("N", [5, 6, 7, 8])
def test_transient_steric(N):
    got, ref = run_both(N, 96, B=35, seed=5 + N, dt=2e-8, nsteps=4, stationary=False,
                        newton_kw={'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * N})
    assert_close(got, ref)


RX6 = [{'lhs': [1], 'rhs': [2], 'kf': 4e5, 'kr': 9e5}, {'lhs': [0, 2, 2], 'rhs': [4, 5], 'kf': 5.0, 'kr': 1e2}]


@pytest.mark.parametrize("N,nx,B,rx,kw", [
    (5, 70, 21, RX6[:1] + [{'lhs': [0, 2], 'rhs': [4], 'kf': 2e3, 'kr': 1e4}, {'lhs': [], 'rhs': [0, 1], 'kf': 2e3, 'kr': 1.5e2}], {}),
    (6, 64, 35, RX6, {}),
    (7, 96, 40, RX6 + [{'lhs': [3, 6], 'rhs': [0], 'kf': 2e2, 'kr': 7e3}], dict(wall_bc='stern', stern_capacitance=0.25, mpb_radius=[3.5e-10] * 7)),
    (8, 51, 33, RX6 + [{'lhs': [6], 'rhs': [7], 'kf': 1e6, 'kr': 3e6}], dict(mpb_radius=[3.5e-10] * 8)),
])
def test_homogeneous_reactions_match_oracle(N, nx, B, rx, kw):
    """MODE 2 instances of the lane-pair kernel (see tests/test_gpu_lane.py): reactions, stationary and transient."""
    got, ref = run_both(N, nx, B=B, seed=31 + N, reactions=rx, newton_kw=kw)
    assert_close(got, ref)
    got, ref = run_both(N, nx, B=B, seed=41 + N, reactions=rx, newton_kw=kw, dt=1e-7, nsteps=3, stationary=False)
    assert_close(got, ref)


@pytest.mark.parametrize("N,nx,B,kw", [(6, 96, 34, dict(wall_bc='stern', stern_capacitance=0.25, mpb_radius=[3.5e-10] * 6)), (7, 80, 19, {})])
def test_convection_velocity_matches_oracle(N, nx, B, kw):
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 21)
    x = np.cumsum(np.concatenate([[0.0], np.geomspace(0.4, 2.5, nx - 1)]))
    Lx = x[-1] * dx
    for v in (3.0 * D.max() / Lx, -2.0 * D.max() / Lx):
        got, ref = run_both(N, nx, B=B, seed=21, newton_kw=kw, x=x, velocity=v)
        assert_close(got, ref)
    dt = 0.2 * (6 * dx) * (nx * dx) / D.max()
    got, ref = run_both(N, nx, B=B, seed=22, newton_kw=kw, x=x, velocity=3.0 * D.max() / Lx, dt=dt, nsteps=3, stationary=False, reactions=RX6)
    assert_close(got, ref)


def test_wall_fluxes_and_point_ions_transient():
    rng = np.random.default_rng(5)
    B, N = 20, 6
    flux = rng.uniform(-2e-4, 2e-4, (B, N))
    got, ref = run_both(N, 96, B=B, seed=13, flux=flux, dt=1e-7, nsteps=3, stationary=False)
    assert_close(got, ref)


@pytest.mark.parametrize("N,nx,B", [(6, 96, 4), (8, 64, 3), (5, 80, 5), (7, 64, 4)])
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


def test_graded_grid():
    from catint_amd.host import graded_mesh
    x = graded_mesh(4000.0, 1.0, 80)
    got, ref = run_both(6, 80, B=3, seed=80, x=x, points_per_debye=8.0)
    assert_close(got, ref)
    got, ref = run_both(6, 80, B=3, seed=81, x=x, points_per_debye=8.0, dt=1e-7, nsteps=3, stationary=False)
    assert_close(got, ref)


def test_bitwise_reproducible_and_equal_to_the_other_kernels(monkeypatch):
    a = run_gpu_only(8, 40, 70, 99)
    b = run_gpu_only(8, 40, 70, 99)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for kern in ('lane', 'lane2', 'team'):
        monkeypatch.setenv('CATINT_NEWTON_KERNEL', kern)
        c = run_gpu_only(8, 40, 70, 99)
        assert np.array_equal(a[2], c[2]) and np.abs(a[0] - c[0]).max() <= 1e-9 * np.abs(c[0]).max() and np.abs(a[1] - c[1]).max() <= 1e-10


def test_lane_mask_nan_and_not_converged():
    N, nx, B = 6, 64, 40
    D, q, cb, dx, phiM = make_lanes(N, nx, B, 3, phi_lo=-0.6, phi_hi=0.6)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    c0[7, 1, 5] = np.nan
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    with _capi.PnpSolver(N, nx, dx, 1.0, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
        s.set_newton(maxit=3)
        s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
        mask = np.ones(B, np.int32)
        mask[11] = 0
        s.set_lane_mask(mask)
        st = s.solve_stationary()
        its = s.newton_iterations()
        c = s.get_state()[0]
    assert st[7] == 2 and its[11] == 0 and np.array_equal(c[11].reshape(N, nx), c0[11])
    others = np.delete(np.arange(B), [7, 11])
    assert set(np.unique(st[others])) <= {0, 1} and (st[others] == 1).any() and (its[st == 1] == 4).all()
