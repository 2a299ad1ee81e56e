"""GPU: the streaming kernel (pnp_stream.hip: persistent single-wave workgroups, next row prefetched behind the current solve)
against the C oracle and -- bit for bit -- against the register-resident kernel whose arithmetic it shares (CATINT_PNP_KERNEL=4;
the LDS-staged family rounds a few products in another order: equal to rtol 1e-9, not bitwise), through the C-ABI.  Forced with
CATINT_PNP_KERNEL = 5 (registers only) / 6 (charge and gradient rows in LDS); CATINT_PNP_ST_WAVES_PER_CU = 1 shrinks the persistent
grid to 256 waves so that every wave walks several operating points, the last ones ragged."""
import numpy as np
import pytest

from oracle import c_oracle as CO
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def run(p, method, c0, pb, vz, fl, nsteps, spl):
    with solver_from_problem(p, method, batch_capacity=c0.shape[0]) as s:
        s.set_batch(c0, pb, vz, fl)
        s.step(nsteps, spl)
        return s.get_state() + (s.get_status(),)


@pytest.mark.parametrize('mode', ['5', '6', '7'])
@pytest.mark.parametrize('N,nx,B', [(3, 512, 700), (2, 200, 513), (6, 1024, 300), (3, 130, 257), (2, 66, 300), (4, 1026, 260), (6, 515, 259)])
@pytest.mark.parametrize('method', ['Crank-Nicolson', 'FTCS'])
def test_streaming_kernel_matches_oracle_and_previous_kernels(mode, N, nx, B, method, monkeypatch):
    p, c0, pb, vz, fl = make_batch(B, N, nx, seed=nx + N, phi_max=0.02, dt_factor=1e-4 if method == 'Crank-Nicolson' else 2e-5)
    rng = np.random.default_rng(nx)
    c0 = c0 * (1 + 0.05 * rng.uniform(-1, 1, c0.shape))
    fl = rng.uniform(-1e-4, 1e-4, fl.shape)
    nsteps = 4
    monkeypatch.setenv('CATINT_PNP_KERNEL', '4')
    monkeypatch.setenv('CATINT_PNP_WAVES_PER_GRID', '1')                 # (several waves per lane add the charge row in another order)
    ref = run(p, method, c0, pb, vz, fl, nsteps, 1)                      # the register-resident kernel, one launch per step
    monkeypatch.delenv('CATINT_PNP_WAVES_PER_GRID')
    monkeypatch.setenv('CATINT_PNP_KERNEL', mode)
    monkeypatch.setenv('CATINT_PNP_ST_WAVES_PER_CU', '1')
    got1 = run(p, method, c0, pb, vz, fl, nsteps, 1)                     # streaming kernel, one launch per step
    got4 = run(p, method, c0, pb, vz, fl, nsteps, 4)                     # ... and all four steps in one launch
    monkeypatch.delenv('CATINT_PNP_ST_WAVES_PER_CU')
    gotf = run(p, method, c0, pb, vz, fl, nsteps, 1)                     # full persistent grid
    for got in (got1, got4, gotf):
        assert (got[4] == 0).all()
        for a, b in zip(got[:4], ref[:4]):
            assert np.array_equal(a, b)
    sub = np.arange(0, B, 41)
    oc = np.ascontiguousarray(c0[sub].reshape(len(sub), N, nx).copy())
    ov, og, ol = CO.steps(p, method, oc, pb[sub], vz[sub], fl[sub], nsteps)
    assert relerr(got1[0][sub], oc) < RTOL and relerr(got1[1][sub], ov) < RTOL and relerr(got1[2][sub], og) < RTOL


def test_streaming_kernel_status_flags(monkeypatch):
    monkeypatch.setenv('CATINT_PNP_KERNEL', '5')
    monkeypatch.setenv('CATINT_PNP_ST_WAVES_PER_CU', '1')
    B = 600
    p, c0, pb, vz, fl = make_batch(B, 2, 128, phi_max=0.01)
    c0 = c0.copy()
    c0[301, 10] = np.nan
    c0[599, :128] = -1.0
    with solver_from_problem(p, 'Crank-Nicolson', batch_capacity=B) as s:
        s.set_batch(c0, pb, vz, fl)
        s.step(2, 1)
        st = s.get_status()
        c = s.get_state(potential=False)
    assert st[301] == 2 and st[599] == 3 and (np.delete(st, [301, 599]) == 0).all()
    assert np.isfinite(np.delete(c, 301, axis=0)).all()


@pytest.mark.parametrize('N,nx,B,kernel', [(3, 512, 1100, None), (2, 200, 1030, '2'), (4, 258, 2049, '4'), (3, 1024, 2300, '7'), (2, 2050, 530, None)])
@pytest.mark.parametrize('method', ['Crank-Nicolson', 'FTCS'])
def test_row_chunks_on_streams_of_their_own_change_nothing(N, nx, B, kernel, method, monkeypatch):
    """pnp_step with several launches in one call cuts the batch into row chunks whose launch sequences run on separate HIP
    streams (pnp_capi.hip: step_streams), and every other launch of one timestep may walk its rows from the last to the first:
    the same kernels on the same rows, so the state is the same to the bit as with one stream and one direction -- also with an odd batch, three and four chunks, the charge row's ping-pong over an odd number of steps, and work
    queued on the handle's stream before (upload) and after (read-back)."""
    p, c0, pb, vz, fl = make_batch(B, N, nx, seed=B, phi_max=0.02, dt_factor=1e-4 if method == 'Crank-Nicolson' else 2e-5)
    rng = np.random.default_rng(B)
    c0 = c0 * (1 + 0.05 * rng.uniform(-1, 1, c0.shape))
    fl = rng.uniform(-1e-4, 1e-4, fl.shape)
    if kernel:
        monkeypatch.setenv('CATINT_PNP_KERNEL', kernel)
    monkeypatch.setenv('CATINT_PNP_STEP_STREAMS', '1')
    monkeypatch.setenv('CATINT_PNP_ALTERNATE_ROWS', '0')
    ref = run(p, method, c0, pb, vz, fl, 5, 1)
    ref2 = run(p, method, c0, pb, vz, fl, 6, 2)
    for S in ('1', '2', '3', '4'):
        monkeypatch.setenv('CATINT_PNP_STEP_STREAMS', S)
        monkeypatch.setenv('CATINT_PNP_ALTERNATE_ROWS', '1')      # (every other one-step launch walks the rows backwards)
        got = run(p, method, c0, pb, vz, fl, 5, 1)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), S
        with solver_from_problem(p, method, batch_capacity=B) as s:       # two calls back to back, then more steps in one launch
            s.set_batch(c0, pb, vz, fl)
            s.step(2, 2)
            s.step(4, 2)
            got2 = s.get_state() + (s.get_status(),)
        for a, b in zip(got2, ref2):
            assert np.array_equal(a, b), S
    monkeypatch.delenv('CATINT_PNP_STEP_STREAMS')
    monkeypatch.delenv('CATINT_PNP_ALTERNATE_ROWS')
    got = run(p, method, c0, pb, vz, fl, 5, 1)                             # the default
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
