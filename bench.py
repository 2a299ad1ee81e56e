#!/usr/bin/env python3
"""Benchmark of the batched 1D PNP timestep path (BASELINE.json metric:
"batched 1D PNP Newton-timesteps/sec at 1/2/4/8 GPU; achieved HBM GB/s vs peak").

One "step" = one pass of the integrator's time-loop body (reference catint/calculator_old.py:512-558)
over one batch of B operating points.  Headline ("state written every step
(ntout = nt)", algorithmic bytes 16*(N+1)*nx per lane-step): the reference's time loop runs inside the
kernel, `--steps-per-launch` timesteps per launch (default 256, what pnp_step/pnp_integrate use); every step's new
state is written to HBM, the previous state is re-used from registers/LDS instead of being re-read, so the
MEASURED HBM traffic (roofline.traffic, rocprofv3) is about half the algorithmic figure.  The
one-launch-per-timestep variant (state read from and written to HBM by every launch) is reported next to
it as `per_step_launch`.  Workload at N=1: BASELINE.json configs[1] -- batch=1024 operating points,
3 species, 512 grid points, fp64.  Multi-GPU: the batch shards embarrassingly
(weak scaling: every rank owns `--batch` lanes), no collective in the timed region; one RCCL
all_gather of the polarization observables afterwards.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=256)
    ap.add_argument('--warmup', type=int, default=64)
    ap.add_argument('--batch', type=int, default=1024, help='operating points per GPU')
    ap.add_argument('--nspecies', type=int, default=3)
    ap.add_argument('--nx', type=int, default=512)
    ap.add_argument('--method', default='Crank-Nicolson')
    ap.add_argument('--steps-per-launch', type=int, default=256,
                    help='timesteps fused into one launch (state written to HBM every step); 1 = one launch per step')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='target CPU-baseline sample length')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fused', action='store_true')
    ap.add_argument('--large-batch', type=int, default=8192,
                    help='extra (non-headline) timed run at this batch size to show the oversubscribed regime; 0 = skip')
    ap.add_argument('--physical-steps', type=int, default=20,
                    help='extra (non-headline) timed run of the implicit physical mode (coupled Newton, block cyclic reduction) '
                         'on the same workload shape; 0 = skip')
    ap.add_argument('--traffic-json', default=os.path.join(ROOT, 'profiles', 'hbm_traffic_latest.json'),
                    help='rocprofv3 PMC result (tools/pmc_traffic.py) for this workload; merged into roofline.traffic')
    return ap.parse_args()


def cpu_baseline(prob, c0, pb, vz, fl, method, target_s):
    """The oracle (C port of the reference algorithm, Thomas solves) on the host cores this process may use."""
    from oracle import c_oracle as CO
    CO.load()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, CO.max_threads()))
    B = c0.shape[0]
    N, nx = prob.N, prob.nx
    c = c0.reshape(B, N, nx).copy()
    CO.steps(prob, method, c, pb, vz, fl, 2, want_potential=False, nthreads=cores)     # spin up the thread pool
    t0 = time.perf_counter()
    CO.steps(prob, method, c, pb, vz, fl, 200, want_potential=False, nthreads=cores)
    rate = B * 200 / max(time.perf_counter() - t0, 1e-9)
    nsteps = int(max(20, min(20000, target_s * rate / B)))
    c = c0.reshape(B, N, nx).copy()
    t0 = time.perf_counter()
    CO.steps(prob, method, c, pb, vz, fl, nsteps, want_potential=False, nthreads=cores)
    dt = time.perf_counter() - t0
    return {
        'value': B * nsteps / dt, 'unit': 'timesteps/s', 'cores': cores, 'kind': 'port',
        'sample': '%d lanes x %d steps of the same workload, C restatement of the reference algorithm '
                  '(oracle/pnp_oracle.c: banded Thomas instead of dense LU, OpenMP over lanes) in %.1f s' % (B, nsteps, dt),
    }


def cpu_reference_faithful(prob, c0, pb, vz, fl, method):
    """dense np.linalg.solve per species, exactly the reference's arithmetic (oracle/pnp_ref.py) -- 1 core."""
    from oracle import pnp_ref as R
    p = R.Problem(D=prob.D, charges=prob.charges, beta=prob.beta, eps=prob.eps, dx=prob.dx, nx=prob.nx, dt=prob.dt,
                  pb=pb[0], vzeta=float(vz[0]), flux_bound=fl[0])
    C0 = c0[0].reshape(prob.N, prob.nx).copy()
    C = C0.copy(); COLD = np.zeros_like(C)
    n = 4
    t0 = time.perf_counter()
    for i in range(n):
        R.cn_step(C, COLD, C0, p, first=(i == 0), solver='dense')
    return n / (time.perf_counter() - t0)


def physical_mode(args, device, with_cpu):
    """Implicit physical mode (PNP_METHOD_NEWTON) on the headline shape with SURVEY 8(d)'s synthetic inputs:
    phiM ~ U(-0.2, 0.2) V, dt = 0.1 lambda_D L / D_max, Newton to a scaled update of 1e-8."""
    from catint_amd import _capi
    from catint_amd.synthetic import make_batch
    B, N, nx = args.batch, args.nspecies, args.nx
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=4242, phi_max=0.2, dt_factor=0.1)
    pb = np.nan_to_num(pb)
    s = _capi.PnpSolver(N, nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton', batch_capacity=B,
                        device=device)
    s.set_newton(tol=1e-8)
    s.set_batch(c0, pb, vz, fl)
    s.step(5)
    s.synchronize()
    s.timer_start()
    s.step(args.physical_steps)
    ms = s.timer_stop()
    it = s.newton_iterations()
    ok = int((s.get_status() == 0).sum())
    s.close()
    sec = ms * 1e-3
    # fp64 VALU instructions per Newton iteration and lane of the pair kernel for N = 3 (static count of the ISA,
    # tools/probe/census_newton.py: 256 threads x (760 + 552 + 8 x 288 + 100)); peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz
    out = {'workload': 'batch=%d, %d species, %d points, backward Euler dt=%.3g s, Dirichlet wall, tol 1e-8' % (B, N, nx, prob.dt),
           'timesteps_per_s': B * args.physical_steps / sec, 'newton_iterations_per_s': float(it.sum()) / sec,
           'mean_newton_iterations_per_step': float(it.sum()) / (B * args.physical_steps),
           'ms_per_step': ms / args.physical_steps, 'lanes_ok': ok, 'bound': 'fp64 VALU issue / LDS exchange (not HBM)'}
    if N == 3 and nx <= 512:
        out['fp64_valu_util'] = 256 * 3716.0 * out['newton_iterations_per_s'] / (256 * 4 * 16 * 2.4e9)
    # the same steps with the quadratic error estimate as the stopping rule (pnp_newton_params.error_estimate: stop when the
    # NEXT update is predicted below tol -- the state meets the same tolerance, the last confirming iteration is not spent)
    try:
        s = _capi.PnpSolver(N, nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton', batch_capacity=B,
                            device=device)
        s.set_newton(tol=1e-8, error_estimate=True)
        s.set_batch(c0, pb, vz, fl)
        s.step(5)
        s.synchronize()
        s.timer_start()
        s.step(args.physical_steps)
        ms_e = s.timer_stop()
        it_e = s.newton_iterations()
        ok_e = int((s.get_status() == 0).sum())
        s.close()
        out['with_error_estimate'] = {'timesteps_per_s': B * args.physical_steps / (ms_e * 1e-3),
                                      'mean_newton_iterations_per_step': float(it_e.sum()) / (B * args.physical_steps), 'lanes_ok': ok_e}
    except Exception as e:
        out['with_error_estimate'] = {'error': str(e)}
    # BASELINE configs[4] shape in the same mode: 8 species (size-modified, Stern wall), 4096 points -- lane-team kernel
    try:
        LB = max(64, min(1024, B))
        p8, c8, pb8, vz8, fl8 = make_batch(LB, 8, 4096, seed=4343, phi_max=0.2, dt_factor=0.1)
        s8 = _capi.PnpSolver(8, 4096, p8.dx, p8.dt, p8.beta, p8.eps, p8.D, p8.charges, method='Newton', batch_capacity=LB, device=device)
        s8.set_newton(wall_bc='stern', stern_capacitance=0.2, tol=1e-8, mpb_radius=[4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 3e-10, 3e-10, 4.5e-10, 3.5e-10])
        s8.set_batch(c8, np.nan_to_num(pb8), vz8, fl8)
        s8.step(1)
        s8.synchronize()
        s8.timer_start()
        s8.step(3)
        ms8 = s8.timer_stop()
        it8 = s8.newton_iterations()
        ok8 = int((s8.get_status() == 0).sum())
        s8.close()
        out['config4_shape'] = {'workload': 'batch=%d, 8 species size-modified, 4096 points, Stern wall' % LB,
                                'timesteps_per_s': LB * 3 / (ms8 * 1e-3), 'newton_iterations_per_s': float(it8.sum()) / (ms8 * 1e-3),
                                'lanes_ok': ok8}
    except Exception as e:      # never let the extra line break the headline
        out['config4_shape'] = {'error': str(e)}
    # large batch of large blocks: the sweep kernel (block Thomas, one lane team per operating point)
    try:
        SB = 8192
        p8, c8, pb8, vz8, fl8 = make_batch(SB, 8, 512, seed=4444, phi_max=0.2, dt_factor=0.1)
        s8 = _capi.PnpSolver(8, 512, p8.dx, p8.dt, p8.beta, p8.eps, p8.D, p8.charges, method='Newton', batch_capacity=SB, device=device)
        s8.set_newton(wall_bc='stern', stern_capacitance=0.2, tol=1e-8, mpb_radius=[4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 3e-10, 3e-10, 4.5e-10, 3.5e-10])
        s8.set_batch(c8, np.nan_to_num(pb8), vz8, fl8)
        s8.step(1)
        s8.synchronize()
        s8.timer_start()
        s8.step(4)
        ms8 = s8.timer_stop()
        it8 = s8.newton_iterations()
        ok8 = int((s8.get_status() == 0).sum())
        s8.close()
        out['large_batch_8_species'] = {'workload': 'batch=%d, 8 species size-modified, 512 points, Stern wall (sweep kernel)' % SB,
                                        'timesteps_per_s': SB * 4 / (ms8 * 1e-3), 'newton_iterations_per_s': float(it8.sum()) / (ms8 * 1e-3),
                                        'lanes_ok': ok8}
    except Exception as e:
        out['large_batch_8_species'] = {'error': str(e)}
    if with_cpu:
        from oracle import pnp_physical as PH
        nl, ns = 2, 2
        t0 = time.perf_counter()
        for b in range(nl):
            p = PH.PhysicalProblem(D=prob.D, charges=prob.charges, beta=prob.beta, eps=prob.eps, dx=prob.dx, nx=nx,
                                   c_bulk=c0[b].reshape(N, nx)[:, -1], phiM=pb[b, 0])
            PH.integrate(p, c0[b].reshape(N, nx), np.zeros(nx), prob.dt, ns, tol=1e-8)
        out['cpu_port_timesteps_per_s_1core'] = nl * ns / (time.perf_counter() - t0)
    return out


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import torch
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback for the transport path)')
    ndev = torch.cuda.device_count()
    device = local_rank % max(ndev, 1)          # one rank per GPU on a full node; ranks share GPUs only in rehearsals
    backend = os.environ.get('CATINT_DIST_BACKEND', 'nccl')   # nccl = RCCL over xGMI; gloo for single-GPU rehearsals
    if world > 1 or os.environ.get('CATINT_FORCE_DIST'):     # the env hook lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist
        torch.cuda.set_device(device)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', device))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(device)
    comm_dev = torch.device('cuda', device) if backend == 'nccl' else torch.device('cpu')

    from catint_amd.synthetic import make_batch
    from catint_amd.host import solver_from_problem
    B, N, nx = args.batch, args.nspecies, args.nx
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=1000 + rank, phi_max=0.025, dt_factor=1e-5)
    solver = solver_from_problem(prob, args.method, batch_capacity=B, device=device)
    solver.set_batch(c0, pb, vz, fl)

    def barrier():
        solver.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def timed(nsteps, spl):
        barrier()
        t0 = time.perf_counter()
        solver.timer_start()
        solver.step(nsteps, spl)
        ev_ms = solver.timer_stop()
        solver.synchronize()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([wall, ev_ms], device=comm_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, ev_ms = float(t[0]), float(t[1])
        return wall, ev_ms

    # Settling (untimed, not part of W).  Measured (tools/probe/cold_start_probe.py, slow_start_probe*.py): the clocks of a cold
    # GPU settle over ~0.1 s of load, and the first two or three launches that follow an upload (pnp_set_batch) run 1.5x slower
    # from start to end however long they are.  The number reported is the sustained rate of a trajectory in flight: 0.3 s of load
    # on a throw-away trajectory, then the state is uploaded and four 8-step launches take the slow launches; the W warmup steps
    # and the K timed steps follow on the same trajectory.  (The reference's lagged-potential integrator does not survive
    # long runs on this workload -- lanes turn NaN after ~3000 steps, tools/probe/long_trajectory_probe.py, in the oracle too --
    # hence the fresh upload; lanes_ok reports the state the timed steps leave behind.)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.3:
        solver.step(512, args.steps_per_launch)
        solver.synchronize()
    solver.set_batch(c0, pb, vz, fl)
    for _ in range(4):
        solver.step(8, 8)
    solver.step(args.warmup, args.steps_per_launch)
    wall, ev_ms = timed(args.steps, args.steps_per_launch)
    status = solver.get_status()
    n_launch = (args.steps + args.steps_per_launch - 1) // args.steps_per_launch
    steps_total = world * B * args.steps
    value = steps_total / wall

    # one launch per timestep (state read from and written to HBM by every launch) -- reported next to the headline
    fused = None
    if not args.no_fused and args.steps_per_launch != 1:
        solver.step(64, 1)
        fw, fe = timed(args.steps, 1)
        lsec = fe * 1e-3 / args.steps
        fused = {'timesteps_per_s': world * B * args.steps / fw, 'ms_per_step': fw / args.steps * 1e3,
                 'steps_per_launch': 1, 'launch_us': lsec * 1e6,
                 'frac': 16.0 * (N + 1) * nx * B / lsec / 1e9 / HBM_PEAK_GBS}

    # the only exchange of the path: gather the polarization observables (RCCL all_gather over xGMI)
    cs, vs, es = solver.get_surface()
    gather_ms = None
    if dist is not None:
        from catint_amd.parallel import gather_observables
        obs = np.concatenate([cs, vs[:, None], es[:, None]], axis=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        curve = gather_observables(obs, world * B, dist, device=comm_dev)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t0) * 1e3
        assert curve.shape == (world * B, N + 2)
    solver.close()

    # oversubscribed regime (not the headline): same lanes-shape, many more of them than SIMDs
    large = None
    if world == 1 and args.large_batch > 0:
        LB = args.large_batch
        lp, lc0, lpb, lvz, lfl = make_batch(LB, N, nx, seed=77, phi_max=0.025, dt_factor=1e-5)
        s2 = solver_from_problem(lp, args.method, batch_capacity=LB, device=device)
        s2.set_batch(lc0, lpb, lvz, lfl)
        s2.step(64, 1)
        ls = max(10, min(args.steps, 50))
        s2.synchronize()
        s2.timer_start()
        s2.step(ls, 1)
        lms = s2.timer_stop()
        lok = int((s2.get_status() == 0).sum())
        s2.close()
        lsec = lms * 1e-3 / ls
        large = {'batch': LB, 'steps': ls, 'timesteps_per_s': LB / lsec, 'launch_us': lsec * 1e6,
                 'achieved_GBs': 16.0 * (N + 1) * nx * LB / lsec / 1e9,
                 'frac': 16.0 * (N + 1) * nx * LB / lsec / 1e9 / HBM_PEAK_GBS, 'lanes_ok': lok, 'steps_per_launch': 1}
        # the same batch in fused launches (like the headline)
        s3 = solver_from_problem(lp, args.method, batch_capacity=LB, device=device)
        s3.set_batch(lc0, lpb, lvz, lfl)
        for _ in range(4):        # past the slow launches that follow an upload (see above)
            s3.step(8, 8)
        s3.step(args.steps_per_launch, args.steps_per_launch)
        s3.synchronize()
        s3.timer_start()
        s3.step(args.steps, args.steps_per_launch)
        fms = s3.timer_stop()
        s3.close()
        large['fused'] = {'timesteps_per_s': LB * args.steps / (fms * 1e-3), 'steps_per_launch': args.steps_per_launch,
                          'frac': 16.0 * (N + 1) * nx * LB * args.steps / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    if rank == 0:
        # 2*8*(N+1)*nx bytes per lane-timestep; a launch advances B lanes by steps_per_launch steps
        alg_bytes_per_launch = 16.0 * (N + 1) * nx * B * args.steps / n_launch
        launch_s = ev_ms * 1e-3 / n_launch
        achieved = alg_bytes_per_launch / launch_s / 1e9
        out = {
            'metric': 'batched 1D PNP Newton-timesteps/sec', 'value': value, 'unit': 'timesteps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': wall / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: batch=%d operating points/GPU, %d species, %d grid points, fp64, %s '
                                   'compat integrator, Dirichlet-Dirichlet Poisson' % (B, N, nx, args.method),
                       'batch_per_gpu': B, 'nspecies': N, 'nx': nx, 'steps_per_launch': args.steps_per_launch,
                       'parallelism': 'batch-sharded x%d' % world},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                         'kernel': 'pnp::step_kernel* (P=%d points/lane)' % next((P for P in (1, 2, 4, 8, 16) if nx - 2 <= 64 * P), 16),
                         'launch_us': launch_s * 1e6, 'algorithmic_bytes_per_launch': alg_bytes_per_launch,
                         'timesteps_per_launch': args.steps / n_launch},
            'lanes_ok': int((status == 0).sum()), 'lanes_total': int(B),
        }
        # measured HBM bytes per launch come from a separate rocprofv3 --pmc pass of this same command
        # (profiles/, MI355X_MICROARCH.md HBM section: FETCH_SIZE doubled on gfx950); null when absent
        try:
            tj = json.load(open(args.traffic_json))
            # only for the launch shape the counters were collected on (same lanes, same fused launch length)
            if (tj.get('batch'), tj.get('nspecies'), tj.get('nx'), tj.get('steps_per_launch')) == \
                    (B, N, nx, args.steps_per_launch) and tj.get('method') == args.method and \
                    args.steps == n_launch * args.steps_per_launch and world == 1:
                out['roofline']['traffic'] = tj['hbm_bytes_per_launch']
                out['roofline']['traffic_source'] = tj.get('source')
                if tj.get('sq_insts_valu_per_wave'):
                    # fp64 vector utilisation next to the HBM figure.  Every VALU instruction of the kernel counted as
                    # a 64-lane fp64 operation (upper bound; SQ_INSTS_VALU of the same rocprofv3 run) against 256 CUs x 4 SIMDs x
                    # 16 fp64 lanes per cycle x 2.4 GHz
                    lane_ops = tj['sq_insts_valu_per_wave'] / args.steps_per_launch * tj.get('waves_per_lane', 1.0) * 64.0
                    out['roofline']['fp64_valu_util'] = lane_ops * (B / launch_s * args.steps / n_launch) / (256 * 4 * 16 * 2.4e9)
        except Exception:
            pass
        if fused:
            out['per_step_launch'] = fused
        if large:
            out['large_batch'] = large
        if world == 1 and args.physical_steps > 0:
            out['physical_mode'] = physical_mode(args, device, not args.no_cpu_baseline)
        if gather_ms is not None:
            out['gather_ms'] = gather_ms
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(prob, c0, pb, vz, fl, args.method, args.cpu_seconds)
            if args.method == 'Crank-Nicolson':
                cb['reference_faithful_dense_1core'] = cpu_reference_faithful(prob, c0, pb, vz, fl, args.method)
            out['cpu_baseline'] = cb
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
