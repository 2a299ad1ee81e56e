#!/usr/bin/env python3
"""Benchmark of the batched 1D PNP timestep path (BASELINE.json metric:
"batched 1D PNP Newton-timesteps/sec at 1/2/4/8 GPU; achieved HBM GB/s vs peak").

One "step" = one pass of the integrator's time-loop body (reference catint/calculator_old.py:512-558) over one batch of
B operating points.  Workload at N=1: BASELINE.json configs[1] -- batch=1024 operating points, 3 species, 512 grid points,
fp64.  The K timed steps are issued as pnp_step(K, steps_per_launch) (default: the library's fused launches, every step's
state written to HBM), bracketed by barrier + synchronize on both sides; `value` is lane-timesteps per WALL second of that
region.  `roofline.achieved` = SURVEY 8(d)'s algorithmic bytes 16(N+1)nx per lane-timestep x the lane-timesteps of one
launch / that launch's duration by HIP events on the library's own stream.

What the line says about itself:
  * `roofline.traffic` is MEASURED IN THIS RUN: before the GPU is touched, rank 0 runs this same script as a child under
    `rocprofv3 --pmc` (separate passes for FETCH_SIZE, WRITE_SIZE and an SQ group, as MI355X_MICROARCH.md prescribes; FETCH_SIZE
    doubled on gfx950) over the same launch shapes, and reads the per-dispatch counters of the timed kernels.  null (with a
    reason) if rocprofv3 is unavailable.  `roofline.traffic_over_algorithmic` << 1 means the state stayed on chip between fused
    steps: such a launch is NOT HBM-bound, and `roofline.limiter` names what the SQ counters say binds instead.
  * `per_step_launch`: one launch per timestep on the same batch (traffic == algorithmic bytes);
  * `beyond_cache`: one GPU's share of BASELINE configs[3] (32768 lanes x 6 species x 1024 points, 2.1 GB of state, far beyond
    the 256 MiB Infinity Cache), one launch per step and fused -- the configuration in which the HBM roofline is the real bound;
  * `first_launch_after_upload`: pnp_set_batch immediately followed by the first K-step launch (what a cold caller sees);
  * `physical_mode`: the implicit coupled-Newton path with a measured roofline of its own (fp64 VALU issue, LDS, HBM bytes).
Multi-GPU: the batch shards embarrassingly (weak scaling: every rank owns `--batch` lanes), no collective in the timed region; one
RCCL all_gather of the polarization observables afterwards.

Prints ONE JSON line on rank 0.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK = 256 * 4 * 16 * 2.4e9      # fp64 lane-operations/s: 256 CUs x 4 SIMDs x 16 fp64 lanes/clk x 2.4 GHz (78.6 TFLOP/s FMA)
BC_SHAPE = (32768, 6, 1024)    # one GPU's share of BASELINE configs[3]
C4_SHAPE = (8192, 8, 4096)     # one GPU's share of BASELINE configs[4]
RADII8 = [4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 3e-10, 3e-10, 4.5e-10, 3.5e-10]


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
    ap.add_argument('--no-extras', action='store_true', help='headline only (no per-step / beyond-cache / physical records)')
    ap.add_argument('--shares-only', action='store_true',
                    help='with a process group: the per-rank configs[3] / configs[4] share records, but none of the single-GPU extras')
    ap.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 counter passes (roofline.traffic = null)')
    ap.add_argument('--large-batch', type=int, default=8192, help='extra timed run at this batch size; 0 = skip')
    ap.add_argument('--physical-steps', type=int, default=20, help='implicit physical mode on the headline shape; 0 = skip')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)   # this process is the profiled child
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------------------------
# workloads (shared by the timed parent and the profiled child so that both launch the same kernels on the same shapes)
# ------------------------------------------------------------------------------------------------------------------------------------
def compat_solver(B, N, nx, method, seed, device=0):
    from catint_amd.synthetic import make_batch
    from catint_amd.host import solver_from_problem
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=seed, phi_max=0.025, dt_factor=1e-5)
    s = solver_from_problem(prob, method, batch_capacity=B, device=device)
    return s, (prob, c0, pb, vz, fl)


def newton_solver(B, N, nx, seed, device=0, steric=False, error_estimate=False, predictor=False, time_order=1, records=None):
    from catint_amd import _capi
    from catint_amd.synthetic import make_batch
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=seed, phi_max=0.2, dt_factor=0.1)
    pb = np.nan_to_num(pb)
    s = _capi.PnpSolver(N, nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton', batch_capacity=B, device=device)
    if steric:
        s.set_newton(wall_bc='stern', stern_capacitance=0.2, tol=1e-8, mpb_radius=RADII8[:N], error_estimate=error_estimate,
                     predictor=predictor, time_order=time_order)
    else:
        s.set_newton(tol=1e-8, error_estimate=error_estimate, predictor=predictor, time_order=time_order)
    if records:
        s.set_option('LANE_RECORDS', records)
    return s, (prob, c0, pb, vz, fl)


def pmc_child(args):
    """The profiled child: the launch shapes of every record, separated by marker dispatches (surface_kernel via get_surface)."""
    B, N, nx = args.batch, args.nspecies, args.nx

    def item(s, inputs, launches):
        s.set_batch(*inputs[1:])
        launches(s, True)          # warm-up
        s.get_surface()            # marker: timed launches follow
        launches(s, False)
        s.get_surface()            # marker: end
        s.close()

    s, inp = compat_solver(B, N, nx, args.method, 1000)
    item(s, inp, lambda s, w: s.step(args.steps, args.steps_per_launch))
    s, inp = compat_solver(B, N, nx, args.method, 1000)
    item(s, inp, lambda s, w: s.step(8, 1))
    if not args.no_extras:
        s, inp = compat_solver(*BC_SHAPE, args.method, 55)
        item(s, inp, lambda s, w: s.step(4, 1))
        s, inp = compat_solver(*BC_SHAPE, args.method, 55)
        item(s, inp, lambda s, w: s.step(32, 32))
        if args.physical_steps > 0:
            s, inp = newton_solver(B, N, nx, 4242)
            item(s, inp, lambda s, w: s.step(3))
            s, inp = newton_solver(8192, 8, 512, 4444, steric=True)
            item(s, inp, lambda s, w: s.step(2))
            s, inp = newton_solver(32768, 8, 512, 4446, steric=True)
            item(s, inp, lambda s, w: s.step(6))
            s, inp = newton_solver(BC_SHAPE[0], BC_SHAPE[1], BC_SHAPE[2], 4447, steric=True)
            item(s, inp, lambda s, w: s.step(4))
            s, inp = newton_solver(C4_SHAPE[0], C4_SHAPE[1], C4_SHAPE[2], 4448, steric=True)
            item(s, inp, lambda s, w: s.step(2))
        if args.large_batch > 0:
            s, inp = compat_solver(args.large_batch, N, nx, args.method, 77)
            item(s, inp, lambda s, w: s.step(8, 1))


PMC_ITEMS = ['headline', 'per_step_launch', 'beyond_cache_per_step', 'beyond_cache_fused', 'physical_pair', 'physical_sweep',
             'physical_lane_32k', 'physical_lane_config3', 'physical_lane_config4', 'large_batch']


def pmc_items(args):
    """The records of this run that have a segment in the profiled child, in the child's order (pmc_child)."""
    names = ['headline', 'per_step_launch']
    if not args.no_extras:
        names += ['beyond_cache_per_step', 'beyond_cache_fused']
        if args.physical_steps > 0:
            names += ['physical_pair', 'physical_sweep', 'physical_lane_32k', 'physical_lane_config3', 'physical_lane_config4']
        if args.large_batch > 0:
            names += ['large_batch']
    return names


PMC_GROUPS = [['FETCH_SIZE'], ['WRITE_SIZE'],
              ['SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY',
               'SQ_ACTIVE_INST_VALU'],
              # fp64 arithmetic counted apart from the other vector instructions (moves, selects, accumulator-register copies, integer)
              ['SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_TRANS_F64']]


def collect_pmc(args):
    """rocprofv3 --pmc passes over the child (run BEFORE this process touches the GPU).  Returns {item: {...}} or {'error': str}."""
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return {'error': 'rocprofv3 not found'}
    tmp = tempfile.mkdtemp(prefix='catint_pmc_', dir='/tmp')
    child = [sys.executable, os.path.abspath(__file__), '--pmc-child', '--batch', str(args.batch), '--nspecies', str(args.nspecies),
             '--nx', str(args.nx), '--method', args.method, '--steps', str(args.steps), '--steps-per-launch', str(args.steps_per_launch),
             '--physical-steps', str(args.physical_steps), '--large-batch', str(args.large_batch)] + (['--no-extras'] if args.no_extras else [])
    # (counters are per dispatch: the child keeps every step's rows in ONE dispatch -- no row chunks on separate streams -- so that a
    # dispatch's bytes are a timestep's bytes; the profiler serialises dispatches anyway)
    env = dict(os.environ, TMPDIR='/tmp', CATINT_PNP_STEP_STREAMS='1')
    out = {}
    try:
        for gi, grp in enumerate(PMC_GROUPS):
            d = os.path.join(tmp, 'g%d' % gi)
            r = subprocess.run([exe, '--pmc'] + grp + ['--output-format', 'csv', '-d', d, '--'] + child, cwd='/tmp', env=env,
                               capture_output=True, text=True, timeout=420)
            files = glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not files:
                if gi >= 3:          # the optional fp64 split: report what the other passes gave
                    for rec in out.values():
                        rec['fp64_split_error'] = 'rocprofv3 pass %s failed (rc %d)' % (grp[0], r.returncode)
                    continue
                return {'error': 'rocprofv3 pass %s failed (rc %d): %s' % (grp[0], r.returncode, (r.stderr or r.stdout)[-300:])}
            rows = list(csv.DictReader(open(files[0])))
            disp = {}
            for row in rows:                       # one row per (dispatch, counter)
                dd = disp.setdefault(int(row['Dispatch_Id']), {'kernel': row['Kernel_Name'], 'grid': int(row['Grid_Size']),
                                                                 'wg': int(row['Workgroup_Size']), 'vgpr': int(row.get('VGPR_Count', 0) or 0),
                                                                 'scratch': int(row.get('Scratch_Size', 0) or 0), 'c': {}})
                dd['c'][row['Counter_Name']] = dd['c'].get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
            seq = [disp[k] for k in sorted(disp)]
            # segments: marker (surface_kernel) ... timed launches ... marker
            marks = [i for i, dd in enumerate(seq) if 'surface_kernel' in dd['kernel']]
            for it, name in enumerate(pmc_items(args)):
                if 2 * it + 1 >= len(marks):
                    break
                seg = seq[marks[2 * it] + 1:marks[2 * it + 1]]
                seg = [dd for dd in seg if 'step_kernel' in dd['kernel'] or 'newton' in dd['kernel']]
                if not seg:
                    continue
                main = max(set(dd['kernel'] for dd in seg), key=lambda k: sum(1 for dd in seg if dd['kernel'] == k))
                seg = [dd for dd in seg if dd['kernel'] == main]
                rec = out.setdefault(name, {'kernel': main.split('(')[0].replace('void ', ''), 'grid_threads': seg[0]['grid'],
                                            'workgroup': seg[0]['wg'], 'vgpr': seg[0]['vgpr'], 'scratch_bytes': seg[0]['scratch'],
                                            'launches': len(seg)})
                for cname in grp:
                    vals = sorted(dd['c'].get(cname, 0.0) for dd in seg)
                    rec[cname] = vals[len(vals) // 2]
    except Exception as e:       # never let the counters break the headline
        return {'error': '%s: %s' % (type(e).__name__, e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for rec in out.values():
        if 'FETCH_SIZE' in rec and 'WRITE_SIZE' in rec:
            # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> x2.
            # The guide establishes the factor for 16-byte-per-lane streaming reads: that is what the compat kernels (buffer_load b128
            # windows) and the lane kernel (16-byte pairs) issue.  The pair / lane-team Newton kernels read 8 bytes per lane
            # (uncalibrated): for them the x1 figure is kept beside the x2 one as a lower bound.
            rec['hbm_bytes_per_launch'] = (2.0 * rec['FETCH_SIZE'] + rec['WRITE_SIZE']) * 1024.0
            rec['hbm_bytes_per_launch_fetch_x1'] = (rec['FETCH_SIZE'] + rec['WRITE_SIZE']) * 1024.0
            rec['fetch_correction'] = 'x2 (16-byte lanes)' if ('step_kernel' in rec['kernel'] or 'newton_lane' in rec['kernel']) else 'x1 ... x2 (8-byte lanes: uncalibrated)' 
    return out


def limiter_from_sq(rec):
    """What the SQ counters of a launch say binds it (shares of wave lifetime; quad-cycle counters, ratios are unit-free)."""
    if not rec or 'SQ_WAVE_CYCLES' not in rec or rec['SQ_WAVE_CYCLES'] <= 0:
        return None
    wc = rec['SQ_WAVE_CYCLES']
    waves = max(rec.get('SQ_WAVES', 0.0), 1.0)
    busy = rec.get('SQ_BUSY_CYCLES', 0.0)
    d = {'wait_any_share': rec.get('SQ_WAIT_ANY', 0.0) / wc, 'wait_inst_share': rec.get('SQ_WAIT_INST_ANY', 0.0) / wc,
         'valu_active_share_per_wave': rec.get('SQ_ACTIVE_INST_VALU', 0.0) / wc,
         'valu_insts_per_wave': rec.get('SQ_INSTS_VALU', 0.0) / waves, 'lds_insts_per_wave': rec.get('SQ_INSTS_LDS', 0.0) / waves}
    # the largest share of a wave's lifetime names the limiter (shares of parked / issue-stalled / issuing-VALU wave cycles)
    d['binds'] = max([(d['wait_any_share'], 'waves parked in s_waitcnt (memory / LDS latency)'),
                      (d['wait_inst_share'], 'issue stalls (dependent fp64 chains, busy pipes)'),
                      (d['valu_active_share_per_wave'], 'VALU issue')])[1]
    if busy > 0:
        d['sq_busy_cycles'] = busy
    return d


# ------------------------------------------------------------------------------------------------------------------------------------
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


def timed_steps(s, nsteps, spl, reps=1):
    """HIP-event time (ms) of pnp_step(nsteps, spl) on the library's stream, median over reps."""
    ms = []
    for _ in range(reps):
        s.timer_start()
        s.step(nsteps, spl)
        ms.append(s.timer_stop())
    return float(np.median(ms))


def hbm_record(B, N, nx, steps, launches, ev_ms, pmc):
    """Record of a timed compat run: algorithmic bytes, per-launch event time, measured traffic when the counters are there."""
    alg = 16.0 * (N + 1) * nx * B * steps / launches
    launch_s = ev_ms * 1e-3 / launches
    rec = {'timesteps_per_s': B * steps / (ev_ms * 1e-3), 'launch_us': launch_s * 1e6, 'timesteps_per_launch': steps / launches,
           'algorithmic_bytes_per_launch': alg, 'achieved_GBs': alg / launch_s / 1e9, 'frac': alg / launch_s / 1e9 / HBM_PEAK_GBS,
           'traffic': None}
    if pmc and 'hbm_bytes_per_launch' in pmc:
        rec['traffic'] = pmc['hbm_bytes_per_launch']
        rec['traffic_over_algorithmic'] = pmc['hbm_bytes_per_launch'] / alg
        rec['hbm_GBs_measured'] = pmc['hbm_bytes_per_launch'] / launch_s / 1e9
        rec['hbm_frac_measured'] = rec['hbm_GBs_measured'] / HBM_PEAK_GBS
        rec['kernel'] = pmc.get('kernel')
        rec['kernel_resources'] = {k: pmc.get(k) for k in ('grid_threads', 'workgroup', 'vgpr', 'scratch_bytes')}
        lim = limiter_from_sq(pmc)
        if lim:
            rec['sq'] = lim
    return rec


def physical_mode(args, device, with_cpu, pmc, warm=lambda: None):
    """Implicit physical mode (PNP_METHOD_NEWTON) with SURVEY 8(d)'s synthetic inputs: phiM ~ U(-0.2, 0.2) V, dt = 0.1 lambda_D L / D_max,
    Newton to a scaled update of 1e-8.  Not HBM-bound: its roofline is fp64 VALU issue (and LDS exchange); measured per kernel."""
    B, N, nx = args.batch, args.nspecies, args.nx

    def newton_roofline(rec, iters_per_s, lanes, alg_bytes_per_iter):
        if not rec or 'SQ_INSTS_VALU' not in rec:
            return None
        # per launch: SQ_INSTS_VALU wave-instructions x 64 lanes; upper bound on fp64 lane-operations (integer/select VALU included)
        lane_ops = rec['SQ_INSTS_VALU'] * 64.0
        out = {'bound': 'fp64 VALU issue', 'kernel': rec.get('kernel'), 'valu_wave_insts_per_launch': rec['SQ_INSTS_VALU'],
               'lds_wave_insts_per_launch': rec.get('SQ_INSTS_LDS'), 'peak_fp64_lane_ops_per_s': FP64_VALU_PEAK,
               'kernel_resources': {k: rec.get(k) for k in ('grid_threads', 'workgroup', 'vgpr', 'scratch_bytes')}}
        if 'hbm_bytes_per_launch' in rec:
            out['hbm_bytes_per_launch'] = rec['hbm_bytes_per_launch']
        lim = limiter_from_sq(rec)
        if lim:
            out['sq'] = lim
        return out, lane_ops

    s, (prob, c0, pb, vz, fl) = newton_solver(B, N, nx, 4242, device)
    s.set_batch(c0, pb, vz, fl)
    s.step(5)
    s.synchronize()
    warm()
    ms = timed_steps(s, args.physical_steps, 0)
    it = s.newton_iterations()
    ok = int((s.get_status() == 0).sum())
    # the launch shape the counters were collected on: 3 steps
    s.set_batch(c0, pb, vz, fl)
    s.step(3)
    ms3 = timed_steps(s, 3, 0)
    it3 = float(s.newton_iterations().sum())
    s.close()
    sec = ms * 1e-3
    out = {'workload': 'batch=%d, %d species, %d points, backward Euler dt=%.3g s, Dirichlet wall, tol 1e-8' % (B, N, nx, prob.dt),
           'timesteps_per_s': B * args.physical_steps / sec, 'newton_iterations_per_s': float(it.sum()) / sec,
           'mean_newton_iterations_per_step': float(it.sum()) / (B * args.physical_steps),
           'ms_per_step': ms / args.physical_steps, 'lanes_ok': ok}
    r = newton_roofline(pmc.get('physical_pair') if isinstance(pmc, dict) else None, 0, B, 0)
    if r:
        roof, lane_ops = r
        roof['achieved_fp64_lane_ops_per_s'] = lane_ops / (ms3 * 1e-3)
        roof['frac'] = roof['achieved_fp64_lane_ops_per_s'] / FP64_VALU_PEAK
        roof['valu_wave_insts_per_lane_iteration'] = roof['valu_wave_insts_per_launch'] / max(it3, 1.0)
        if 'hbm_bytes_per_launch' in roof:
            roof['hbm_bytes_per_lane_iteration'] = roof['hbm_bytes_per_launch'] / max(it3, 1.0)
            pp = pmc.get('physical_pair', {})
            if 'hbm_bytes_per_launch_fetch_x1' in pp:      # 8-byte lanes: the gfx950 FETCH_SIZE factor is uncalibrated there
                roof['hbm_bytes_per_lane_iteration_range'] = [pp['hbm_bytes_per_launch_fetch_x1'] / max(it3, 1.0), roof['hbm_bytes_per_lane_iteration']]
                roof['fetch_correction'] = pp.get('fetch_correction')
            roof['algorithmic_state_bytes_per_lane_timestep'] = 16.0 * (N + 1) * nx
        out['roofline'] = roof
    try:
        s, inp = newton_solver(B, N, nx, 4242, device)
        s.set_newton(tol=1e-8, error_estimate=True)
        s.set_batch(*inp[1:])
        s.step(5)
        s.synchronize()
        warm()
        ms_e = timed_steps(s, args.physical_steps, 0)
        it_e = s.newton_iterations()
        ok_e = int((s.get_status() == 0).sum())
        s.close()
        out['with_error_estimate'] = {'timesteps_per_s': B * args.physical_steps / (ms_e * 1e-3),
                                      'mean_newton_iterations_per_step': float(it_e.sum()) / (B * args.physical_steps), 'lanes_ok': ok_e}
    except Exception as e:
        out['with_error_estimate'] = {'error': str(e)}
    # BASELINE configs[4] shape in the same mode: 8 species (size-modified, Stern wall), 4096 points
    try:
        LB = max(64, min(1024, B))
        s8, inp = newton_solver(LB, 8, 4096, 4343, device, steric=True)
        s8.set_batch(*inp[1:])
        s8.step(1)
        s8.synchronize()
        warm()
        ms8 = timed_steps(s8, 3, 0)
        it8 = s8.newton_iterations()
        ok8 = int((s8.get_status() == 0).sum())
        s8.close()
        out['config4_shape'] = {'workload': 'batch=%d, 8 species size-modified, 4096 points, Stern wall' % LB,
                                'timesteps_per_s': LB * 3 / (ms8 * 1e-3), 'newton_iterations_per_s': float(it8.sum()) / (ms8 * 1e-3),
                                'lanes_ok': ok8}
    except Exception as e:      # never let the extra line break the headline
        out['config4_shape'] = {'error': str(e)}
    # large batches of large blocks: the lane kernel (pnp_lane.hip: one operating point per lane, block Thomas from both ends in
    # registers).  Its roofline is HBM: the block-Thomas records (N+1)(N+2) doubles per grid row are written by the forward pass and
    # read by the back-substitution of EVERY Newton iteration; algorithmic bytes per lane-iteration = 8 nx (2 (N+1)(N+2) + 6 N + 5).
    def lane_bytes(N_, nx_, fused=False):
        # per grid row and Newton iteration, in doubles: records (N+1)(N+2) written by the forward pass and read by the back-substitution;
        # state: forward reads c, phi, c_old (2N+1); then either the Newton update is written (N+1), read back with the state (2N+2) and
        # the new state written (N+1) -- three passes, 6N+5 -- or (the lane kernel on timesteps: update fused into the
        # back-substitution, two state copies) the state is read once more and written once: 4N+3
        return 8.0 * nx_ * (2 * (N_ + 1) * (N_ + 2) + (4 * N_ + 3 if fused else 6 * N_ + 5))

    def lane_record(LB, LN, LX, seed, steps, pmc_key, what, pmc_steps=2):
        s8, inp = newton_solver(LB, LN, LX, seed, device, steric=True)
        s8.set_batch(*inp[1:])
        s8.step(1)
        s8.synchronize()
        # where the multi-GB workspace of a lane kernel lies decides which of a few discrete rates an HBM-bound launch runs at (DESIGN.md
        # section 7b): up to six placements are tried on two timesteps each, the fastest stays (pnp_tune_placement; the state is put back)
        placement = s8.tune_placement(2, 6)
        warm()
        ms8 = timed_steps(s8, steps, 0)
        it8 = s8.newton_iterations()
        ok8 = int((s8.get_status() == 0).sum())
        s8.set_batch(*inp[1:])
        s8.step(pmc_steps)
        ms2 = timed_steps(s8, pmc_steps, 0)        # the launch shape the counters were collected on (pmc_child)
        it2 = float(s8.newton_iterations().sum())
        fused = s8.default_family() == 'lane+fused'          # (pnp_lane.hip: launch_lane_nb)
        s8.close()
        its = float(it8.sum())
        alg = lane_bytes(LN, LX, fused)
        rec = {'workload': 'batch=%d, %d species size-modified, %d points, Stern wall, backward Euler: %s' % (LB, LN, LX, what),
               'timesteps_per_s': LB * steps / (ms8 * 1e-3), 'newton_iterations_per_s': its / (ms8 * 1e-3),
               'mean_newton_iterations_per_step': its / (LB * steps), 'ms_per_step': ms8 / steps, 'lanes_ok': ok8,
               'workspace_placement_trials_ms_per_step': [round(v, 3) for v in placement]}
        roof = {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'algorithmic_bytes_per_lane_iteration': alg,
                'achieved': alg * its / (ms8 * 1e-3) / 1e9, 'traffic': None,
                'achieved_is': 'algorithmic bytes per Newton iteration and operating point (state + block-Thomas records, written by the '
                               'forward pass and read by the back-substitution) x iterations / HIP-event time of the launch'}
        roof['frac'] = roof['achieved'] / HBM_PEAK_GBS
        # SURVEY 8(d)'s own yardstick beside the design floor: 16 (N+1) nx bytes per lane-TIMESTEP (state read and written once), as if the
        # block-Thomas records never left the chip -- they do by design (a direct block solve of 9 x 9 blocks over 512 rows holds 47 KB
        # of records per operating point), which is why the design-floor fraction above is the one the kernel is tuned against
        roof['survey_8d_bytes_per_lane_timestep'] = 16.0 * (LN + 1) * LX
        roof['survey_8d_frac'] = 16.0 * (LN + 1) * LX * rec['timesteps_per_s'] / 1e9 / HBM_PEAK_GBS
        roof['update_fused_into_back_substitution'] = bool(fused)
        pr = pmc.get(pmc_key) if isinstance(pmc, dict) else None
        if pr and 'hbm_bytes_per_launch' in pr:
            roof['kernel'] = pr.get('kernel')
            roof['kernel_resources'] = {k: pr.get(k) for k in ('grid_threads', 'workgroup', 'vgpr', 'scratch_bytes')}
            roof['traffic'] = pr['hbm_bytes_per_launch'] / max(it2, 1.0)           # per lane-iteration, like the algorithmic figure
            roof['traffic_over_algorithmic'] = roof['traffic'] / alg
            roof['hbm_GBs_measured'] = pr['hbm_bytes_per_launch'] / (ms2 * 1e-3) / 1e9
            roof['hbm_frac_measured'] = roof['hbm_GBs_measured'] / HBM_PEAK_GBS
            roof['traffic_note'] = ('a wave iterates until the slowest of its 32 operating points has finished its timesteps; lanes that '
                                    'finished earlier still stream their records, so measured bytes per LANE-iteration exceed the '
                                    'algorithmic figure by the spread of the iteration counts (2-step launch: ~1.15x, long launches less)')
            lim = limiter_from_sq(pr)
            if lim:
                roof['sq'] = lim
            if 'SQ_INSTS_VALU' in pr:
                roof['valu_wave_insts_per_lane_iteration'] = pr['SQ_INSTS_VALU'] / max(it2, 1.0)
            f64 = [pr.get(k) for k in ('SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_TRANS_F64')]
            if all(v is not None for v in f64) and 'SQ_INSTS_VALU' in pr:
                n64 = float(sum(f64))
                roof['fp64_valu'] = {'fp64_wave_insts_per_launch': n64, 'share_of_valu_insts': n64 / max(pr['SQ_INSTS_VALU'], 1.0),
                                     'fma_share_of_fp64': f64[2] / max(n64, 1.0),
                                     # lane-operations: 64 per wave instruction, two flops per fused multiply-add
                                     'fp64_lane_ops_per_s': n64 * 64.0 / (ms2 * 1e-3), 'frac_of_fp64_vector_peak': n64 * 64.0 / (ms2 * 1e-3) / FP64_VALU_PEAK}
            elif pr.get('fp64_split_error'):
                roof['fp64_valu'] = {'error': pr['fp64_split_error']}
        rec['roofline'] = roof
        return rec

    def f32_records(key, EB, EN, EX, seed, est):
        """The lane kernel with the COLUMNS of its block-Thomas records in single precision (option LANE_RECORDS = f32; t, the elimination
        and the residual stay double: the Newton update is inexact to ~1e-7, the converged state is the same to rounding) -- opt-in."""
        sub = {}
        for tag, ee in (('default_stopping_rule', False), ('with_error_estimate', True)):
            sf, inpf = newton_solver(EB, EN, EX, seed, device, steric=True, error_estimate=ee, records='f32')
            sf.set_batch(*inpf[1:])
            sf.step(1)
            sf.synchronize()
            sf.tune_placement(2, 6)
            warm()
            msf = timed_steps(sf, est, 0)
            itf = sf.newton_iterations()
            okf = int((sf.get_status() == 0).sum())
            sf.close()
            del inpf
            sub[tag] = {'timesteps_per_s': EB * est / (msf * 1e-3), 'mean_newton_iterations_per_step': float(itf.sum()) / (EB * est),
                        'lanes_ok': okf}
        sub['note'] = ('opt-in: 139 instead of 215 doubles per grid row and Newton iteration through HBM; same iteration counts in all but '
                       '~0.4 % of the operating points (one more iteration), final states equal to 1e-15 (tools/probe/f32_records_probe.py)')
        if isinstance(out.get(key), dict) and 'error' not in out[key]:
            out[key]['with_f32_record_columns'] = sub

    try:
        out['large_batch_8_species'] = lane_record(8192, 8, 512, 4444, 20, 'physical_sweep',
                                                   'lane-quad kernel, eight lanes per operating point: 1024 waves, one per SIMD')
        out['large_batch_8_species_32k'] = lane_record(32768, 8, 512, 4446, 20, 'physical_lane_32k',
                                                       'lane kernel, one wave per SIMD (1024 waves); 20 timesteps in one launch', pmc_steps=6)
        # the same with the quadratic error estimate as stopping rule (pnp_newton_params.error_estimate: saves the iteration that only
        # confirms convergence; the default rule is the one every other record uses)
        for key, EB, seed in (('large_batch_8_species', 8192, 4444), ('large_batch_8_species_32k', 32768, 4446)):
            se, inpe = newton_solver(EB, 8, 512, seed, device, steric=True, error_estimate=True)
            se.set_batch(*inpe[1:])
            se.step(1)
            se.synchronize()
            se.tune_placement(2, 6)
            warm()
            mse = timed_steps(se, 20, 0)
            ite = se.newton_iterations()
            oke = int((se.get_status() == 0).sum())
            se.close()
            del inpe
            out[key]['with_error_estimate'] = {'timesteps_per_s': EB * 20 / (mse * 1e-3),
                                               'mean_newton_iterations_per_step': float(ite.sum()) / (EB * 20), 'lanes_ok': oke}
        f32_records('large_batch_8_species_32k', 32768, 8, 512, 4446, 20)
    except Exception as e:
        out['large_batch_8_species'] = {'error': str(e)}
    try:      # one GPU's share of BASELINE configs[3] in the coupled-Newton mode
        out['config3_share'] = lane_record(BC_SHAPE[0], BC_SHAPE[1], BC_SHAPE[2], 4447, 10, 'physical_lane_config3',
                                           "one GPU's share of configs[3] (262144 points over 8 GPUs); 10 timesteps in one launch", pmc_steps=4)
        f32_records('config3_share', BC_SHAPE[0], BC_SHAPE[1], BC_SHAPE[2], 4447, 10)
    except Exception as e:
        out['config3_share'] = {'error': str(e)}
    try:      # one GPU's share of BASELINE configs[4]: 8192 lanes x 8 species x 4096 points (24 GB of records)
        out['config4_share'] = lane_record(8192, 8, 4096, 4448, 4, 'physical_lane_config4', "one GPU's share of configs[4] (65536 points over 8 GPUs)")
    except Exception as e:
        out['config4_share'] = {'error': str(e)}
    try:      # the kernel-family thresholds checked against a measurement on THIS device (pnp_autotune: every family, 6 timed steps each)
        sel = {}
        for SB, SN, SX, seed in ((1024, 8, 512, 4445), (8192, 8, 512, 4444), (32768, 8, 512, 4446), BC_SHAPE + (4447,)):
            sa, inpa = newton_solver(SB, SN, SX, seed, device, steric=True)
            sa.set_batch(*inpa[1:])
            del inpa
            sa.step(1)
            sa.synchronize()
            lib_choice = sa.default_family()
            fastest, fam_ms = sa.autotune(6)
            sa.close()
            sel['%dx%dx%d' % (SB, SN, SX)] = {'ms_per_timestep': fam_ms, 'fastest': fastest, 'library_choice': lib_choice,
                                              'library_choice_over_fastest': fam_ms[lib_choice] / fam_ms[fastest]}
        out['kernel_selection_measured'] = sel
    except Exception as e:
        out['kernel_selection_measured'] = {'error': str(e)}
    try:      # below the lane kernel's crossover: lane teams, two-sided sweep
        s8, inp = newton_solver(1024, 8, 512, 4445, device, steric=True)
        s8.set_batch(*inp[1:])
        s8.step(1)
        s8.synchronize()
        warm()
        ms4 = timed_steps(s8, 4, 0)
        it4 = s8.newton_iterations()
        ok4 = int((s8.get_status() == 0).sum())
        s8.close()
        out['small_batch_8_species'] = {'workload': 'batch=1024, 8 species size-modified, 512 points, Stern wall (lane-team kernels)',
                                        'timesteps_per_s': 1024 * 4 / (ms4 * 1e-3), 'newton_iterations_per_s': float(it4.sum()) / (ms4 * 1e-3),
                                        'lanes_ok': ok4}
    except Exception as e:
        out['small_batch_8_species'] = {'error': str(e)}
    try:      # BASELINE configs[2]: the reference's CO2R example (7 species, 5 buffer reactions, Stern wall, graded mesh), 4096 voltages
        import importlib.util
        ex = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'examples')
        sys.path.insert(0, ex)
        spec = importlib.util.spec_from_file_location('co2r_physical_sweep', os.path.join(ex, 'co2r_physical_sweep.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        from catint_amd.calculator import Calculator
        tp, phis = mod.build(4096, 384)
        calc = Calculator(transport=tp, calc='comsol', device=device)
        tp.newton = {'tol': 1e-8, 'maxit': 80}
        calc.set_surface_kinetics([{'species': 'CO2', 'rate': mod.tafel_rate(tp), 'stoichiometry': {'CO2': -1.0, 'CO': 1.0, 'OH-': 2.0}}])
        warm()
        t0 = time.perf_counter()
        calc.run()
        t_all = time.perf_counter() - t0
        out['configs2_co2r_sweep'] = {
            'workload': 'BASELINE configs[2]: examples/co2r_physical_sweep.py -- the run.py system (%d species, homogeneous buffer reactions, '
                        'size-modified K+, Stern wall, wall-graded %d-point mesh), 4096 voltages -0.5 ... -2.0 V as one batch, first-order Tafel '
                        'kinetics coupled implicitly (CatMAP is not available), stationary solves along the continuation ramp'
                        % (tp.nspecies, tp.nx),
            'lanes': 4096, 'lanes_converged': int((calc.status == 0).sum()), 'continuation_stages': int(calc.continuation_stages),
            'transport_solve_seconds': float(calc.solve_seconds), 'seconds_incl_host_result_dictionaries': t_all,
            'operating_points_per_s': 4096 / float(calc.solve_seconds)}
        its2 = float(getattr(calc, 'newton_iterations_total', 0))
        if its2 > 0:
            r2 = out['configs2_co2r_sweep']
            alg2 = lane_bytes(tp.nspecies, tp.nx)
            r2['newton_iterations'] = its2
            r2['newton_iterations_per_s'] = its2 / float(calc.solve_seconds)
            r2['mean_newton_iterations_per_stage'] = its2 / (4096.0 * max(int(calc.continuation_stages), 1))
            r2['newton_iterations_of_the_slowest_lane_summed_over_stages'] = int(getattr(calc, 'newton_iterations_slowest', 0))
            r2['lanes_handed_back_by_the_pivot_monitor'] = int(sum(len(e.get('lanes', [])) for e in getattr(calc, 'retry_log', []) or []))
            r2['roofline'] = {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'algorithmic_bytes_per_lane_iteration': alg2,
                              'achieved': alg2 * its2 / float(calc.solve_seconds) / 1e9, 'frac': alg2 * its2 / float(calc.solve_seconds) / 1e9 / HBM_PEAK_GBS,
                              'traffic': None,
                              'note': 'wall time of the eleven stage solves incl. their host calls (uploads of the stage potentials, one '
                                      'synchronisation per stage); a stage lasts as long as its slowest lane (8-9 iterations against 6.5 on '
                                      'average) and 4096 points give the lane-quad kernel 512 waves on 1024 SIMDs: latency-, not HBM-bound'}
    except Exception as e:
        out['configs2_co2r_sweep'] = {'error': '%s: %s' % (type(e).__name__, e)}
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
    if args.pmc_child:
        pmc_child(args)
        return
    if args.gpus > 1 and 'RANK' not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher -- N ranks (one per GPU, RCCL rendezvous
        # on 127.0.0.1) are started BEFORE anything here touches the GPU, rank 0 prints the one JSON line, the exit code is theirs.
        # (Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` RANK is set and every process is a rank.)
        from catint_amd.parallel import spawn_ranks
        sys.exit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # stdout carries exactly ONE line, rank 0's JSON: file descriptor 1 is pointed at stderr for everything else that writes to it
    # (gloo / RCCL banners come from C++, warnings from libraries), the line itself goes to the descriptor saved here
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    # the cpu_baseline leg's checker library: loaded (and, if stale, rebuilt by `make`) BEFORE this process initialises the GPU --
    # no child process may be started afterwards on the GPU box
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle as _co
        _co.load()
    # counters first: the child processes must be started before this process initialises the GPU
    pmc = {'error': 'skipped'}
    # (not when this process is itself being profiled -- `rocprofv3 ... -- python3 bench.py`: the profiler's preloaded library has
    # initialised the GPU already, starting child processes from here is off limits on the GPU box, and nested profilers collide)
    profiled = any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', '')
    if profiled:
        pmc = {'error': 'this process runs under rocprofv3: no nested counter collection'}
    elif rank == 0 and world == 1 and not args.no_pmc:
        pmc = collect_pmc(args)
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
        import datetime
        # (a collective that one rank never joins -- a rank that failed on its own -- raises after five minutes instead of hanging)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', device), timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=300))
    torch.cuda.set_device(device)
    comm_dev = torch.device('cuda', device) if backend == 'nccl' else torch.device('cpu')

    B, N, nx = args.batch, args.nspecies, args.nx
    solver, (prob, c0, pb, vz, fl) = compat_solver(B, N, nx, args.method, 1000 + rank, device)

    per_rank_wall = []
    from catint_amd.parallel import aligned_start, gather_numbers

    def timed_call(s, fn, extra=lambda: ()):
        """The contract's timed region on solver `s`: barrier + synchronize, a start instant agreed by all ranks (aligned_start: a
        deadline on the node's common clock instead of barrier exit), fn(), synchronize + barrier.  Wall and HIP-event time are the
        MAX over ranks; `extra()` numbers of every rank come back as columns 3.. of the table."""
        s.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0, _ = aligned_start(dist, comm_dev)
        s.timer_start()
        fn()
        ev_ms = s.timer_stop()
        s.synchronize()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        tab = gather_numbers([t0, wall, ev_ms] + list(extra()), dist, comm_dev)
        return {'wall': float(tab[:, 1].max()), 'ev_ms': float(tab[:, 2].max()), 'per_rank_wall': [float(v) for v in tab[:, 1]],
                'start_skew_us': float(tab[:, 0].max() - tab[:, 0].min()) * 1e6, 'table': tab}

    start_skew = []

    def timed(nsteps, spl):
        t = timed_call(solver, lambda: solver.step(nsteps, spl))
        per_rank_wall[:] = t['per_rank_wall']
        start_skew[:] = [t['start_skew_us']]
        return t['wall'], t['ev_ms']

    # ---- settling (untimed, not part of W): code objects loaded, clocks up (a cold GPU's clocks settle over ~0.1 s of load) ---------
    solver.set_batch(c0, pb, vz, fl)
    solver.step(args.steps, args.steps_per_launch)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.3:       # (launches of the timed length only: the rocprofv3 --stats average of the
        for _ in range(8):                            # headline kernel then IS the per-launch time of this shape)
            solver.step(args.steps, args.steps_per_launch)
        solver.synchronize()
    # ---- what a caller sees right after an upload: pnp_set_batch, then immediately the K-step launch, reported next to the sustained
    # rate (round 1: 1.45 x slower; its cause -- the dispatch preceding a launch that only just fits -- is removed inside
    # pnp_set_batch since round 2).  The reference's lagged-potential integrator does not survive long runs on
    # this workload (lanes turn NaN after ~3000 steps, in the oracle too), hence the fresh upload before the timed region as well.
    solver.set_batch(c0, pb, vz, fl)
    cold_wall, cold_ev = timed(args.steps, args.steps_per_launch)
    # ---- the timed region of the contract: upload, the W warm-up steps, then exactly K timed steps -----------------------------------
    solver.set_batch(c0, pb, vz, fl)
    solver.step(args.warmup, args.steps_per_launch)
    wall, ev_ms = timed(args.steps, args.steps_per_launch)
    status = solver.get_status()
    n_launch = (args.steps + args.steps_per_launch - 1) // args.steps_per_launch
    steps_total = world * B * args.steps
    value = steps_total / wall

    extras = rank == 0 and world == 1 and not args.no_extras and not args.shares_only
    per_step = None
    if extras and args.steps_per_launch != 1:
        solver.step(64, 1)
        fw, fe = timed(args.steps, 1)          # K launches back to back, no host synchronisation in between
        per_step = hbm_record(B, N, nx, args.steps, args.steps, fe, pmc.get('per_step_launch'))
        per_step.update({'wall_timesteps_per_s': B * args.steps / fw, 'ms_per_step_wall': fw / args.steps * 1e3, 'steps_per_launch': 1,
                         'wall_over_event': fw / (fe * 1e-3)})

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

    def warm_clocks(seconds=0.2):
        """The sub-benchmarks below build their inputs on the host for up to seconds while the GPU idles; >= 10 ms of idle time cost
        the next launches ~10 % (clocks, profiles/r02_slow_start_after_upload.txt).  Load from the headline handle right before a
        timed loop brings them back up, as the settling loop does for the headline."""
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(8):                        # (launches of the timed length, like the settling loop: the rocprofv3 --stats
                solver.step(args.steps, args.steps_per_launch)     # average of the headline kernel stays its per-launch time)
            solver.synchronize()

    # ---- N > 1: one GPU's share of BASELINE configs[3] and configs[4] on EVERY rank (the configurations BASELINE quotes for 8 GPUs),
    # compat step per launch and coupled-Newton lane kernels, each with the contract's timed region (common start, MAX over ranks) --
    # the scaling curve on those shapes; at N = 1 the same shapes are the `beyond_cache` / `physical_mode` records below
    shares = None
    if dist is not None and not args.no_extras:          # (world > 1, or CATINT_FORCE_DIST: the RCCL rehearsal of one rank)
        shares = {}

        def lane_bytes_(N_, nx_, fused):      # (as lane_bytes of physical_mode: the lane kernel fuses the update into the back-substitution)
            return 8.0 * nx_ * (2 * (N_ + 1) * (N_ + 2) + (4 * N_ + 3 if fused else 6 * N_ + 5))

        def compat_share(shape, seed, nsteps):
            SB, SN, SX = shape
            s_, inp = compat_solver(SB, SN, SX, args.method, seed + rank, device)
            s_.set_batch(*inp[1:])
            del inp
            s_.step(nsteps, 1)
            s_.synchronize()
            warm_clocks()
            s_.step(nsteps, 1)
            t = timed_call(s_, lambda: s_.step(nsteps, 1), lambda: [float((s_.get_status() == 0).sum())])
            s_.close()
            alg = 16.0 * (SN + 1) * SX * SB
            return {'workload': 'per GPU: batch=%d, %d species, %d grid points, compat step, one launch per timestep' % (SB, SN, SX),
                    'value': world * SB * nsteps / t['wall'], 'unit': 'timesteps/s', 'steps': nsteps,
                    'per_rank_timesteps_per_s': [SB * nsteps / w for w in t['per_rank_wall']], 'start_skew_us': t['start_skew_us'],
                    'hbm_frac_algorithmic_slowest_rank': alg * nsteps / (t['ev_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    'lanes_ok': int(t['table'][:, 3].sum()), 'lanes_total': world * SB}

        def newton_share(shape, seed, nsteps):
            SB, SN, SX = shape
            s_, inp = newton_solver(SB, SN, SX, seed + rank, device, steric=True)
            s_.set_batch(*inp[1:])
            del inp
            s_.step(1)
            s_.synchronize()
            s_.tune_placement(2, 6)          # (the workspace where it runs fastest: see lane_record)
            warm_clocks()
            t = timed_call(s_, lambda: s_.step(nsteps), lambda: [float(s_.newton_iterations().sum()), float((s_.get_status() == 0).sum())])
            fused = s_.default_family() == 'lane+fused'
            s_.close()
            its = t['table'][:, 3]
            return {'workload': 'per GPU: batch=%d, %d species size-modified, %d points, Stern wall, backward Euler, coupled Newton'
                                % (SB, SN, SX),
                    'value': world * SB * nsteps / t['wall'], 'unit': 'timesteps/s', 'steps': nsteps,
                    'newton_iterations_per_s': float(its.sum()) / t['wall'],
                    'per_rank_timesteps_per_s': [SB * nsteps / w for w in t['per_rank_wall']], 'start_skew_us': t['start_skew_us'],
                    'hbm_frac_algorithmic_slowest_rank': lane_bytes_(SN, SX, fused) * float(its.max()) / (t['ev_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    'lanes_ok': int(t['table'][:, 4].sum()), 'lanes_total': world * SB}

        div = max(1, int(os.environ.get('CATINT_BENCH_SHARE_DIV', '1')))      # tests: the same records on 1/div of the batch
        for key, shape, seeds, nst in (('configs3_share', BC_SHAPE, (55, 4447), (8, 10)), ('configs4_share', C4_SHAPE, (56, 4448), (4, 4))):
            shape = (shape[0] // div, shape[1], shape[2])
            rec = {}
            for name, fn, seed, n_ in (('compat_per_step', compat_share, seeds[0], nst[0]), ('newton', newton_share, seeds[1], nst[1])):
                try:
                    rec[name] = fn(shape, seed, n_)
                except Exception as e:      # (every rank takes the same path: a failure here is a failure on all of them)
                    rec[name] = {'error': '%s: %s' % (type(e).__name__, e)}
            shares[key] = rec

    large = None
    if extras and args.large_batch > 0:
        LB = args.large_batch
        s2, inp = compat_solver(LB, N, nx, args.method, 77, device)
        s2.set_batch(*inp[1:])
        s2.step(64, 1)
        ls = max(10, min(args.steps, 50))
        s2.synchronize()
        warm_clocks()
        s2.step(16, 1)
        lms = timed_steps(s2, ls, 1)
        lok = int((s2.get_status() == 0).sum())
        large = hbm_record(LB, N, nx, ls, ls, lms, pmc.get('large_batch') if isinstance(pmc, dict) else None)
        large.update({'batch': LB, 'steps': ls, 'lanes_ok': lok, 'steps_per_launch': 1, 'state_MB': 8.0 * (N + 2) * nx * LB / 1e6,
                      'row_chunks': s2.step_row_chunks(ls)})
        s2.set_batch(*inp[1:])
        for _ in range(4):
            s2.step(8, 8)
        s2.step(args.steps_per_launch, args.steps_per_launch)
        s2.synchronize()
        warm_clocks()
        fms = timed_steps(s2, args.steps, args.steps_per_launch)
        s2.close()
        large['fused'] = {'timesteps_per_s': LB * args.steps / (fms * 1e-3), 'steps_per_launch': args.steps_per_launch,
                          'frac': 16.0 * (N + 1) * nx * LB * args.steps / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    beyond = None
    if extras:
        BB, BN, BX = BC_SHAPE
        try:
            s3, inp = compat_solver(BB, BN, BX, args.method, 55, device)
            s3.set_batch(*inp[1:])
            s3.step(8, 1)
            s3.synchronize()
            warm_clocks()
            s3.step(8, 1)
            ms1 = timed_steps(s3, 8, 1, reps=3)
            ok1 = int((s3.get_status() == 0).sum())
            rec1 = hbm_record(BB, BN, BX, 8, 8, ms1, pmc.get('beyond_cache_per_step'))
            rec1['row_chunks'] = s3.step_row_chunks(8)      # (launch_us: per timestep = one launch per chunk, on streams of their own)
            s3.set_batch(*inp[1:])
            s3.step(32, 32)
            s3.synchronize()
            warm_clocks()
            ms32 = timed_steps(s3, 32, 32, reps=3)
            rec32 = hbm_record(BB, BN, BX, 32, 1, ms32, pmc.get('beyond_cache_fused'))
            s3.close()
            beyond = {'workload': "one GPU's share of BASELINE configs[3]: batch=%d, %d species, %d grid points, Crank-Nicolson compat "
                                  'integrator' % (BB, BN, BX), 'state_MB': 8.0 * (BN + 2) * BX * BB / 1e6, 'lanes_ok': ok1,
                      'per_step_launch': rec1, 'fused_32_steps_per_launch': rec32}
            # one GPU's share of configs[4]: 4096 grid points = four waves per tridiagonal system (step_kernel_mw)
            CB, CN_, CX = C4_SHAPE
            del inp
            s4, inp4 = compat_solver(CB, CN_, CX, args.method, 56, device)
            s4.set_batch(*inp4[1:])
            del inp4
            s4.step(4, 1)
            s4.synchronize()
            warm_clocks()
            s4.step(4, 1)
            ms4 = timed_steps(s4, 4, 1, reps=3)
            ok4 = int((s4.get_status() == 0).sum())
            rec4 = hbm_record(CB, CN_, CX, 4, 4, ms4, None)
            rec4.update({'row_chunks': s4.step_row_chunks(4), 'lanes_ok': ok4, 'state_MB': 8.0 * (CN_ + 2) * CX * CB / 1e6,
                         'workload': "one GPU's share of BASELINE configs[4]: batch=%d, %d species, %d grid points, one launch per "
                                     'timestep (step_kernel_mw: four waves per tridiagonal system)' % (CB, CN_, CX)})
            s4.close()
            beyond['beyond_cache_config4'] = rec4
        except Exception as e:
            beyond = {'error': '%s: %s' % (type(e).__name__, e)}

    if rank == 0:
        # 2*8*(N+1)*nx bytes per lane-timestep; a launch advances B lanes by steps_per_launch steps
        head = hbm_record(B, N, nx, args.steps, n_launch, ev_ms, pmc.get('headline') if isinstance(pmc, dict) else None)
        roof = {'bound': 'hbm', 'achieved': head['achieved_GBs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': head['frac'],
                'traffic': head['traffic'],
                'kernel': head.get('kernel') or 'pnp::step_kernel* (P=%d points/lane)' % next((P for P in (1, 2, 4, 8, 16) if nx - 2 <= 64 * P), 16),
                'launch_us': head['launch_us'], 'algorithmic_bytes_per_launch': head['algorithmic_bytes_per_launch'],
                'timesteps_per_launch': head['timesteps_per_launch'],
                'achieved_is': 'algorithmic bytes (16(N+1)nx per lane-timestep, state written every step) / HIP-event launch time'}
        for k in ('traffic_over_algorithmic', 'hbm_GBs_measured', 'hbm_frac_measured', 'kernel_resources', 'sq'):
            if k in head:
                roof[k] = head[k]
        if head['traffic'] is not None:
            roof['traffic_source'] = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ (separate passes) over this command\'s launch shapes, ' \
                                     'collected by this run before its timed region; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB'
            if head['traffic_over_algorithmic'] < 0.7:
                roof['note'] = ('fused launch: the previous state is re-used from LDS/registers and L2 write-combines successive steps, so ' +
                                'HBM sees %.2fx the algorithmic bytes -- this launch is bound by %s, not by HBM; the HBM-bound figures are ' +
                                'per_step_launch and beyond_cache') % (head['traffic_over_algorithmic'],
                                                                       head.get('sq', {}).get('binds', 'the latency chain of a step'))
        else:
            roof['traffic_reason'] = pmc.get('error', 'no counters for this launch') if isinstance(pmc, dict) else 'no counters'
        out = {
            'metric': 'batched 1D PNP Newton-timesteps/sec', 'value': value, 'unit': 'timesteps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': wall / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: batch=%d operating points/GPU, %d species, %d grid points, fp64, %s '
                                   'compat integrator, Dirichlet-Dirichlet Poisson' % (B, N, nx, args.method),
                       'batch_per_gpu': B, 'nspecies': N, 'nx': nx, 'steps_per_launch': args.steps_per_launch,
                       'parallelism': 'batch-sharded x%d' % world},
            'roofline': roof,
            'timed_region': {'wall_us': wall * 1e6, 'event_us': ev_ms * 1e3, 'host_launch_and_sync_us': wall * 1e6 - ev_ms * 1e3,
                             'event_timesteps_per_s': B * args.steps / (ev_ms * 1e-3), 'wall_over_event': wall / (ev_ms * 1e-3)},
            'first_launch_after_upload': {'timesteps_per_s_wall': world * B * args.steps / cold_wall,
                                          'timesteps_per_s_event': B * args.steps / (cold_ev * 1e-3), 'event_us': cold_ev * 1e3,
                                          'sustained_over_cold': (cold_ev / ev_ms) if ev_ms > 0 else None},
            'lanes_ok': int((status == 0).sum()), 'lanes_total': int(B),
        }
        if per_step:
            out['per_step_launch'] = per_step
        if large:
            out['large_batch'] = large
        if beyond:
            out['beyond_cache'] = beyond
        if extras and args.physical_steps > 0:
            out['physical_mode'] = physical_mode(args, device, not args.no_cpu_baseline, pmc if isinstance(pmc, dict) else {}, warm_clocks)
        if gather_ms is not None:
            out['gather_ms'] = gather_ms
            out['gather'] = {'collective': 'all_gather_into_tensor of [B_local, N+2] fp64 observables', 'backend': backend,
                             'bytes_per_rank': int(B * (N + 2) * 8)}
        if world > 1:
            out['per_rank_timesteps_per_s'] = [B * args.steps / w for w in per_rank_wall]
            out['n_ranks_launched'] = world
        if dist is not None:
            out['start_skew_us'] = start_skew[0] if start_skew else None
        if shares:
            out.update(shares)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(prob, c0, pb, vz, fl, args.method, args.cpu_seconds)
            if args.method == 'Crank-Nicolson':
                cb['reference_faithful_dense_1core'] = cpu_reference_faithful(prob, c0, pb, vz, fl, args.method)
            out['cpu_baseline'] = cb
        json_out.write(json.dumps(out) + '\n')
        json_out.flush()
    solver.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
