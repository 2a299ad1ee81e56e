"""The one JSON line of bench.py (the driver's contract): keys, types and internal consistency -- on the committed line of the driver's
command (profiles/, CPU) and on a fresh short run (GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {'metric': str, 'value': float, 'unit': str, 'n_gpus': int, 'steps': int, 'warmup': int, 'ms_per_step': float,
            'higher_is_better': bool, 'scaling': str, 'dtype': str, 'data': str, 'config': dict, 'roofline': dict, 'cpu_baseline': dict}


def check_line(d, with_cpu=True):
    for k, t in REQUIRED.items():
        if k == 'cpu_baseline' and not with_cpu:
            continue
        assert k in d, k
        assert isinstance(d[k], t), (k, type(d[k]))
    assert 'vs_baseline' in d and d['vs_baseline'] is None            # BASELINE.md holds no published number for this metric
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['dtype'] == 'f64' and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    B = d['config']['batch_per_gpu']
    # value = the units all ranks processed / the timed K steps
    assert d['value'] == pytest.approx(d['n_gpus'] * B / (d['ms_per_step'] * 1e-3), rel=1e-9)
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and r['peak'] == 8000.0
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=1e-12) and 0.0 < r['frac'] < 1.0
    # achieved = algorithmic bytes per launch / the launch's HIP-event time
    assert r['achieved'] == pytest.approx(r['algorithmic_bytes_per_launch'] / (r['launch_us'] * 1e-6) / 1e9, rel=1e-6)
    assert r['algorithmic_bytes_per_launch'] == 16.0 * (d['config']['nspecies'] + 1) * d['config']['nx'] * B * r['timesteps_per_launch']
    if with_cpu:
        c = d['cpu_baseline']
        for k in ('value', 'unit', 'cores', 'kind', 'sample'):
            assert k in c, k
        assert c['kind'] in ('reference', 'port') and c['unit'] == d['unit'] and c['cores'] >= 1 and c['value'] > 0


def test_committed_bench_line_of_the_drivers_command():
    d = json.loads(open(os.path.join(ROOT, 'profiles', 'r04_bench_line_steps20.json')).read().strip().splitlines()[-1])
    check_line(d)
    assert d['steps'] == 20 and d['warmup'] == 5 and d['n_gpus'] == 1 and d['lanes_ok'] == d['lanes_total']
    assert r'configs[1]' in d['config']['workload']
    assert d['roofline']['traffic'] is not None                       # counters were collected by the run itself
    # the records beside the headline that DESIGN.md quotes
    pm, bc = d['physical_mode'], d['beyond_cache']
    for k in ('large_batch_8_species', 'large_batch_8_species_32k', 'config3_share', 'config4_share', 'configs2_co2r_sweep'):
        assert k in pm and 'error' not in pm[k], k
    c2 = pm['configs2_co2r_sweep']
    assert c2['lanes_converged'] == 4096 and c2['lanes_handed_back_by_the_pivot_monitor'] == 0
    assert c2['newton_iterations_per_s'] > 0 and 0.0 < c2['roofline']['frac'] < 1.0
    assert pm['config4_share']['roofline']['traffic'] is not None      # counters for the configs[4] share
    for k in ('per_step_launch', 'fused_32_steps_per_launch', 'beyond_cache_config4'):
        assert 0.2 < bc[k]['frac'] < 1.0


def test_committed_two_rank_line_is_one_json_document():
    """`CATINT_DIST_BACKEND=gloo python bench.py --gpus 2 ...` on a one-GPU box: the committed stdout is ONE JSON document (gloo / RCCL
    chatter goes to stderr) with the per-rank shares of configs[3] / configs[4] and the start skew of the aligned timed regions."""
    d = json.loads(open(os.path.join(ROOT, 'profiles', 'r04_bench_line_gpus2_gloo_rehearsal.json')).read())
    check_line(d, with_cpu=False)
    assert d['n_gpus'] == 2 and len(d['per_rank_timesteps_per_s']) == 2 and d['start_skew_us'] < 1000.0
    for key in ('configs3_share', 'configs4_share'):
        for leg in ('compat_per_step', 'newton'):
            r = d[key][leg]
            assert r['lanes_ok'] == r['lanes_total'] and len(r['per_rank_timesteps_per_s']) == 2
            assert r['value'] == pytest.approx(sum(r['per_rank_timesteps_per_s']), rel=0.2)


@pytest.mark.gpu
def test_fresh_short_run_prints_one_line_that_keeps_the_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '1', '--no-extras', '--no-pmc',
                          '--no-cpu-baseline'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    check_line(d, with_cpu=False)
    assert d['steps'] == 4 and d['warmup'] == 1 and d['n_gpus'] == 1
