"""In-tree build of the HIP library (gfx950 only; cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_DIR = os.path.join(_HERE, 'lib')
LIB = os.path.join(LIB_DIR, 'libcatint_pnp.so')
SOURCES = ['pnp_kernels.hip', 'pnp_stream.hip', 'pnp_newton.hip', 'pnp_lane.hip', 'pnp_lane2.hip', 'pnp_lane4.hip', 'pnp_scf.hip', 'pnp_ode.hip', 'pnp_rkc.hip', 'pnp_capi.hip']
HEADERS = [os.path.join(CSRC, 'pnp_internal.h'), os.path.join(CSRC, 'pnp_lane_common.h'), os.path.join(CSRC, 'pnp_wave.h'), os.path.join(CSRC, 'pnp_step_table.h'), os.path.join(CSRC, 'pnp_math.h'), os.path.join(CSRC, 'pnp_dop853_coeffs.h'), os.path.join(_HERE, '..', 'include', 'catint_pnp.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wall', '-Wno-unused-function']


PARTIAL = os.path.join(LIB_DIR, '.partial')      # left by tools/devbuild.sh: the library holds only one block size


def needs_build():
    if not os.path.exists(LIB) or os.path.exists(PARTIAL):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, obj, verbose):
    cmd = [HIPCC] + [f for f in FLAGS if f != '-shared'] + ['-c', src, '-o', obj]
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed on %s:\n%s%s' % (src, r.stdout, r.stderr))


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950: one object per source (kept under lib/obj, recompiled when the source or a header is newer),
    linked into catint_amd/lib/libcatint_pnp.so"""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(LIB_DIR, 'obj')
    os.makedirs(obj_dir, exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in HEADERS)
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            jobs.append((src, obj))
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for f in [ex.submit(_compile, src, obj, verbose) for src, obj in jobs]:
            f.result()
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc link failed:\n' + r.stdout + r.stderr)
    if os.path.exists(PARTIAL):
        os.remove(PARTIAL)
    return LIB
