"""In-tree build of the HIP library (gfx950 only; cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_DIR = os.path.join(_HERE, 'lib')
LIB = os.path.join(LIB_DIR, 'libcatint_pnp.so')
SOURCES = ['pnp_kernels.hip', 'pnp_stream.hip', 'pnp_newton.hip', 'pnp_scf.hip', 'pnp_ode.hip', 'pnp_capi.hip']
HEADERS = [os.path.join(CSRC, 'pnp_internal.h'), os.path.join(CSRC, 'pnp_wave.h'), os.path.join(CSRC, 'pnp_step_table.h'), os.path.join(CSRC, 'pnp_math.h'), os.path.join(CSRC, 'pnp_dop853_coeffs.h'), os.path.join(_HERE, '..', 'include', 'catint_pnp.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wall', '-Wno-unused-function']


PARTIAL = os.path.join(LIB_DIR, '.partial')      # left by tools/devbuild.sh: the library holds only one block size


def needs_build():
    if not os.path.exists(LIB) or os.path.exists(PARTIAL):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared  ->  catint_amd/lib/libcatint_pnp.so"""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', LIB]
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)
    if os.path.exists(PARTIAL):
        os.remove(PARTIAL)
    return LIB
