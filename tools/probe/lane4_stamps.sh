#!/bin/bash
# diagnosis build of the lane-quad kernel with cycle stamps per pass (forward / backward / update); GPU box.
# usage: bash tools/probe/lane4_stamps.sh NSPECIES NX BATCH
N=${1:-8}; NX=${2:-512}; B=${3:-8192}
R=$PWD; D=/tmp/lane4stamps; mkdir -p $D/catint_amd
cp -r $R/catint_amd/* $D/catint_amd/ && cp -r $R/tools $R/include $D/
cd $D && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DPNP_LANE_STAMPS $EXTRA catint_amd/csrc/pnp_lane4.hip -o catint_amd/lib/obj/pnp_lane4.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC catint_amd/lib/obj/*.o -o catint_amd/lib/libcatint_pnp.so || exit 1
CATINT_NEWTON_KERNEL=lane4 CATINT_LANE_ORDER=0 python3 - $N $NX $B <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from catint_amd import _capi
from catint_amd.synthetic import make_batch
N, nx, B = map(int, sys.argv[1:4])
prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=0, phi_max=0.2, dt_factor=0.1)
s = _capi.PnpSolver(prob.N, prob.nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton', batch_capacity=B)
s.set_newton(wall_bc='stern', stern_capacitance=0.2, tol=1e-8, mpb_radius=[4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 3e-10, 3e-10, 4.5e-10, 3.5e-10][:N])
import os
if os.environ.get('REACT'):      # five buffer-like reactions (the shape of the reference's CO2R table: data.py:7-122)
    s.set_reactions([([1, 2], [4], 5.93e3, 1.34e2), ([4, 2], [5], 1e5, 2.1e4), ([], [6, 2], 2.4e-2, 2.4e3), ([1], [4, 6], 3.7e-2, 8.3e1), ([4], [5, 6], 5.9e1, 1.3e6)][:int(os.environ['REACT'])])
s.set_batch(c0, np.nan_to_num(pb), vz, fl)
s.step(2); s.synchronize()
s.timer_start(); s.step(10); ms = s.timer_stop()
it = s.newton_iterations().reshape(-1, 8)
f, b, u, n = it[:, 0].astype(float), it[:, 1].astype(float), it[:, 2].astype(float), it[:, 3].astype(float)
tot = f + b + u
print('N=%d nx=%d B=%d: %.3f ms per step; per wave and Newton iteration: forward %.0f cycles (%.0f per row), backward %.0f (%.0f), update %.0f (%.0f per row of a lane); shares %.2f / %.2f / %.2f; iterations per wave %.1f'
      % (N, nx, B, ms / 10, f.mean(), f.mean() / (nx / 2), b.mean(), b.mean() / (nx / 2), u.mean(), u.mean() / (nx / 8), (f / tot).mean(), (b / tot).mean(), (u / tot).mean(), n.mean()))
print('forward row, mean cycles: assembly (edges, right-hand side) %.0f, columns of D\' and Ah %.0f, Gauss-Jordan %.0f, arrival + record stores + hand-over %.0f'
      % (it[:, 4].mean(), it[:, 5].mean(), it[:, 6].mean(), it[:, 7].mean()))
PY
