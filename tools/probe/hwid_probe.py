#!/usr/bin/env python3
"""Diagnosis (library built with -DPNP_HWID_PROBE): SIMD placement of the three waves of every workgroup of the fused headline
launch in its slow mode (first launch after pnp_set_batch) and in its fast mode (after a short kernel)."""
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd import _capi                      # noqa: E402
from catint_amd.synthetic import make_batch       # noqa: E402

B = 1024
prob, c0, pb, vz, fl = make_batch(B, 3, 512, seed=0, dt_factor=1e-5)
s = _capi.PnpSolver(3, 512, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Crank-Nicolson', batch_capacity=B)


def one(tag, cure):
    s.set_batch(c0, pb, vz, fl)
    if cure:
        s.get_surface(); s.get_surface()
    s.timer_start(); s.step(256, 256); ms = s.timer_stop()
    st = s.get_status().astype(np.uint32)
    simd = np.zeros((B, 3), int)
    for w in range(3):
        bits = (st >> (8 + 4 * w)) & 0xF
        simd[:, w] = np.log2(np.maximum(bits, 1)).astype(int)
    cu = (st >> 20) & 0xF
    se = (st >> 24) & 0x7
    pat = collections.Counter(tuple(r) for r in simd)
    per_simd = np.bincount(simd.ravel(), minlength=4)
    distinct = sum(1 for r in simd if len(set(r)) == 3)
    # blocks per (se, cu) -- XCDs are not distinguished by HW_ID, so this folds the 8 XCDs together
    cu_load = collections.Counter(zip(se.tolist(), cu.tolist()))
    loads = np.array(sorted(cu_load.values()))
    print('%-22s %.2f us/step | waves per SIMD %s | workgroups with 3 distinct SIMDs %d/%d | patterns %s | (se,cu) slots %d, blocks per slot min %d max %d'
          % (tag, ms / 256 * 1e3, per_simd.tolist(), distinct, B, pat.most_common(4), len(cu_load), loads.min(), loads.max()), flush=True)


for rep in range(4):
    one('after set_batch', False)
    one('after set_batch + cure', True)
