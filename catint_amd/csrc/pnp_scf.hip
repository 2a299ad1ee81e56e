// Kinetics <-> transport self-consistency loop, per lane, on the device.
//
// The reference alternates CatMAP and the transport solve on the host, one operating point at a time
// (Calculator.run_scf_cycle, catint/calculator.py:294-406): mix the surface concentrations with the previous iterate
// (:328-344, negative values fall back to the previous iterate), update the surface pH (:346-359), ask the kinetic model for
// fluxes (:373), solve the transport problem (:385), compare current densities with the previous iteration
// (evaluate_accuracy :260-283, signed quotient) and decay the mixing factor every 40 iterations (:319-323).  Here the
// kinetic model is analytic -- the table of pnp_set_wall_kinetics, flux_k = sum_r nu_rk K_r[lane] g_r(c_s(r)(x=0)) E_r, a Tafel
// law in the electrode potential once K carries it, Butler-Volmer in the Stern-layer drop and Langmuir saturation through
// pnp_set_wall_rate_law (E_r = exp(alpha_r (phiM - phi(0))), g_r = c/(1 + K_sat c)) -- so the whole iteration stays on the device:
// scf_pre_kernel (bookkeeping + fluxes), the Newton solve of the active lanes (pnp_newton.hip, lane mask), scf_keep_kernel
// (a converged lane's state becomes its snapshot, a failed lane gets its snapshot back), scf_post_kernel (surface state,
// accuracy, flags).  The host only reads one counter every few iterations to see whether lanes are left.
//
// The arithmetic mirrors catint_amd/calculator.py:run_scf_cycle (the batched host loop with a Python callback) operation by
// operation, with explicitly rounded products and sums (no fused multiply-add), so both loops walk the same iterates.
#include <hip/hip_runtime.h>

#include <cmath>

#include "pnp_internal.h"

namespace pnp {

__global__ __launch_bounds__(256) void scf_pre_kernel(const ScfArgs A) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= A.B) return;
  if (!A.active[b]) return;
  atomicMax(&A.counters[64], A.istep);
  const int N = A.N;
  double mix = A.mix[b];
  if (A.istep - A.step_to_check[b] > 40) {       // calculator.py:319-323
    mix = __dmul_rn(mix, 0.9);
    A.mix[b] = mix;
    A.step_to_check[b] = A.istep;
  }
  double s[PNP_NEWTON_MAX_SPECIES];
  for (int k = 0; k < N; ++k) {
    const double cur = A.sc[b * N + k], old = A.sc_old[b * N + k];
    double m;
    if (A.istep > 2) m = cur < 0.0 ? old : __dadd_rn(__dmul_rn(mix, cur), __dmul_rn(__dsub_rn(1.0, mix), old));      // :328-338
    else m = cur < 0.0 ? 1e-20 : cur;                                                                                  // :341-344
    s[k] = m;
    A.sc[b * N + k] = m;
    A.sc_old[b * N + k] = m;                                                                                           // :362-364
  }
  if (A.iH >= 0) {                                                                                                     // :346-359
    if (s[A.iH] > 0.0) A.surface_pH[b] = -log10(s[A.iH] / 1000.0);
  } else if (A.iOH >= 0) {
    if (s[A.iOH] > 0.0) A.surface_pH[b] = 14.0 + log10(s[A.iOH] / 1000.0);
  }
  double f[PNP_NEWTON_MAX_SPECIES];
  for (int k = 0; k < N; ++k) f[k] = 0.0;
  for (int r = 0; r < A.n_wk; ++r) {
    const double K = A.wk_k[b * PNP_MAX_WALL_REACTIONS + r];
    const int sp = A.wk_species[r];
    const double cs = sp >= 0 ? fmax(s[sp], 0.0) : 1.0;
    // rate law of pnp_set_wall_rate_law, evaluated explicitly: Langmuir saturation in the mixed surface concentration,
    // Butler-Volmer factor in the Stern-layer drop of the PREVIOUS transport solve (the state the reference hands CatMAP,
    // catmap_wrapper.py: surface potential of the last COMSOL run).  alpha = K_sat = 0: g = cs exactly.
    const double al = A.wk_alpha[r], ks = A.wk_sat[r];
    double g = cs;
    if (ks != 0.0) g = __dmul_rn(g, __ddiv_rn(1.0, __dadd_rn(1.0, __dmul_rn(ks, cs))));
    if (al != 0.0) g = __dmul_rn(g, exp(__dmul_rn(al, __dsub_rn(A.pb[b * 4], A.vsurf[b]))));
    for (int k = 0; k < N; ++k) {
      const double nu = A.wk_nu[r][k];
      if (nu != 0.0) f[k] = __dadd_rn(f[k], __dmul_rn(__dmul_rn(nu, K), g));
    }
  }
  for (int k = 0; k < N; ++k) A.flux[b * N + k] = f[k];
}

// One workgroup per lane, before scf_post_kernel updates the activity flags.  The reference restarts COMSOL from scratch after
// a failed solve; here the lane goes back to the state of its last converged solve (the caller hands over converged states).
__global__ __launch_bounds__(256) void scf_keep_kernel(const ScfArgs A) {
  const int64_t b = blockIdx.x;
  if (!A.active[b]) return;
  const int st = A.status[b];
  const size_t nc = (size_t)A.N * A.ldx;
  double* c = A.c + b * nc;
  double* sc = A.snap_c + b * nc;
  double* p = A.phi + b * (size_t)A.ldx;
  double* sp = A.snap_phi + b * (size_t)A.ldx;
  if (st == PNP_STATUS_OK) {
    for (size_t e = threadIdx.x; e < nc; e += blockDim.x) sc[e] = c[e];
    for (int e = threadIdx.x; e < A.ldx; e += blockDim.x) sp[e] = p[e];
  } else if (st == PNP_STATUS_MAXIT) {
    for (size_t e = threadIdx.x; e < nc; e += blockDim.x) c[e] = sc[e];
    for (int e = threadIdx.x; e < A.ldx; e += blockDim.x) p[e] = sp[e];
  }
}

__global__ __launch_bounds__(256) void scf_post_kernel(const ScfArgs A) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= A.B) return;
  if (!A.active[b]) return;
  const int N = A.N;
  const int st = A.status[b];
  // A lane whose Newton iteration did not converge is reported the way COMSOL reports an unreachable flux: a negative
  // surface concentration (the next iteration then falls back to the previous iterate, calculator.py:328-344).
  const bool stuck = st == PNP_STATUS_MAXIT;
  bool finite = true, negative = false;
  double err = -INFINITY;
  for (int k = 0; k < N; ++k) {
    const double cs = stuck ? -1.0 : A.c[((size_t)b * N + k) * A.ldx];
    A.sc[b * N + k] = cs;
    finite = finite && (fabs(cs) < INFINITY);
    negative = negative || cs < 0.0;
    // mA/cm^2: flux*nel*F/nprod/10 (calculator.py:389-400), evaluated left to right
    const double cd = __ddiv_rn(__ddiv_rn(__dmul_rn(__dmul_rn(A.flux[b * N + k], A.nel[k]), A.faraday), A.nprod[k]), 10.0);
    const double old = A.cd_old[b * N + k];
    if (cd != 0.0) err = fmax(err, __ddiv_rn(fabs(__dsub_rn(cd, old)), cd));      // evaluate_accuracy :260-283: SIGNED quotient
    A.cd_old[b * N + k] = cd;
  }
  const double p0 = A.phi[(size_t)b * A.ldx], p1 = A.phi[(size_t)b * A.ldx + 1];
  A.vsurf[b] = p0;
  A.esurf[b] = -(p1 - p0) / A.h0;
  double acc = A.acc[b];
  if (A.istep > 1) {
    acc = err;
    A.acc[b] = acc;
  }
  const bool bad = !finite || st == PNP_STATUS_NAN;
  if (bad) A.failed[b] = 1;
  const bool go_on = !bad && (acc > A.tau || negative);                               // :316, :404-406
  A.active[b] = go_on ? 1 : 0;
  if (go_on) atomicAdd(&A.counters[A.slot], 1);
}

hipError_t launch_scf_pre(const ScfArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(scf_pre_kernel, dim3((unsigned)((a.B + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_scf_keep(const ScfArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(scf_keep_kernel, dim3((unsigned)a.B), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_scf_post(const ScfArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(scf_post_kernel, dim3((unsigned)((a.B + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace pnp
