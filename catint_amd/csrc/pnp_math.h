// Elementary functions of the Newton kernels with every constant in SCALAR registers at the point of use.
// The library versions keep their polynomial coefficients in vector registers hoisted out of the Newton loop; in the
// 128-register kernels those were spilled and re-read from scratch (which misses L2) -- one dependent memory round trip per
// Horner step.  Accuracy is checked on the device by tools/probe/mathfn_probe.hip (expm1_sc <= 2 ulp, log1p_sc <= 3 ulp against the host libm over 1e6 arguments each).
#pragma once
#include <hip/hip_runtime.h>

namespace pnp {

__device__ __forceinline__ double nrcp(double x) {   // v_rcp_f64 + two Newton steps (1.1e-16 relative)
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}

// Third way out of the Newton loop (besides update < tol and the quadratic error estimate): the iteration has reached its rounding
// floor.  Near the solution a full step contracts quadratically; two consecutive full steps within 100 tol of which the second is NOT
// SMALLER than the first are the rounding noise of an ill-conditioned Jacobian (stiff reaction terms: cond(J) eps ~ 3e-9 in the case
// of tests/golden/fuzz/newton_case117.json), not Newton -- without this rule such a lane iterates until the noise happens to dip below
// tol (20 iterations on the device, 31 in the oracle, 37 in another kernel for the same state).  Round 3 accepted a pair whose second
// update was more than HALF the first: that also lets a LINEARLY converging iteration out (singular Jacobian at the root: every update a
// fixed fraction of the one before, error upd / (1 - ratio), far above tol -- ADVICE r03).  A linearly converging sequence shrinks at
// every step and never satisfies "not smaller"; it runs on to tol or to the iteration limit and is then reported as not converged.
// (A rule that asked for a non-monotone TRIPLE was tried first: as safe, but the exits of different linear solvers then lie up to
// three iterations apart instead of one, and a lane at the iteration limit flips between converged and not.)
// Same rule in oracle/pnp_physical.py (at_rounding_floor); contract stated in include/catint_pnp.h.
__device__ __forceinline__ bool newton_at_rounding_floor(double upd, double upd_prev, double tol) {
#ifdef PNP_NO_ROUNDING_FLOOR_EXIT      // (diagnosis builds)
  return false;
#else
  const double w = 100.0 * tol;
  return upd_prev < w && upd < w && upd >= upd_prev;
#endif
}

// exp(u) - 1 for |u| >= 0.05 (smaller arguments take the Taylor branch of the caller), all constants in scalar registers.
// The library expm1 keeps its ~10 polynomial coefficients in vector registers hoisted out of the Newton loop; in the
// 128-register kernels they were spilled and re-read from scratch -- one dependent memory round trip per Horner step.
// u = k ln2 + r, |r| <= ln2/2;  e^r - 1 by a degree-13 Taylor polynomial (remainder < 4e-18);  e^u - 1 = 2^k (e^r - 1) + (2^k - 1).
__device__ __forceinline__ double expm1_sc(double u) {
  double l2e = 1.4426950408889634, ln2h = 6.93147180369123816490e-01, ln2l = 1.90821492927058770002e-10;
  double c2 = 1.0 / 2, c3 = 1.0 / 6, c4 = 1.0 / 24, c5 = 1.0 / 120, c6 = 1.0 / 720, c7 = 1.0 / 5040, c8 = 1.0 / 40320,
         c9 = 1.0 / 362880, c10 = 1.0 / 3628800, c11 = 1.0 / 39916800, c12 = 1.0 / 479001600, c13 = 1.0 / 6227020800.0;
  asm volatile("" : "+s"(l2e), "+s"(ln2h), "+s"(ln2l), "+s"(c2), "+s"(c3), "+s"(c4), "+s"(c5), "+s"(c6), "+s"(c7), "+s"(c8),
               "+s"(c9), "+s"(c10), "+s"(c11), "+s"(c12), "+s"(c13));
  const double uc = fmin(fmax(u, -60.0), 709.0);           // e^-60 - 1 == -1 in double; beyond 709 the result is inf anyway
  const double kf = __builtin_rint(uc * l2e);
  const double r = __builtin_fma(-kf, ln2l, __builtin_fma(-kf, ln2h, uc));
  double p = c13;
  p = __builtin_fma(p, r, c12);
  p = __builtin_fma(p, r, c11);
  p = __builtin_fma(p, r, c10);
  p = __builtin_fma(p, r, c9);
  p = __builtin_fma(p, r, c8);
  p = __builtin_fma(p, r, c7);
  p = __builtin_fma(p, r, c6);
  p = __builtin_fma(p, r, c5);
  p = __builtin_fma(p, r, c4);
  p = __builtin_fma(p, r, c3);
  p = __builtin_fma(p, r, c2);
  p = __builtin_fma(p * r, r, r);                             // e^r - 1
  const int k = (int)kf;
  const double t = __builtin_ldexp(1.0, k);
  const double res = __builtin_fma(t, p, t - 1.0);
  return u > 709.0 ? INFINITY : res;
}

// log(1 + x) for x in (-1, 0] (x = -phi0, the occupied volume fraction).  y = 1 + x with its rounding error c = x - (y - 1)
// carried along;  y = 2^k m, m in [sqrt(1/2), sqrt(2));  log m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716: odd series to
// s^23 (remainder < 1e-18);  result = k ln2 + log m + c/y.
__device__ __forceinline__ double log1p_sc(double x) {
  double ln2h = 6.93147180369123816490e-01, ln2l = 1.90821492927058770002e-10, rth = 0.70710678118654752;
  double a3 = 1.0 / 3, a5 = 1.0 / 5, a7 = 1.0 / 7, a9 = 1.0 / 9, a11 = 1.0 / 11, a13 = 1.0 / 13, a15 = 1.0 / 15, a17 = 1.0 / 17,
         a19 = 1.0 / 19, a21 = 1.0 / 21, a23 = 1.0 / 23;
  asm volatile("" : "+s"(ln2h), "+s"(ln2l), "+s"(rth), "+s"(a3), "+s"(a5), "+s"(a7), "+s"(a9), "+s"(a11), "+s"(a13), "+s"(a15),
               "+s"(a17), "+s"(a19), "+s"(a21), "+s"(a23));
  const double y = 1.0 + x;
  const double c = x - (y - 1.0);
  int k = __builtin_amdgcn_frexp_exp(y);                    // y = mant * 2^k, mant in [0.5, 1)
  double m = __builtin_amdgcn_frexp_mant(y);
  const bool low = m < rth;                                  // bring m into [sqrt(1/2), sqrt(2))
  m = low ? 2.0 * m : m;
  k = low ? k - 1 : k;
  const double s = (m - 1.0) * nrcp(m + 1.0);
  const double s2 = s * s;
  double p = a23;
  p = __builtin_fma(p, s2, a21);
  p = __builtin_fma(p, s2, a19);
  p = __builtin_fma(p, s2, a17);
  p = __builtin_fma(p, s2, a15);
  p = __builtin_fma(p, s2, a13);
  p = __builtin_fma(p, s2, a11);
  p = __builtin_fma(p, s2, a9);
  p = __builtin_fma(p, s2, a7);
  p = __builtin_fma(p, s2, a5);
  p = __builtin_fma(p, s2, a3);
  const double lm = 2.0 * __builtin_fma(p * s2, s, s);
  const double kf = (double)k;
  return __builtin_fma(kf, ln2h, lm) + __builtin_fma(kf, ln2l, c * nrcp(y));
}

}  // namespace pnp
