// Method-of-lines time integration on the device: Dormand-Prince 5(4) per lane.
//
// The reference hands ode_func (catint/calculator_old.py:827-935) to scipy.integrate.ode(...).set_integrator('dopri5', nsteps=10000)
// and calls r.integrate(r.t + dt) once per output interval (:955-963).  scipy's 'dopri5' wraps E. Hairer's DOPRI5 (Hairer, Norsett,
// Wanner, Solving ODEs I, II.4-II.5) -- restated and pinned bit for bit against scipy in oracle/dopri5.py.  Round 1 evaluated the
// right-hand side on the device and left the integrator in scipy: one host -> device -> host round trip of the full state per
// evaluation.  Here the whole DOPCOR loop runs on the device, every lane with its own step size, error estimate and accept / reject
// history; the host only enqueues steps and reads one counter every few of them (as in pnp_scf.hip).
//
// Per attempted step: six stage kernels (element-wise, y + h sum a_sj k_j, products and sums rounded one by one in the order of
// the Fortran source), six right-hand-side evaluations (the kernels of pnp_kernels.hip, on all lanes), and one control kernel with
// a workgroup per lane: error norm, step-size controller with Lund stabilisation, stiffness detection, commit (y <- y1, k1 <- k7:
// first same as last) or rejection, and the checks that open the next step (NMAX, step size below round-off, last step of the
// interval).  Lanes that have reached the end of the interval (or failed) are masked in every kernel of this file.
//
// Against the oracle the only difference is the order of the sums inside the norms (tree per workgroup instead of sequential): the
// step sizes agree to rounding and the accept / reject sequence is the same unless an error estimate lands within rounding of 1.
#include <hip/hip_runtime.h>

#include <cmath>

#include "pnp_internal.h"
#include "pnp_dop853_coeffs.h"

namespace pnp {

namespace {

constexpr double C_A21 = 0.2;
constexpr double C_A31 = 3.0 / 40.0, C_A32 = 9.0 / 40.0;
constexpr double C_A41 = 44.0 / 45.0, C_A42 = -56.0 / 15.0, C_A43 = 32.0 / 9.0;
constexpr double C_A51 = 19372.0 / 6561.0, C_A52 = -25360.0 / 2187.0, C_A53 = 64448.0 / 6561.0, C_A54 = -212.0 / 729.0;
constexpr double C_A61 = 9017.0 / 3168.0, C_A62 = -355.0 / 33.0, C_A63 = 46732.0 / 5247.0, C_A64 = 49.0 / 176.0, C_A65 = -5103.0 / 18656.0;
constexpr double C_A71 = 35.0 / 384.0, C_A73 = 500.0 / 1113.0, C_A74 = 125.0 / 192.0, C_A75 = -2187.0 / 6784.0, C_A76 = 11.0 / 84.0;
constexpr double C_E1 = 71.0 / 57600.0, C_E3 = -71.0 / 16695.0, C_E4 = 71.0 / 1920.0, C_E5 = -17253.0 / 339200.0, C_E6 = 22.0 / 525.0,
                 C_E7 = -1.0 / 40.0;
constexpr double UROUND = 2.3e-16;
constexpr int TPB = 256;

__device__ __forceinline__ double mul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double add(double a, double b) { return __dadd_rn(a, b); }

// The checks that open a step (DOPCOR label 1): NMAX, step size below round-off, last step of the interval.
__device__ void open_step(const OdeArgs& A, int64_t b) {
  double* d = A.d + b * ODE_ND;
  int32_t* s = A.i + b * ODE_NI;
  if (s[ODE_NSTEP] > A.nmax) {
    s[ODE_IDID] = -2;
    s[ODE_ACTIVE] = 0;
    return;
  }
  double h = d[ODE_H];
  const double x = d[ODE_X];
  if (0.1 * fabs(h) <= fabs(x) * UROUND) {
    s[ODE_IDID] = -3;
    s[ODE_ACTIVE] = 0;
    return;
  }
  if (add(add(x, mul(1.01, h)), -d[ODE_XEND]) > 0.0) {
    h = add(d[ODE_XEND], -x);
    d[ODE_H] = h;
    s[ODE_LAST] = 1;
  }
  s[ODE_NSTEP] += 1;
  s[ODE_TOT_NSTEP] += 1;
}

// block-wide sums of up to three values (tree; every thread returns the totals)
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* red) {
  const int t = threadIdx.x;
  red[t] = a;
  red[TPB + t] = b;
  red[2 * TPB + t] = c;
  __syncthreads();
  for (int w = TPB / 2; w > 0; w >>= 1) {
    if (t < w) {
      red[t] += red[t + w];
      red[TPB + t] += red[TPB + t + w];
      red[2 * TPB + t] += red[2 * TPB + t + w];
    }
    __syncthreads();
  }
  a = red[0];
  b = red[TPB];
  c = red[2 * TPB];
  __syncthreads();
}

}  // namespace

// One thread per lane: a new DOPRI5 call (= one r.integrate(r.t + dt) of the reference's loop).
__global__ __launch_bounds__(TPB) void ode_begin_kernel(const OdeArgs A) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= A.B) return;
  double* d = A.d + b * ODE_ND;
  int32_t* s = A.i + b * ODE_NI;
  if (s[ODE_IDID] < 0) {       // r.successful() is false: the reference's loop has stopped for this lane
    s[ODE_ACTIVE] = 0;
    return;
  }
  const double x = d[ODE_X];
  const double xend = add(x, A.dt);
  d[ODE_XEND] = xend;
  d[ODE_HMAX] = fabs(A.max_step != 0.0 ? A.max_step : add(xend, -x));
  d[ODE_FACOLD] = 1.0e-4;
  d[ODE_HLAMB] = 0.0;
  s[ODE_LAST] = 0;
  s[ODE_REJECT] = 0;
  s[ODE_IASTI] = 0;
  s[ODE_NONSTI] = 0;
  s[ODE_NSTEP] = 0;
  s[ODE_NACCPT] = 0;
  s[ODE_ACTIVE] = 1;
  s[ODE_PENDING] = 0;
  s[ODE_TOT_NFCN] += 2;
  s[ODE_INTERVAL] = A.interval;
}

// HINIT, first half (workgroup per lane): norms of y and f0, explicit Euler step y1 = y + h f0.
__global__ __launch_bounds__(TPB) void ode_hinit_a_kernel(const OdeArgs A) {
  __shared__ double red[3 * TPB];
  __shared__ double hs;
  const int64_t b = blockIdx.x;
  double* d = A.d + b * ODE_ND;
  const int32_t* s = A.i + b * ODE_NI;
  if (!s[ODE_ACTIVE] || d[ODE_H] != 0.0) return;
  const size_t base = (size_t)b * A.N * A.ldx;
  double dnf = 0.0, dny = 0.0, zero = 0.0;
  const int n = A.N * A.nx;
  for (int e = threadIdx.x; e < n; e += TPB) {
    const size_t o = base + (size_t)(e / A.nx) * A.ldx + (e % A.nx);
    const double y = A.y[o], f0 = A.k[0][o];
    const double sk = add(A.atol, mul(A.rtol, fabs(y)));
    const double a = f0 / sk, c = y / sk;
    dnf += a * a;
    dny += c * c;
  }
  block_sum3(dnf, dny, zero, red);
  if (threadIdx.x == 0) {
    double h = (dnf <= 1e-10 || dny <= 1e-10) ? 1.0e-6 : sqrt(dny / dnf) * 0.01;
    h = fmin(h, d[ODE_HMAX]);
    d[ODE_HTRY] = h;
    d[ODE_DNF] = dnf;
    hs = h;
  }
  __syncthreads();
  const double h = hs;
  for (int e = threadIdx.x; e < n; e += TPB) {
    const size_t o = base + (size_t)(e / A.nx) * A.ldx + (e % A.nx);
    A.y1[o] = add(A.y[o], mul(h, A.k[0][o]));
  }
}

// HINIT, second half: second-derivative estimate from f1 = f(y1) (in k2) -> first step size.
__global__ __launch_bounds__(TPB) void ode_hinit_b_kernel(const OdeArgs A) {
  __shared__ double red[3 * TPB];
  const int64_t b = blockIdx.x;
  double* d = A.d + b * ODE_ND;
  const int32_t* s = A.i + b * ODE_NI;
  if (!s[ODE_ACTIVE] || d[ODE_H] != 0.0) return;
  const size_t base = (size_t)b * A.N * A.ldx;
  double der2 = 0.0, z1 = 0.0, z2 = 0.0;
  const int n = A.N * A.nx;
  for (int e = threadIdx.x; e < n; e += TPB) {
    const size_t o = base + (size_t)(e / A.nx) * A.ldx + (e % A.nx);
    const double sk = add(A.atol, mul(A.rtol, fabs(A.y[o])));
    const double a = add(A.k[1][o], -A.k[0][o]) / sk;
    der2 += a * a;
  }
  block_sum3(der2, z1, z2, red);
  if (threadIdx.x == 0) {
    const double h = d[ODE_HTRY];
    der2 = sqrt(der2) / h;
    const double der12 = fmax(fabs(der2), sqrt(d[ODE_DNF]));
    const double h1 = der12 <= 1e-15 ? fmax(1.0e-6, fabs(h) * 1.0e-3) : pow(0.01 / der12, A.hinit_expo);
    d[ODE_H] = fmin(fmin(100 * fabs(h), h1), d[ODE_HMAX]);
  }
}

// One thread per lane, after k1 (and HINIT): the checks of the first step of the interval.
__global__ __launch_bounds__(TPB) void ode_open_kernel(const OdeArgs A) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= A.B) return;
  if (!A.i[b * ODE_NI + ODE_ACTIVE]) return;
  open_step(A, b);
  if (A.i[b * ODE_NI + ODE_ACTIVE]) atomicAdd(&A.counters[A.slot], 1);      // lanes that enter the interval's step loop
}

// Stage argument S = 2..7: out = y + h (a_S1 k1 + ...), one workgroup per row (lane, species).
template <int S>
__global__ __launch_bounds__(TPB) void ode_stage_kernel(const OdeArgs A) {
  const int64_t row = blockIdx.x;
  const int64_t b = row / A.N;
  if (!A.i[b * ODE_NI + ODE_ACTIVE]) return;
  const double h = A.d[b * ODE_ND + ODE_H];
  const size_t base = (size_t)row * A.ldx;
  double* out = (S == 6 ? A.ysti : A.y1) + base;
  const double* y = A.y + base;
  const double *k1 = A.k[0] + base, *k2 = A.k[1] + base, *k3 = A.k[2] + base, *k4 = A.k[3] + base, *k5 = A.k[4] + base,
               *k6 = A.k[5] + base;
  for (int x = threadIdx.x; x < A.nx; x += TPB) {
    double v;
    if constexpr (S == 2) v = mul(mul(h, C_A21), k1[x]);
    if constexpr (S == 3) v = mul(h, add(mul(C_A31, k1[x]), mul(C_A32, k2[x])));
    if constexpr (S == 4) v = mul(h, add(add(mul(C_A41, k1[x]), mul(C_A42, k2[x])), mul(C_A43, k3[x])));
    if constexpr (S == 5) v = mul(h, add(add(add(mul(C_A51, k1[x]), mul(C_A52, k2[x])), mul(C_A53, k3[x])), mul(C_A54, k4[x])));
    if constexpr (S == 6)
      v = mul(h, add(add(add(add(mul(C_A61, k1[x]), mul(C_A62, k2[x])), mul(C_A63, k3[x])), mul(C_A64, k4[x])), mul(C_A65, k5[x])));
    if constexpr (S == 7)
      v = mul(h, add(add(add(add(mul(C_A71, k1[x]), mul(C_A73, k3[x])), mul(C_A74, k4[x])), mul(C_A75, k5[x])), mul(C_A76, k6[x])));
    out[x] = add(y[x], v);
  }
}

// Error estimate, controller, commit / reject, opening of the next step: one workgroup per lane.  k7 = f(y1) sits in k2's buffer
// (as in the Fortran source: a_72 = e_2 = 0).
__global__ __launch_bounds__(TPB) void ode_control_kernel(const OdeArgs A) {
  __shared__ double red[3 * TPB];
  __shared__ int accept_s;
  const int64_t b = blockIdx.x;
  double* d = A.d + b * ODE_ND;
  int32_t* s = A.i + b * ODE_NI;
  if (!s[ODE_ACTIVE]) return;
  const double h = d[ODE_H];
  const size_t base = (size_t)b * A.N * A.ldx;
  const int n = A.N * A.nx;
  const bool stiff_check = ((s[ODE_NACCPT] + 1) % A.nstiff == 0) || s[ODE_IASTI] > 0;
  double err = 0.0, stnum = 0.0, stden = 0.0;
  for (int e = threadIdx.x; e < n; e += TPB) {
    const size_t o = base + (size_t)(e / A.nx) * A.ldx + (e % A.nx);
    const double k7 = A.k[1][o], k6 = A.k[5][o], y1 = A.y1[o];
    double v = add(mul(C_E1, A.k[0][o]), mul(C_E3, A.k[2][o]));
    v = add(v, mul(C_E4, A.k[3][o]));
    v = add(v, mul(C_E5, A.k[4][o]));
    v = add(v, mul(C_E6, k6));
    v = mul(add(v, mul(C_E7, k7)), h);
    const double sk = add(A.atol, mul(A.rtol, fmax(fabs(A.y[o]), fabs(y1))));
    const double q = v / sk;
    err += q * q;
    if (stiff_check) {
      const double a = add(k7, -k6), c = add(y1, -A.ysti[o]);
      stnum += a * a;
      stden += c * c;
    }
  }
  block_sum3(err, stnum, stden, red);
  double hnew = 0.0, fac11 = 0.0;
  if (threadIdx.x == 0) {
    err = sqrt(err / n);
    d[ODE_ERR] = err;
    fac11 = pow(err, A.expo1);
    double fac = fac11 / pow(d[ODE_FACOLD], A.beta);
    fac = fmax(A.facc2, fmin(A.facc1, fac / A.safe));
    hnew = h / fac;
    int accept = err <= 1.0 ? 1 : 0;
    s[ODE_TOT_NFCN] += 6;
    if (accept) {
      d[ODE_FACOLD] = fmax(err, 1.0e-4);
      s[ODE_NACCPT] += 1;
      s[ODE_TOT_NACCPT] += 1;
      if (stiff_check) {
        if (stden > 0.0) d[ODE_HLAMB] = h * sqrt(stnum / stden);
        if (d[ODE_HLAMB] > 3.25) {
          s[ODE_NONSTI] = 0;
          s[ODE_IASTI] += 1;
          if (s[ODE_IASTI] == 15) {       // "the problem seems to become stiff": DOPRI5 leaves before committing the step
            s[ODE_IDID] = -4;
            s[ODE_ACTIVE] = 0;
            accept = 2;
          }
        } else {
          s[ODE_NONSTI] += 1;
          if (s[ODE_NONSTI] == 6) s[ODE_IASTI] = 0;
        }
      }
    }
    accept_s = accept;
  }
  __syncthreads();
  const int accept = accept_s;
  if (accept == 2) return;
  if (accept == 1) {
    for (int e = threadIdx.x; e < n; e += TPB) {
      const size_t o = base + (size_t)(e / A.nx) * A.ldx + (e % A.nx);
      A.k[0][o] = A.k[1][o];
      A.y[o] = A.y1[o];
    }
  }
  if (threadIdx.x != 0) return;
  if (accept == 1) {
    d[ODE_X] = add(d[ODE_X], h);
    if (s[ODE_LAST]) {          // normal exit: the predicted step goes back into WORK(7) for the next call
      d[ODE_H] = hnew;
      s[ODE_IDID] = 1;
      s[ODE_ACTIVE] = 0;
      return;
    }
    if (fabs(hnew) > d[ODE_HMAX]) hnew = d[ODE_HMAX];
    if (s[ODE_REJECT]) hnew = fmin(fabs(hnew), fabs(h));
    s[ODE_REJECT] = 0;
  } else {
    hnew = h / fmin(A.facc1, fac11 / A.safe);
    s[ODE_REJECT] = 1;
    if (s[ODE_NACCPT] >= 1) s[ODE_TOT_NREJCT] += 1;
    s[ODE_LAST] = 0;
  }
  d[ODE_H] = hnew;
  open_step(A, b);
  if (s[ODE_ACTIVE]) atomicAdd(&A.counters[A.slot], 1);
}


// ------------------------------------------------------------------------------------------------------------------------------------
// DOP853 (scipy's 'dop853' = Hairer's dop853.f, restated and pinned against scipy in oracle/dop853.py): 12 stages of order 8, error
// estimate from the embedded 5th- and 3rd-order formulas, otherwise DOPRI5's machinery (open_step, HINIT, controller).  Per attempted
// step: 11 x (stage kernel + right-hand side), control A, the right-hand side at the new state (used only if the step was accepted:
// the accepted step's k1 of the next one and the stiffness test), control B.
// ------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void ode853_stage_kernel(const OdeArgs A, const dop853::Stage S) {
  const int64_t row = blockIdx.x;
  const int64_t b = row / A.N;
  if (!A.i[b * ODE_NI + ODE_ACTIVE]) return;
  const double h = A.d[b * ODE_ND + ODE_H];
  const size_t base = (size_t)row * A.ldx;
  double* out = A.y1 + base;
  const double* y = A.y + base;
  for (int x = threadIdx.x; x < A.nx; x += TPB) {
    double v;
    if (S.n == 1) {
      v = mul(mul(h, S.coef[0]), A.k[S.buf[0]][base + x]);      // Y+H*A21*K1
    } else {
      double acc = mul(S.coef[0], A.k[S.buf[0]][base + x]);
      for (int j = 1; j < S.n; ++j) acc = add(acc, mul(S.coef[j], A.k[S.buf[j]][base + x]));
      v = mul(h, acc);
    }
    out[x] = add(y[x], v);
  }
}

// control A: K4 = sum b_j k_j, new state K5 = y + h K4 (-> ysti), both error norms, decision, controller (dop853.f labels 35 ... 41)
__global__ __launch_bounds__(TPB) void ode853_control_a_kernel(const OdeArgs A) {
  __shared__ double red[3 * TPB];
  const int64_t b = blockIdx.x;
  double* d = A.d + b * ODE_ND;
  int32_t* s = A.i + b * ODE_NI;
  if (!s[ODE_ACTIVE]) return;
  const double h = d[ODE_H];
  const size_t base = (size_t)b * A.N * A.ldx;
  const int n = A.N * A.nx;
  double err = 0.0, err2 = 0.0, zero = 0.0;
  for (int e = threadIdx.x; e < n; e += TPB) {
    const size_t o = base + (size_t)(e / A.nx) * A.ldx + (e % A.nx);
    double kk[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) kk[j] = A.k[dop853::kBBuf[j]][o];
    double k4 = mul(dop853::kB[0], kk[0]);
    double e5 = mul(dop853::kER[0], kk[0]);
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      k4 = add(k4, mul(dop853::kB[j], kk[j]));
      e5 = add(e5, mul(dop853::kER[j], kk[j]));
    }
    const double yo = A.y[o];
    const double k5 = add(yo, mul(h, k4));
    A.ysti[o] = k5;
    const double sk = add(A.atol, mul(A.rtol, fmax(fabs(yo), fabs(k5))));
    // ERRI = K4 - BHH1*K1 - BHH2*K9 - BHH3*K12 (k9 = kk[4], k12 = kk[7])
    const double e3 = add(add(add(k4, -mul(dop853::kBHH1, kk[0])), -mul(dop853::kBHH2, kk[4])), -mul(dop853::kBHH3, kk[7]));
    const double q3 = e3 / sk, q5 = e5 / sk;
    err2 += q3 * q3;
    err += q5 * q5;
  }
  block_sum3(err, err2, zero, red);
  if (threadIdx.x != 0) return;
  double deno = err + 0.01 * err2;
  if (deno <= 0.0) deno = 1.0;
  err = fabs(h) * err * sqrt(1.0 / (n * deno));
  d[ODE_ERR] = err;
  const double fac11 = pow(err, A.expo1);
  double fac = fac11 / pow(d[ODE_FACOLD], A.beta);
  fac = fmax(A.facc2, fmin(A.facc1, fac / A.safe));
  double hnew = h / fac;
  s[ODE_TOT_NFCN] += 11;
  if (err <= 1.0) {
    d[ODE_FACOLD] = fmax(err, 1.0e-4);
    s[ODE_NACCPT] += 1;
    s[ODE_TOT_NACCPT] += 1;
    s[ODE_TOT_NFCN] += 1;          // f(new state), evaluated between the two control kernels
    d[ODE_HNEW] = hnew;
    s[ODE_PENDING] = 1;
  } else {
    hnew = h / fmin(A.facc1, fac11 / A.safe);
    s[ODE_REJECT] = 1;
    if (s[ODE_NACCPT] >= 1) s[ODE_TOT_NREJCT] += 1;
    s[ODE_LAST] = 0;
    s[ODE_PENDING] = 0;
    d[ODE_H] = hnew;
    open_step(A, b);
  }
}

// control B: an accepted step -- stiffness detection with f(new state) in k[3] (K4 of dop853.f) against k12 and the argument of stage
// 12, commit (k1 <- f(new state), y <- new state), end of the interval or the next step; counts the lanes still inside the interval
__global__ __launch_bounds__(TPB) void ode853_control_b_kernel(const OdeArgs A) {
  __shared__ double red[3 * TPB];
  __shared__ int go_s;
  const int64_t b = blockIdx.x;
  double* d = A.d + b * ODE_ND;
  int32_t* s = A.i + b * ODE_NI;
  if (!s[ODE_ACTIVE]) return;
  if (!s[ODE_PENDING]) {       // rejected in control A: the next attempt is already open
    if (threadIdx.x == 0) atomicAdd(&A.counters[A.slot], 1);
    return;
  }
  const double h = d[ODE_H];
  const size_t base = (size_t)b * A.N * A.ldx;
  const int n = A.N * A.nx;
  const bool stiff_check = (s[ODE_NACCPT] % A.nstiff == 0) || s[ODE_IASTI] > 0;
  double stnum = 0.0, stden = 0.0, zero = 0.0;
  if (stiff_check) {
    for (int e = threadIdx.x; e < n; e += TPB) {
      const size_t o = base + (size_t)(e / A.nx) * A.ldx + (e % A.nx);
      const double a = add(A.k[3][o], -A.k[2][o]), c = add(A.ysti[o], -A.y1[o]);
      stnum += a * a;
      stden += c * c;
    }
  }
  block_sum3(stnum, stden, zero, red);
  if (threadIdx.x == 0) {
    int go = 1;
    if (stiff_check) {
      if (stden > 0.0) d[ODE_HLAMB] = fabs(h) * sqrt(stnum / stden);
      if (d[ODE_HLAMB] > 6.1) {
        s[ODE_NONSTI] = 0;
        s[ODE_IASTI] += 1;
        if (s[ODE_IASTI] == 15) {
          s[ODE_IDID] = -4;
          s[ODE_ACTIVE] = 0;
          go = 0;
        }
      } else {
        s[ODE_NONSTI] += 1;
        if (s[ODE_NONSTI] == 6) s[ODE_IASTI] = 0;
      }
    }
    s[ODE_PENDING] = 0;
    go_s = go;
  }
  __syncthreads();
  if (!go_s) return;
  for (int e = threadIdx.x; e < n; e += TPB) {
    const size_t o = base + (size_t)(e / A.nx) * A.ldx + (e % A.nx);
    A.k[0][o] = A.k[3][o];
    A.y[o] = A.ysti[o];
  }
  if (threadIdx.x != 0) return;
  double hnew = d[ODE_HNEW];
  d[ODE_X] = add(d[ODE_X], h);
  if (s[ODE_LAST]) {
    d[ODE_H] = hnew;
    s[ODE_IDID] = 1;
    s[ODE_ACTIVE] = 0;
    return;
  }
  if (fabs(hnew) > d[ODE_HMAX]) hnew = d[ODE_HMAX];
  if (s[ODE_REJECT]) hnew = fmin(fabs(hnew), fabs(h));
  s[ODE_REJECT] = 0;
  d[ODE_H] = hnew;
  open_step(A, b);
  if (s[ODE_ACTIVE]) atomicAdd(&A.counters[A.slot], 1);
}

hipError_t launch_ode_begin(const OdeArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(ode_begin_kernel, dim3((unsigned)((a.B + TPB - 1) / TPB)), dim3(TPB), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_ode_hinit(const OdeArgs& a, int half, hipStream_t stream) {
  if (half == 0) hipLaunchKernelGGL(ode_hinit_a_kernel, dim3((unsigned)a.B), dim3(TPB), 0, stream, a);
  else hipLaunchKernelGGL(ode_hinit_b_kernel, dim3((unsigned)a.B), dim3(TPB), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_ode_open(const OdeArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(ode_open_kernel, dim3((unsigned)((a.B + TPB - 1) / TPB)), dim3(TPB), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_ode_stage(const OdeArgs& a, int stage, hipStream_t stream) {
  const dim3 grid((unsigned)(a.B * a.N)), block(TPB);
  switch (stage) {
    case 2: hipLaunchKernelGGL(ode_stage_kernel<2>, grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL(ode_stage_kernel<3>, grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL(ode_stage_kernel<4>, grid, block, 0, stream, a); break;
    case 5: hipLaunchKernelGGL(ode_stage_kernel<5>, grid, block, 0, stream, a); break;
    case 6: hipLaunchKernelGGL(ode_stage_kernel<6>, grid, block, 0, stream, a); break;
    case 7: hipLaunchKernelGGL(ode_stage_kernel<7>, grid, block, 0, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_ode_control(const OdeArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(ode_control_kernel, dim3((unsigned)a.B), dim3(TPB), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_ode853_stage(const OdeArgs& a, int stage, hipStream_t stream) {
  if (stage < 2 || stage > 12) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ode853_stage_kernel, dim3((unsigned)(a.B * a.N)), dim3(TPB), 0, stream, a, dop853::kStages[stage - 2]);
  return hipGetLastError();
}

int ode853_stage_dst(int stage) { return dop853::kStages[stage - 2].dst; }

hipError_t launch_ode853_control(const OdeArgs& a, int half, hipStream_t stream) {
  if (half == 0) hipLaunchKernelGGL(ode853_control_a_kernel, dim3((unsigned)a.B), dim3(TPB), 0, stream, a);
  else hipLaunchKernelGGL(ode853_control_b_kernel, dim3((unsigned)a.B), dim3(TPB), 0, stream, a);
  return hipGetLastError();
}

}  // namespace pnp
