// Method-of-lines time integration on the device for STIFF right-hand sides: Runge-Kutta-Chebyshev of second order with error
// control, every lane with its own step size, stage count and spectral-radius estimate.
//
// The reference hands ode_func (catint/calculator_old.py:827-935) to scipy.integrate.odeint (LSODA, :946-948) or to
// ode('vode' | 'lsoda' | ...) (:955-963): one operating point per call, implicit multistep formulas with finite-difference Jacobians
// on the host.  A batch of thousands of operating points wants an integrator that needs nothing but right-hand sides -- those are the
// HIP kernels of pnp_kernels.hip -- and still takes steps far beyond the explicit stability limit dx^2 / (2 D).  That is what
// stabilised explicit methods are for: RKC (B. P. Sommeijer, L. F. Shampine, J. G. Verwer, J. Comput. Appl. Math. 88 (1998) 315-326)
// covers a stretch of the negative real axis of length 0.65 m^2 with m stages (the spectrum of ode_func is diffusion, 4 D / dx^2, and
// dielectric relaxation, sum q mu c / eps: negative real up to the centred drift's small imaginary parts, which the damping 2/13
// admits), chooses m per step from a spectral radius it estimates itself by a nonlinear power iteration, and controls the local error
// like any embedded pair.  Restated from the paper in oracle/rkc.py (CPU, test infrastructure); this file is the same algorithm as a
// per-lane state machine.
//
// One TICK = one right-hand-side evaluation of every lane's argument buffer (`arg` -> `F`, all lanes in one set of launches) + one
// launch of rkc_advance_kernel (a workgroup per lane) that consumes F according to the lane's phase and prepares the lane's next
// argument.  Lanes in different phases -- power iteration, first-step estimate, stage j of m, error estimate -- advance side by side;
// the host only enqueues ticks and reads one counter every few of them.
#include <hip/hip_runtime.h>

#include <cmath>

#include "pnp_internal.h"

namespace pnp {

namespace {

constexpr int TPB = 256;
constexpr double UROUND = 2.22e-16;

enum { PH_INIT = 0, PH_RESUME, PH_F0, PH_RHO, PH_HINIT, PH_STAGE, PH_FINAL };
enum { ACT_NONE = 0, ACT_DECIDE, ACT_RHO, ACT_HINIT, ACT_STEP };

// block-wide sums of two values (tree; every thread returns the totals)
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red) {
  const int t = threadIdx.x;
  red[t] = a;
  red[TPB + t] = b;
  __syncthreads();
  for (int w = TPB / 2; w > 0; w >>= 1) {
    if (t < w) {
      red[t] += red[t + w];
      red[TPB + t] += red[TPB + t + w];
    }
    __syncthreads();
  }
  a = red[0];
  b = red[TPB];
  __syncthreads();
}

}  // namespace

// One thread per lane: a new call (= one output interval of the reference's loop).  The integrator's memory -- step size, spectral
// radius, eigenvector estimate, controller history -- carries over from interval to interval.
__global__ __launch_bounds__(TPB) void rkc_begin_kernel(const RkcArgs A) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= A.B) return;
  double* d = A.d + b * RKC_ND;
  int32_t* s = A.i + b * RKC_NI;
  if (s[RKI_IDID] < 0) {
    s[RKI_ACTIVE] = 0;
    return;
  }
  const double t = d[RKC_T];
  const double tend = (double)(A.interval + 1) * A.dt;
  d[RKC_TEND] = tend;
  const double span = fabs(tend - t);
  d[RKC_HMAX] = (A.max_step != 0.0 && A.max_step < span) ? A.max_step : span;
  s[RKI_NSTEP_CALL] = 0;
  s[RKI_ACTIVE] = 1;
  s[RKI_INTERVAL] = A.interval;
  s[RKI_PHASE] = s[RKI_STARTED] ? PH_RESUME : PH_INIT;
}

// One workgroup per lane: consume F = f(arg) according to the lane's phase, then prepare the next argument.
__global__ __launch_bounds__(TPB) void rkc_advance_kernel(const RkcArgs A) {
  __shared__ double red[2 * TPB];
  __shared__ int sh_act;
  __shared__ double sh_a, sh_b, sh_c;
  const int64_t b = blockIdx.x;
  double* d = A.d + b * RKC_ND;
  int32_t* s = A.i + b * RKC_NI;
  // the next tick's "lanes left" counter is cleared here (its last use lies 63 ticks back, the host only ever reads the slot of
  // the tick it enqueued last): no separate memset per tick
  if (b == 0 && threadIdx.x == 0) A.counters[(A.slot + 1) & 63] = 0;
  if (!s[RKI_ACTIVE]) return;
  const int tid = threadIdx.x;
  const int n = A.N * A.nx;
  const size_t base = (size_t)b * A.N * A.ldx;
  auto off = [&](int e) { return base + (size_t)(e / A.nx) * A.ldx + (e % A.nx); };
  double* yn = A.y;
  double* fn = A.fn;
  double* F = A.F;
  double* arg = A.arg;
  double* yjm2 = A.yjm2;
  double* ev = A.ev;
  const int phase = s[RKI_PHASE];
  const double hmax = d[RKC_HMAX];
  __syncthreads();      // every thread has read the phase before thread 0 changes it
  int act = ACT_NONE;

  // ---------------------------------------------------------------- consume ----------------------------------------------------
  if (phase == PH_INIT) {
    for (int e = tid; e < n; e += TPB) arg[off(e)] = yn[off(e)];
    if (tid == 0) {
      s[RKI_PHASE] = PH_F0;
      s[RKI_STARTED] = 1;
      s[RKI_NEWSPC] = 1;
      s[RKI_JACATT] = 0;
      s[RKI_NSTSIG] = 0;
      s[RKI_FIRST] = 1;
      s[RKI_HAVE_EV] = 0;
      d[RKC_ERROLD] = 0.0;
      d[RKC_HOLD] = 0.0;
      d[RKC_ABSH] = 0.0;
    }
  } else if (phase == PH_RESUME) {
    act = ACT_DECIDE;
  } else if (phase == PH_F0) {
    for (int e = tid; e < n; e += TPB) fn[off(e)] = F[off(e)];
    if (tid == 0) s[RKI_TOT_NFE] += 1;
    act = ACT_DECIDE;
  } else if (phase == PH_RHO) {
    double dfn = 0.0, zero = 0.0;
    for (int e = tid; e < n; e += TPB) {
      const double q = F[off(e)] - fn[off(e)];
      dfn += q * q;
    }
    block_sum2(dfn, zero, red);
    if (tid == 0) {
      const double dfnrm = sqrt(dfn);
      const double dynrm = d[RKC_DYNRM];
      const double sigmal = d[RKC_SIGMA];
      const double sigma = dfnrm / dynrm;
      const int it = s[RKI_RHO_IT];
      d[RKC_SIGMA] = sigma;
      d[RKC_SPRAD] = 1.2 * sigma;
      s[RKI_TOT_NFESIG] += 1;
      int r;      // 0: converged, 1: next iterate along F - fn, 2: next iterate along the state, 3: failed
      if (it >= 2 && fabs(sigma - sigmal) <= fmax(sigma, 1.0 / hmax) * 0.01) r = 0;
      else if (it >= 50) r = 3;
      else r = dfnrm != 0.0 ? 1 : 2;
      s[RKI_RHO_IT] = it + 1;
      sh_act = r;
      sh_a = dfnrm != 0.0 ? dynrm / dfnrm : 0.0;
    }
    __syncthreads();
    const int r = sh_act;
    const double sc = sh_a;
    __syncthreads();
    if (r == 0) {
      for (int e = tid; e < n; e += TPB) ev[off(e)] = arg[off(e)] - yn[off(e)];
      if (tid == 0) {
        s[RKI_HAVE_EV] = 1;
        s[RKI_JACATT] = 1;
        s[RKI_NEWSPC] = 0;
      }
      act = s[RKI_FIRST] ? ACT_HINIT : ACT_STEP;
    } else if (r == 1) {
      for (int e = tid; e < n; e += TPB) arg[off(e)] = yn[off(e)] + (F[off(e)] - fn[off(e)]) * sc;
    } else if (r == 2) {
      const double sq = sqrt(UROUND);
      for (int e = tid; e < n; e += TPB) arg[off(e)] = yn[off(e)] + yn[off(e)] * sq;
    } else if (tid == 0) {
      s[RKI_IDID] = -6;
      s[RKI_ACTIVE] = 0;
    }
    if (r == 3) return;
  } else if (phase == PH_HINIT) {
    double q2 = 0.0, zero = 0.0;
    for (int e = tid; e < n; e += TPB) {
      const double wt = A.atol + A.rtol * fabs(yn[off(e)]);
      const double q = (F[off(e)] - fn[off(e)]) / wt;
      q2 += q * q;
    }
    block_sum2(q2, zero, red);
    if (tid == 0) {
      double absh = d[RKC_ABSH];
      const double hmin = 10.0 * UROUND * fmax(fabs(d[RKC_T]), hmax);
      const double est = absh * sqrt(q2 / n);
      if (0.1 * absh < hmax * sqrt(est)) absh = fmax(0.1 * absh / sqrt(est), hmin);
      else absh = hmax;
      d[RKC_ABSH] = absh;
      s[RKI_FIRST] = 0;
      s[RKI_TOT_NFE] += 1;
    }
    act = ACT_STEP;
  } else if (phase == PH_STAGE) {
    // stage j = 2..m of the three-term recurrence; arg holds Y_{j-1}
    const double w0 = d[RKC_W0], w1 = d[RKC_W1], bjm1 = d[RKC_BJM1], bjm2 = d[RKC_BJM2];
    const double zjm1 = d[RKC_ZJM1], zjm2 = d[RKC_ZJM2], dzjm1 = d[RKC_DZJM1], dzjm2 = d[RKC_DZJM2];
    const double d2zjm1 = d[RKC_D2ZJM1], d2zjm2 = d[RKC_D2ZJM2];
    const double h = d[RKC_H];
    const int j = s[RKI_J], m = s[RKI_M];
    __syncthreads();
    const double zj = 2.0 * w0 * zjm1 - zjm2;
    const double dzj = 2.0 * w0 * dzjm1 - dzjm2 + 2.0 * zjm1;
    const double d2zj = 2.0 * w0 * d2zjm1 - d2zjm2 + 4.0 * dzjm1;
    const double bj = d2zj / (dzj * dzj);
    const double ajm1 = 1.0 - zjm1 * bjm1;
    const double mu = 2.0 * w0 * bj / bjm1;
    const double nu = -bj / bjm2;
    const double mus = mu * w1 / w0;
    const double c0 = 1.0 - mu - nu, hm = h * mus;
    for (int e = tid; e < n; e += TPB) {
      const size_t o = off(e);
      const double yjm1 = arg[o];
      const double y = mu * yjm1 + nu * yjm2[o] + c0 * yn[o] + hm * (F[o] - ajm1 * fn[o]);
      yjm2[o] = yjm1;
      arg[o] = y;
    }
    if (tid == 0) {
      d[RKC_BJM2] = bjm1;
      d[RKC_BJM1] = bj;
      d[RKC_ZJM2] = zjm1;
      d[RKC_ZJM1] = zj;
      d[RKC_DZJM2] = dzjm1;
      d[RKC_DZJM1] = dzj;
      d[RKC_D2ZJM2] = d2zjm1;
      d[RKC_D2ZJM1] = d2zj;
      s[RKI_TOT_NFE] += 1;
      if (j < m) s[RKI_J] = j + 1;
      else s[RKI_PHASE] = PH_FINAL;
    }
  } else {   // PH_FINAL: arg = y_{n+1}, F = f(y_{n+1})
    const double h = d[RKC_H];
    double q2 = 0.0, zero = 0.0;
    for (int e = tid; e < n; e += TPB) {
      const size_t o = off(e);
      const double y0 = yn[o], y1 = arg[o];
      const double wt = A.atol + A.rtol * fmax(fabs(y1), fabs(y0));
      const double est = 0.8 * (y0 - y1) + 0.4 * h * (fn[o] + F[o]);
      const double q = est / wt;
      q2 += q * q;
    }
    block_sum2(q2, zero, red);
    if (tid == 0) {
      const double err = sqrt(q2 / n);
      const double absh = fabs(h);
      const double hmin = 10.0 * UROUND * fmax(fabs(d[RKC_T]), hmax);
      d[RKC_ERR] = err;
      s[RKI_TOT_NFE] += 1;
      s[RKI_TOT_NSTEP] += 1;
      int r;     // 1 accepted, 2 accepted and the interval is done, 0 rejected, 3 failed
      if (!(err <= 1.0)) {
        s[RKI_TOT_NREJCT] += 1;
        const double nabsh = (err != err || err > 1.0e300) ? 0.1 * absh : 0.8 * absh / pow(err, 1.0 / 3.0);
        if (nabsh < hmin) {
          s[RKI_IDID] = -3;
          s[RKI_ACTIVE] = 0;
          r = 3;
        } else {
          d[RKC_ABSH] = nabsh;
          s[RKI_NEWSPC] = s[RKI_JACATT] ? 0 : 1;
          r = 0;
        }
      } else {
        const int naccpt = s[RKI_TOT_NACCPT] + 1;
        s[RKI_TOT_NACCPT] = naccpt;
        const int last = s[RKI_LAST];
        d[RKC_T] = last ? d[RKC_TEND] : d[RKC_T] + h;
        s[RKI_JACATT] = 0;
        const int nstsig = (s[RKI_NSTSIG] + 1) % 25;
        s[RKI_NSTSIG] = nstsig;
        s[RKI_NEWSPC] = nstsig == 0 ? 1 : 0;
        double fac = 10.0;
        if (naccpt == 1) {
          const double t2 = pow(err, 1.0 / 3.0);
          if (0.8 < fac * t2) fac = 0.8 / t2;
        } else {
          const double t1 = 0.8 * absh * pow(d[RKC_ERROLD], 1.0 / 3.0);
          const double t2 = fabs(d[RKC_HOLD]) * pow(err, 2.0 / 3.0);
          if (t1 < fac * t2) fac = t1 / t2;
        }
        d[RKC_ABSH] = fmax(hmin, fmax(0.1, fac) * absh);
        d[RKC_ERROLD] = err;
        d[RKC_HOLD] = h;
        r = last ? 2 : 1;
        if (last) s[RKI_ACTIVE] = 0;       // idid stays 1
      }
      sh_act = r;
    }
    __syncthreads();
    const int r = sh_act;
    __syncthreads();
    if (r == 1 || r == 2) {
      for (int e = tid; e < n; e += TPB) {
        const size_t o = off(e);
        yn[o] = arg[o];
        fn[o] = F[o];
      }
    }
    if (r >= 2) return;
    act = ACT_DECIDE;
  }

  // ---------------------------------------------------------------- prepare ----------------------------------------------------
  if (act == ACT_DECIDE) {
    __syncthreads();
    if (tid == 0) sh_act = s[RKI_NEWSPC] ? ACT_RHO : (s[RKI_FIRST] ? ACT_HINIT : ACT_STEP);
    __syncthreads();
    act = sh_act;
  }
  __syncthreads();      // the element-wise work of the consume part is complete before the buffers are read again
  if (act == ACT_RHO) {
    // first iterate of the power iteration: the last eigenvector estimate (or f_n) scaled to a perturbation of size sqrt(u) |y_n|
    const bool have = s[RKI_HAVE_EV] != 0;
    const double* v = have ? ev : fn;
    double y2 = 0.0, v2 = 0.0;
    for (int e = tid; e < n; e += TPB) {
      const double a = yn[off(e)], c = v[off(e)];
      y2 += a * a;
      v2 += c * c;
    }
    block_sum2(y2, v2, red);
    const double ynrm = sqrt(y2), vnrm = sqrt(v2), sq = sqrt(UROUND);
    double dynrm;
    if (ynrm != 0.0 && vnrm != 0.0) {
      dynrm = ynrm * sq;
      const double sc = dynrm / vnrm;
      for (int e = tid; e < n; e += TPB) arg[off(e)] = yn[off(e)] + v[off(e)] * sc;
    } else if (ynrm != 0.0) {
      dynrm = ynrm * sq;
      for (int e = tid; e < n; e += TPB) arg[off(e)] = yn[off(e)] + yn[off(e)] * sq;
    } else if (vnrm != 0.0) {
      dynrm = UROUND;
      const double sc = dynrm / vnrm;
      for (int e = tid; e < n; e += TPB) arg[off(e)] = v[off(e)] * sc;
    } else {
      dynrm = UROUND;
      for (int e = tid; e < n; e += TPB) arg[off(e)] = dynrm;
    }
    if (tid == 0) {
      d[RKC_DYNRM] = dynrm;
      d[RKC_SIGMA] = 0.0;
      s[RKI_RHO_IT] = 1;
      s[RKI_PHASE] = PH_RHO;
    }
  } else if (act == ACT_HINIT) {
    if (tid == 0) {
      const double hmin = 10.0 * UROUND * fmax(fabs(d[RKC_T]), hmax);
      double absh = hmax;
      if (d[RKC_SPRAD] * absh > 1.0) absh = 1.0 / d[RKC_SPRAD];
      absh = fmax(absh, hmin);
      d[RKC_ABSH] = absh;
      sh_a = absh;
      s[RKI_PHASE] = PH_HINIT;
    }
    __syncthreads();
    const double absh = sh_a;
    for (int e = tid; e < n; e += TPB) arg[off(e)] = yn[off(e)] + absh * fn[off(e)];
  } else if (act == ACT_STEP) {
    if (tid == 0) {
      double absh = fmin(d[RKC_ABSH], hmax);
      const double left = fabs(d[RKC_TEND] - d[RKC_T]);
      int last = 0;
      if (1.1 * absh >= left) {
        absh = left;
        last = 1;
      }
      const double sprad = d[RKC_SPRAD];
      int m = 1 + (int)sqrt(1.54 * absh * sprad + 1.0);
      if (m > A.mmax) {
        m = A.mmax;
        absh = (double)(m * (double)m - 1.0) / (1.54 * sprad);
        last = 0;
      }
      const int nstep = s[RKI_NSTEP_CALL] + 1;
      s[RKI_NSTEP_CALL] = nstep;
      if (nstep > A.nmax) {
        s[RKI_IDID] = -2;
        s[RKI_ACTIVE] = 0;
        sh_act = 0;
      } else {
        if (m > s[RKI_MAXM]) s[RKI_MAXM] = m;
        const double w0 = 1.0 + 2.0 / (13.0 * (double)m * (double)m);
        const double t1 = w0 * w0 - 1.0;
        const double t2 = sqrt(t1);
        const double ag = m * log(w0 + t2);
        const double w1 = sinh(ag) * t1 / (cosh(ag) * m * t2 - w0 * sinh(ag));
        const double b0 = 1.0 / ((2.0 * w0) * (2.0 * w0));
        d[RKC_H] = absh;
        d[RKC_W0] = w0;
        d[RKC_W1] = w1;
        d[RKC_BJM1] = b0;
        d[RKC_BJM2] = b0;
        d[RKC_ZJM1] = w0;
        d[RKC_ZJM2] = 1.0;
        d[RKC_DZJM1] = 1.0;
        d[RKC_DZJM2] = 0.0;
        d[RKC_D2ZJM1] = 0.0;
        d[RKC_D2ZJM2] = 0.0;
        s[RKI_M] = m;
        s[RKI_J] = 2;
        s[RKI_LAST] = last;
        s[RKI_PHASE] = PH_STAGE;
        sh_a = absh * (w1 * b0);
        sh_act = 1;
      }
    }
    __syncthreads();
    if (!sh_act) return;
    const double hm = sh_a;
    for (int e = tid; e < n; e += TPB) {
      const size_t o = off(e);
      const double y0 = yn[o];
      yjm2[o] = y0;
      arg[o] = y0 + hm * fn[o];
    }
  }
  if (tid == 0) atomicAdd(&A.counters[A.slot], 1);
}

hipError_t launch_rkc_begin(const RkcArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(rkc_begin_kernel, dim3((unsigned)((a.B + TPB - 1) / TPB)), dim3(TPB), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_rkc_advance(const RkcArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(rkc_advance_kernel, dim3((unsigned)a.B), dim3(TPB), 0, stream, a);
  return hipGetLastError();
}

}  // namespace pnp
