// Physical mode of the batched 1D PNP transport path for MI355X (gfx950): fully implicit (backward-Euler or
// stationary) coupled Poisson + N-species drift-diffusion, ONE nonlinear system per operating point and timestep,
// solved by damped Newton with a block-tridiagonal Jacobian ((N+1)x(N+1) blocks: N species + potential per grid
// point) and block parallel cyclic reduction.
//
// The reference hands this solve to COMSOL (catint/comsol_wrapper.py:145,158); the physics is what its generator
// states (catint/comsol_model.py): Poisson :609-612/:1003-1009, Nernst-Planck with the
// size-modified drift :682-919/:1041-1063, wall flux :770, bulk Dirichlet :771-772, Stern Robin wall :613/:982.
// Discretisation, scaling, damping and convergence test are documented in oracle/pnp_physical.py (the CPU
// restatement the parity tests compare with); this file evaluates the same formulas.
//
// Mapping: one operating point per workgroup; a thread owns whole block rows (all N+1 unknowns of a grid point), so
// the dense (N+1)^3 block algebra of cyclic reduction runs in registers with no cross-lane traffic, and rows are
// exchanged between PCR levels through element-major ping-pong buffers (coalesced: consecutive threads touch
// consecutive addresses).  The buffers live in LDS when both fit, else in a per-workgroup slice of device memory that
// stays L2/MALL resident.
#include <hip/hip_runtime.h>

#include "pnp_internal.h"

namespace pnp {

__device__ __forceinline__ double nrcp(double x) {   // v_rcp_f64 + two Newton steps (1.1e-16 relative)
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}

// Bernoulli function B(u) = u/(exp(u)-1) and its derivative; Taylor series below |u| = 0.05
// (oracle/pnp_physical.py: bernoulli, SERIES_U).
__device__ __forceinline__ void bernoulli(double u, double& B, double& dB) {
  if (fabs(u) < 0.05) {
    const double u2 = u * u;
    B = 1.0 - 0.5 * u + u2 * (1.0 / 12.0 + u2 * (-1.0 / 720.0 + u2 * (1.0 / 30240.0)));
    dB = -0.5 + u * (1.0 / 6.0 + u2 * (-1.0 / 180.0 + u2 * (1.0 / 5040.0)));
  } else {
    const double E = expm1(u);
    B = u / E;
    dB = (1.0 - B - u) / E;
  }
}

// X <- M^-1 X for a dense NB x NB block M and NC right-hand-side columns, Gauss-Jordan in registers.
// PIVOT: partial (row) pivoting, used for the raw Jacobian blocks; the PCR levels work on I - (small products)
// and run without.
template <int NB, int NC, bool PIVOT>
__device__ __forceinline__ void block_solve(double (&M)[NB][NB], double (&X)[NB][NC]) {
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    if constexpr (PIVOT) {
#pragma unroll
      for (int j = k + 1; j < NB; ++j) {
        const bool sw = fabs(M[j][k]) > fabs(M[k][k]);
#pragma unroll
        for (int cc = k; cc < NB; ++cc) {
          const double a = M[k][cc], b = M[j][cc];
          M[k][cc] = sw ? b : a;
          M[j][cc] = sw ? a : b;
        }
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
          const double a = X[k][cc], b = X[j][cc];
          X[k][cc] = sw ? b : a;
          X[j][cc] = sw ? a : b;
        }
      }
    }
    const double inv = nrcp(M[k][k]);
#pragma unroll
    for (int cc = k + 1; cc < NB; ++cc) M[k][cc] *= inv;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) X[k][cc] *= inv;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (j == k) continue;
      const double f = M[j][k];
#pragma unroll
      for (int cc = k + 1; cc < NB; ++cc) M[j][cc] = __builtin_fma(-f, M[k][cc], M[j][cc]);
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) X[j][cc] = __builtin_fma(-f, X[k][cc], X[j][cc]);
    }
  }
}

// Element-major row store: element e of block row `row` lives at buf[e*RS + row].
// Element order: Ltilde (NB*NB, row-major), Utilde (NB*NB), rtilde (NB).
template <int NB>
__device__ __forceinline__ void store_row(double* __restrict__ buf, int RS, int row, const double (&X)[NB][2 * NB + 1]) {
  double* p = buf + row;
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int cc = 0; cc < NB; ++cc) {
      p[(size_t)(r * NB + cc) * RS] = X[r][cc];
      p[(size_t)(NB * NB + r * NB + cc) * RS] = X[r][NB + cc];
    }
#pragma unroll
  for (int r = 0; r < NB; ++r) p[(size_t)(2 * NB * NB + r) * RS] = X[r][2 * NB];
}

template <int NB>
__device__ __forceinline__ void load_block(const double* __restrict__ buf, int RS, int row, int which, double (&Bk)[NB][NB]) {
  const double* p = buf + row + (size_t)which * NB * NB * RS;
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int cc = 0; cc < NB; ++cc) Bk[r][cc] = p[(size_t)(r * NB + cc) * RS];
}

template <int NB>
__device__ __forceinline__ void load_rhs(const double* __restrict__ buf, int RS, int row, double (&v)[NB]) {
  const double* p = buf + row + (size_t)2 * NB * NB * RS;
#pragma unroll
  for (int r = 0; r < NB; ++r) v[r] = p[(size_t)r * RS];
}

// One PCR level for block row `row` (unit diagonal block):  Lt x[row-s] + x[row] + Ut x[row+s] = rt.
// Substituting rows row-s and row+s (absent rows beyond either end contribute nothing):
//   (I - Lt Ut[-s] - Ut Lt[+s]) x[row] - Lt Lt[-s] x[row-2s] - Ut Ut[+s] x[row+2s] = rt - Lt rt[-s] - Ut rt[+s]
template <int NB>
__device__ __forceinline__ void pcr_row(const double* __restrict__ src, double* __restrict__ dst, int RS, int row, int s, int n) {
  constexpr int NC = 2 * NB + 1;
  double Lt[NB][NB], Ut[NB][NB], rt[NB];
  load_block<NB>(src, RS, row, 0, Lt);
  load_block<NB>(src, RS, row, 1, Ut);
  load_rhs<NB>(src, RS, row, rt);
  double D[NB][NB], X[NB][NC];
#pragma unroll
  for (int r = 0; r < NB; ++r) {
#pragma unroll
    for (int cc = 0; cc < NB; ++cc) {
      D[r][cc] = (r == cc) ? 1.0 : 0.0;
      X[r][cc] = 0.0;
      X[r][NB + cc] = 0.0;
    }
    X[r][2 * NB] = rt[r];
  }
  if (row - s >= 0) {
    double Q[NB][NB], qv[NB];
    load_block<NB>(src, RS, row - s, 1, Q);   // Ut[-s]
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int cc = 0; cc < NB; ++cc)
#pragma unroll
        for (int j = 0; j < NB; ++j) D[r][cc] = __builtin_fma(-Lt[r][j], Q[j][cc], D[r][cc]);
    load_block<NB>(src, RS, row - s, 0, Q);   // Lt[-s]
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int cc = 0; cc < NB; ++cc)
#pragma unroll
        for (int j = 0; j < NB; ++j) X[r][cc] = __builtin_fma(-Lt[r][j], Q[j][cc], X[r][cc]);
    load_rhs<NB>(src, RS, row - s, qv);
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int j = 0; j < NB; ++j) X[r][2 * NB] = __builtin_fma(-Lt[r][j], qv[j], X[r][2 * NB]);
  }
  if (row + s < n) {
    double Q[NB][NB], qv[NB];
    load_block<NB>(src, RS, row + s, 0, Q);   // Lt[+s]
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int cc = 0; cc < NB; ++cc)
#pragma unroll
        for (int j = 0; j < NB; ++j) D[r][cc] = __builtin_fma(-Ut[r][j], Q[j][cc], D[r][cc]);
    load_block<NB>(src, RS, row + s, 1, Q);   // Ut[+s]
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int cc = 0; cc < NB; ++cc)
#pragma unroll
        for (int j = 0; j < NB; ++j) X[r][NB + cc] = __builtin_fma(-Ut[r][j], Q[j][cc], X[r][NB + cc]);
    load_rhs<NB>(src, RS, row + s, qv);
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int j = 0; j < NB; ++j) X[r][2 * NB] = __builtin_fma(-Ut[r][j], qv[j], X[r][2 * NB]);
  }
  block_solve<NB, NC, false>(D, X);
  store_row<NB>(dst, RS, row, X);
}

// Residual F and Jacobian blocks (L, M, U) of block row i, returned as M and X = [L | U | -F]
// (oracle/pnp_physical.py: residual_and_jacobian; same scaling: species rows dx^2/D_k, Poisson row dx^2/eps).
template <int NB, bool MPB>
__device__ __forceinline__ void assemble_row(const NewtonArgs& A, const double* __restrict__ c, const double* __restrict__ co,
                                             const double* __restrict__ phi, const double* __restrict__ flux,
                                             const double* __restrict__ cb, double phiM, double phiB, int i,
                                             double (&M)[NB][NB], double (&X)[NB][2 * NB + 1]) {
  constexpr int N = NB - 1;
  constexpr int NC = 2 * NB + 1;
  const int nx = A.nx, ldx = A.ldx;
#pragma unroll
  for (int r = 0; r < NB; ++r) {
#pragma unroll
    for (int cc = 0; cc < NB; ++cc) M[r][cc] = 0.0;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) X[r][cc] = 0.0;
  }
  const int im = i > 0 ? i - 1 : 0;
  const int ip = i < nx - 1 ? i + 1 : nx - 1;
  const double pm = phi[im], p0 = phi[i], pp = phi[ip];
  double cm[N], c0[N], cp[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    cm[k] = c[k * ldx + im];
    c0[k] = c[k * ldx + i];
    cp[k] = c[k * ldx + ip];
  }
  double dwm = 0.0, dwp = 0.0;                 // w_i - w_{i-1}, w_{i+1} - w_i with w = -ln(1-phi0)
  double gm[N], g0[N], gp[N];                  // d w / d c_j at i-1, i, i+1
  if constexpr (MPB) {
    double fm = 0.0, f0 = 0.0, fp = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      fm = __builtin_fma(A.vol[k], cm[k], fm);
      f0 = __builtin_fma(A.vol[k], c0[k], f0);
      fp = __builtin_fma(A.vol[k], cp[k], fp);
    }
    const double wm = -log1p(-fm), w0 = -log1p(-f0), wp = -log1p(-fp);
    dwm = w0 - wm;
    dwp = wp - w0;
    const double im_ = 1.0 / (1.0 - fm), i0_ = 1.0 / (1.0 - f0), ip_ = 1.0 / (1.0 - fp);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      gm[k] = A.vol[k] * im_;
      g0[k] = A.vol[k] * i0_;
      gp[k] = A.vol[k] * ip_;
    }
  }
  const bool wall = (i == 0), bulk = (i == nx - 1);
  double rho = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double qb = A.qb[k], sig = A.sig[k];
    rho = __builtin_fma(A.peq[k], c0[k], rho);
    if (bulk) {
      X[k][2 * NB] = -(c0[k] - cb[k]);
      M[k][k] = 1.0;
      continue;
    }
    const double up = qb * (pp - p0) + dwp;
    double Bp, dBp;
    bernoulli(up, Bp, dBp);
    const double Bmp = Bp + up;
    const double Jp = -(Bmp * cp[k] - Bp * c0[k]);
    const double Jup = -((dBp + 1.0) * cp[k] - dBp * c0[k]);
    if (wall) {
      const double F = 0.5 * sig * (c0[k] - co[k * ldx + i]) + Jp - flux[k] * A.fl[k];
      X[k][2 * NB] = -F;
      M[k][k] += 0.5 * sig + Bp;
      X[k][NB + k] += -Bmp;
      M[k][N] += -Jup * qb;
      X[k][NB + N] += Jup * qb;
      if constexpr (MPB) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          M[k][j] += -Jup * g0[j];
          X[k][NB + j] += Jup * gp[j];
        }
      }
    } else {
      const double um = qb * (p0 - pm) + dwm;
      double Bq, dBq;
      bernoulli(um, Bq, dBq);
      const double Bmq = Bq + um;
      const double Jm = -(Bmq * c0[k] - Bq * cm[k]);
      const double Jum = -((dBq + 1.0) * c0[k] - dBq * cm[k]);
      const double F = sig * (c0[k] - co[k * ldx + i]) + Jp - Jm;
      X[k][2 * NB] = -F;
      M[k][k] += sig + Bp + Bmq;
      X[k][NB + k] += -Bmp;
      X[k][k] += -Bq;
      M[k][N] += Jup * (-qb) - Jum * qb;
      X[k][NB + N] += Jup * qb;
      X[k][N] += Jum * qb;
      if constexpr (MPB) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          M[k][j] += -Jup * g0[j] - Jum * g0[j];
          X[k][NB + j] += Jup * gp[j];
          X[k][j] += Jum * gm[j];
        }
      }
    }
  }
  if (bulk) {
    X[N][2 * NB] = -(p0 - phiB);
    M[N][N] = 1.0;
  } else if (wall) {
    if (A.wall_bc == 0) {
      X[N][2 * NB] = -(p0 - phiM);
      M[N][N] = 1.0;
    } else {
      X[N][2 * NB] = -((pp - p0) + A.stern * (phiM - A.phi_pzc - p0));
      M[N][N] = -1.0 - A.stern;
      X[N][NB + N] = 1.0;
    }
  } else {
    X[N][2 * NB] = -(pp - 2.0 * p0 + pm + rho);
#pragma unroll
    for (int k = 0; k < N; ++k) M[N][k] = A.peq[k];
    M[N][N] = -2.0;
    X[N][N] = 1.0;
    X[N][NB + N] = 1.0;
  }
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// One workgroup per operating point (grid-stride over the batch).  blockDim.x = T threads, thread t owns block rows
// t, t+T, ...  Dynamic LDS: the two PCR buffers when A.work == nullptr.
template <int NB, int TMAX, bool MPB>
__global__ __launch_bounds__(TMAX) void newton_kernel(const NewtonArgs A) {
  constexpr int N = NB - 1;
  constexpr int NE = 2 * NB * NB + NB;
  extern __shared__ double newton_lds[];
  __shared__ double red[2][16];
  const int tid = threadIdx.x, T = blockDim.x;
  const int nx = A.nx, ldx = A.ldx, RS = A.RS;
  double* buf0 = A.work ? A.work + (size_t)blockIdx.x * A.work_stride : newton_lds;
  double* buf1 = buf0 + (size_t)NE * RS;
  for (int64_t b = blockIdx.x; b < A.B; b += gridDim.x) {
    double* c = A.c + (size_t)b * N * ldx;
    double* co = A.c_old + (size_t)b * N * ldx;
    double* phi = A.phi + (size_t)b * ldx;
    const double* flux = A.flux + (size_t)b * N;
    const double* cb = A.cbulk + (size_t)b * N;
    const double phiM = A.pb[b * 4 + 0], phiB = A.pb[b * 4 + 1];
    int total_it = 0, st = PNP_STATUS_OK;
    for (int step = 0; step < A.nsteps; ++step) {
      for (int e = tid; e < N * ldx; e += T) co[e] = c[e];
      __syncthreads();
      bool conv = false;
      int it = 1;
      for (; it <= A.maxit; ++it) {
        for (int row = tid; row < nx; row += T) {
          double M[NB][NB], X[NB][2 * NB + 1];
          assemble_row<NB, MPB>(A, c, co, phi, flux, cb, phiM, phiB, row, M, X);
          block_solve<NB, 2 * NB + 1, true>(M, X);
          store_row<NB>(buf0, RS, row, X);
        }
        __syncthreads();
        double* src = buf0;
        double* dst = buf1;
        for (int s = 1; s < nx; s <<= 1) {
          for (int row = tid; row < nx; row += T) pcr_row<NB>(src, dst, RS, row, s, nx);
          __syncthreads();
          double* t_ = src;
          src = dst;
          dst = t_;
        }
        // src holds the Newton update in its rhs slots
        double mphi = 0.0, upd = 0.0;
        for (int row = tid; row < nx; row += T) {
          double du[NB];
          load_rhs<NB>(src, RS, row, du);
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const double ck = c[k * ldx + row];
            upd = fmax(upd, fabs(du[k]) / (fabs(ck) + fabs(cb[k]) + 1e-300));
            if (!(du[k] == du[k])) upd = INFINITY;
          }
          const double a = fabs(du[N]);
          mphi = fmax(mphi, a);
          if (!(a == a)) mphi = INFINITY;
        }
        upd = fmax(upd, mphi * A.vt_inv);
        mphi = wave_max(mphi);
        upd = wave_max(upd);
        if ((tid & 63) == 0) {
          red[0][tid >> 6] = mphi;
          red[1][tid >> 6] = upd;
        }
        __syncthreads();
        mphi = 0.0;
        upd = 0.0;
        for (int w = 0; w < (T >> 6); ++w) {
          mphi = fmax(mphi, red[0][w]);
          upd = fmax(upd, red[1][w]);
        }
        double lam = 1.0;
        if (A.dphi_max > 0.0 && mphi > A.dphi_max) lam = A.dphi_max / mphi;
        for (int row = tid; row < nx; row += T) {
          double du[NB];
          load_rhs<NB>(src, RS, row, du);
          double cn[N], cc_[N];
#pragma unroll
          for (int k = 0; k < N; ++k) {
            cc_[k] = c[k * ldx + row];
            const double t_ = __builtin_fma(lam, du[k], cc_[k]);
            const double lo = 0.1 * cc_[k];
            cn[k] = t_ < lo ? lo : t_;      // a concentration never loses more than 90 % per iteration
          }
          if constexpr (MPB) {               // ... and neither does the free volume fraction 1 - phi0
            double f_old = 0.0, f_new = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
              f_old = __builtin_fma(A.vol[k], cc_[k], f_old);
              f_new = __builtin_fma(A.vol[k], cn[k], f_new);
            }
            const double free_ = 1.0 - f_old;
            const double target = fmax(0.1 * free_, 1e-12);
            if ((1.0 - f_new) < target) {
              const double theta = (free_ - target) / (f_new - f_old);
#pragma unroll
              for (int k = 0; k < N; ++k) cn[k] = __builtin_fma(theta, cn[k] - cc_[k], cc_[k]);
            }
          }
#pragma unroll
          for (int k = 0; k < N; ++k) c[k * ldx + row] = cn[k];
          phi[row] = __builtin_fma(lam, du[N], phi[row]);
        }
        __syncthreads();
        if (upd < A.tol && lam == 1.0) {
          conv = true;
          break;
        }
      }
      total_it += conv ? it : A.maxit + 1;
      if (!conv) st = PNP_STATUS_MAXIT;
    }
    // non-finite state -> NaN status (replaces the NaN test of calculator.py:409-414)
    double bad = 0.0;
    for (int e = tid; e < nx; e += T) {
      double sacc = phi[e];
#pragma unroll
      for (int k = 0; k < N; ++k) sacc += c[k * ldx + e];
      if (!(fabs(sacc) < INFINITY)) bad = 1.0;
    }
    bad = wave_max(bad);
    if ((tid & 63) == 0) red[0][tid >> 6] = bad;
    __syncthreads();
    if (tid == 0) {
      for (int w = 0; w < (T >> 6); ++w) bad = fmax(bad, red[0][w]);
      A.status[b] = bad > 0.0 ? PNP_STATUS_NAN : st;
      A.iters[b] = total_it;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static constexpr size_t kLdsBudget = 160 * 1024 - 512;

int newton_threads(int nb, int nx) {
  const int tmax = nb <= 4 ? 512 : 256;
  const int t = (nx + 63) / 64 * 64;
  return t < tmax ? t : tmax;
}

size_t newton_exchange_doubles(int nb, int nx) {   // both ping-pong buffers of one workgroup
  const size_t rs = (size_t)(nx + 15) / 16 * 16;
  return 2 * (size_t)(2 * nb * nb + nb) * rs;
}

bool newton_exchange_in_lds(int nb, int nx) { return newton_exchange_doubles(nb, nx) * sizeof(double) <= kLdsBudget; }

template <int NB, int TMAX>
static hipError_t launch_newton_nb(const NewtonArgs& a, int blocks, hipStream_t stream) {
  const int T = newton_threads(NB, a.nx);
  const size_t lds = a.work ? 0 : newton_exchange_doubles(NB, a.nx) * sizeof(double);
  if (a.mpb) {
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)newton_kernel<NB, TMAX, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((newton_kernel<NB, TMAX, true>), dim3(blocks), dim3(T), lds, stream, a);
  } else {
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)newton_kernel<NB, TMAX, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((newton_kernel<NB, TMAX, false>), dim3(blocks), dim3(T), lds, stream, a);
  }
  return hipGetLastError();
}

hipError_t launch_newton(const NewtonArgs& a, int blocks, hipStream_t stream) {
  switch (a.N + 1) {
    case 2: return launch_newton_nb<2, 512>(a, blocks, stream);
    case 3: return launch_newton_nb<3, 512>(a, blocks, stream);
    case 4: return launch_newton_nb<4, 512>(a, blocks, stream);
    case 5: return launch_newton_nb<5, 256>(a, blocks, stream);
    case 6: return launch_newton_nb<6, 256>(a, blocks, stream);
    case 7: return launch_newton_nb<7, 256>(a, blocks, stream);
    case 8: return launch_newton_nb<8, 256>(a, blocks, stream);
    case 9: return launch_newton_nb<9, 256>(a, blocks, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace pnp
