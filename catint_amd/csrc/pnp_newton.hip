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
// exchanged between reduction levels through element-major buffers (coalesced: consecutive threads touch consecutive
// addresses) in LDS or, for shapes that do not fit, in a per-workgroup slice of device memory.  Two kernels:
// newton_pair_kernel (N <= 4, nx <= 1024: register-resident rows, parallel cyclic reduction through one LDS buffer) and
// newton_kernel (everything else: in-place cyclic reduction).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "pnp_internal.h"
#include "pnp_math.h"

namespace pnp {

// X <- M^-1 X for a dense NB x NB block M and NC right-hand-side columns, Gauss-Jordan in registers.
// PIVOT: partial (row) pivoting, used for the raw Jacobian blocks when steric coupling or reactions fill the species block;
// the PCR levels work on I - (small products) and run without.  Point ions without reactions need none either: a species
// row has a positive diagonal (sigma + B(u) + B(-u') > 0) and couples only to the potential column, so eliminating the
// species columns first only makes the Poisson pivot more negative (-2 - sum_k (dx^2/eps) q_k^2 beta |dJ/du| / M_kk) --
// no cancellation, no growth; the pair-reduced row is a Schur complement of the same structure.
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
__device__ __forceinline__ void store_row(double* buf, int RS, int row, const double (&X)[NB][2 * NB + 1]) {
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
__device__ __forceinline__ void load_block(const double* buf, int RS, int row, int which, double (&Bk)[NB][NB]) {
  const double* p = buf + row + (size_t)which * NB * NB * RS;
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int cc = 0; cc < NB; ++cc) Bk[r][cc] = p[(size_t)(r * NB + cc) * RS];
}

template <int NB>
__device__ __forceinline__ void load_rhs(const double* buf, int RS, int row, double (&v)[NB]) {
  const double* p = buf + row + (size_t)2 * NB * NB * RS;
#pragma unroll
  for (int r = 0; r < NB; ++r) v[r] = p[(size_t)r * RS];
}

// One PCR level for block row `row` (unit diagonal block):  Lt x[row-s] + x[row] + Ut x[row+s] = rt.
// Substituting rows row-s and row+s (absent rows beyond either end contribute nothing):
//   (I - Lt Ut[-s] - Ut Lt[+s]) x[row] - Lt Lt[-s] x[row-2s] - Ut Ut[+s] x[row+2s] = rt - Lt rt[-s] - Ut rt[+s]
template <int NB>
__device__ __forceinline__ void pcr_row(const double* src, double* dst, int RS, int row, int s, int n) {   // dst may be src
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

// The same row operation organised for blocks that do not fit the register file (used for N = 4..6): only D (then D^-1,
// inverted in place) and one operand block stay in registers.  D is accumulated as a sum of outer products, the new row is
// produced column by column (y = -Lt Lm[:,j], x = D^-1 y) and every column is stored as soon as it is complete; the other
// operands are streamed from the row buffer.  Live set ~ 2 NB^2 + 3 NB doubles instead of ~ 6 NB^2 (measured: +20 % at
// N = 6; for N >= 7 even two blocks overflow the 256 registers and the plain version is as fast).  Works in place: the own
// Lt (Ut) block is in registers before its new columns are stored, neighbours are never written at this level.
template <int NB>
__device__ __forceinline__ void cr_row_streamed(double* buf, int RS, int row, int s, int n) {
  const bool hm = row - s >= 0, hp = row + s < n;
  double* own = buf + row;
  const double* nm = buf + (hm ? row - s : row);
  const double* np_ = buf + (hp ? row + s : row);
  auto EL = [&](const double* p, int which, int r, int cc) { return p[(size_t)(which * NB * NB + r * NB + cc) * RS]; };
  auto ER = [&](const double* p, int r) { return p[(size_t)(2 * NB * NB + r) * RS]; };
  // ---- D = I - Lt Um - Ut Lp as a sum of outer products (column j of Lt/Ut times row j of Um/Lp)
  double D[NB][NB];
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int cc = 0; cc < NB; ++cc) D[r][cc] = (r == cc) ? 1.0 : 0.0;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const bool have = half == 0 ? hm : hp;
    const double* nb_ = half == 0 ? nm : np_;
    if (have) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        double a[NB], b[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) a[r] = EL(own, half, r, j);
#pragma unroll
        for (int cc = 0; cc < NB; ++cc) b[cc] = EL(nb_, 1 - half, j, cc);
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
          for (int cc = 0; cc < NB; ++cc) D[r][cc] = __builtin_fma(-a[r], b[cc], D[r][cc]);
      }
    }
  }
  // ---- D <- D^-1 (Gauss-Jordan in place; D = I - small products, no pivoting)
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const double p = nrcp(D[k][k]);
    D[k][k] = 1.0;
#pragma unroll
    for (int cc = 0; cc < NB; ++cc) D[k][cc] *= p;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      if (r == k) continue;
      const double f = D[r][k];
      D[r][k] = 0.0;
#pragma unroll
      for (int cc = 0; cc < NB; ++cc) D[r][cc] = __builtin_fma(-f, D[k][cc], D[r][cc]);
    }
  }
  double xr[NB];        // new right-hand side, before the multiplication with D^-1
#pragma unroll
  for (int r = 0; r < NB; ++r) xr[r] = ER(own, r);
  // ---- the two halves: new Lt = D^-1 (-Lt Lm), new Ut = D^-1 (-Ut Up)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const bool have = half == 0 ? hm : hp;
    const double* nb_ = half == 0 ? nm : np_;
    double W[NB][NB];
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int cc = 0; cc < NB; ++cc) W[r][cc] = EL(own, half, r, cc);
    double* o = own + (size_t)half * NB * NB * RS;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      double col[NB], y[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) col[k] = have ? EL(nb_, half, k, j) : 0.0;
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) acc = __builtin_fma(-W[r][k], col[k], acc);
        y[r] = acc;
      }
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) acc = __builtin_fma(D[r][k], y[k], acc);
        o[(size_t)(r * NB + j) * RS] = acc;
      }
    }
    if (have) {
      double col[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) col[k] = ER(nb_, k);
#pragma unroll
      for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int k = 0; k < NB; ++k) xr[r] = __builtin_fma(-W[r][k], col[k], xr[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) acc = __builtin_fma(D[r][k], xr[k], acc);
    own[(size_t)(2 * NB * NB + r) * RS] = acc;
  }
}

// Back-substitution of cyclic reduction for a row eliminated at stride s:  x[row] = rt - Lt x[row-s] - Ut x[row+s];
// solutions live in the rhs slots.
template <int NB>
__device__ __forceinline__ void cr_backsub_row(double* buf, int RS, int row, int s, int n) {
  double x[NB], Q[NB][NB], xv[NB];
  load_rhs<NB>(buf, RS, row, x);
  if (row - s >= 0) {
    load_block<NB>(buf, RS, row, 0, Q);
    load_rhs<NB>(buf, RS, row - s, xv);
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int j = 0; j < NB; ++j) x[r] = __builtin_fma(-Q[r][j], xv[j], x[r]);
  }
  if (row + s < n) {
    load_block<NB>(buf, RS, row, 1, Q);
    load_rhs<NB>(buf, RS, row + s, xv);
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int j = 0; j < NB; ++j) x[r] = __builtin_fma(-Q[r][j], xv[j], x[r]);
  }
  double* p = buf + row + (size_t)2 * NB * NB * RS;
#pragma unroll
  for (int r = 0; r < NB; ++r) p[(size_t)r * RS] = x[r];
}

// Scharfetter-Gummel flux of one species across one edge (left point l, right point r), scaled by dx/D:
//   J = -(B(-u) c_r - B(u) c_l),  u = psi_r - psi_l;   Ju = dJ/du;   dJ/dc_l = Bp, dJ/dc_r = -Bm
struct Edge {
  double Bp, Bm, J, Ju;
};

__device__ __forceinline__ Edge edge_flux(double u, double cl, double cr, double w) {   // w = dx/h_e (1 on a uniform grid)
  double B, dB;
  if (fabs(u) < 0.05) {   // oracle/pnp_physical.py: bernoulli, SERIES_U
    // the six series coefficients are materialised in scalar registers right here: left to the compiler they are hoisted out
    // of the Newton loop as vector registers and -- in the 128-register kernels -- spilled to scratch and reloaded per call
    double k12 = 1.0 / 12.0, k720 = -1.0 / 720.0, k30240 = 1.0 / 30240.0, k6 = 1.0 / 6.0, k180 = -1.0 / 180.0, k5040 = 1.0 / 5040.0;
    asm volatile("" : "+s"(k12), "+s"(k720), "+s"(k30240), "+s"(k6), "+s"(k180), "+s"(k5040));
    const double u2 = u * u;
    B = 1.0 - 0.5 * u + u2 * (k12 + u2 * (k720 + u2 * k30240));
    dB = -0.5 + u * (k6 + u2 * (k180 + u2 * k5040));
  } else {
    const double rE = nrcp(expm1_sc(u));
    B = u * rE;
    dB = (1.0 - B - u) * rE;
  }
  Edge e;
  e.Bp = w * B;
  e.Bm = w * (B + u);
  e.J = -(e.Bm * cr - e.Bp * cl);
  e.Ju = -w * ((dB + 1.0) * cr - dB * cl);
  return e;
}

// State of one grid point as the assembly needs it: concentrations, potential and (MPB) w = -ln(1-phi0), dw/dc_j.
template <int N, int MODE>
struct Point {
  double c[N], phi, w, g[N], gam;     // gam = 1/(1-phi0): activity coefficient (comsol_model.py:1060)
};

template <int N, int MODE>
__device__ __forceinline__ Point<N, MODE> load_point(const NewtonArgs& A, const double* c, const double* phi,
                                                   int i) {
  constexpr bool MPB = MODE >= 1;
  Point<N, MODE> P;
  P.phi = phi[i];
#pragma unroll
  for (int k = 0; k < N; ++k) P.c[k] = c[k * A.ldx + i];
  P.w = 0.0;
  P.gam = 1.0;
  if constexpr (MPB) {
    double f = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) f = __builtin_fma(A.vol[k], P.c[k], f);
    P.w = -log1p_sc(-f);
    const double inv = 1.0 / (1.0 - f);
    P.gam = inv;
#pragma unroll
    for (int k = 0; k < N; ++k) P.g[k] = A.vol[k] * inv;
  }
  return P;
}

template <int N, int MODE>
__device__ __forceinline__ void edge_fluxes(const NewtonArgs& A, const Point<N, MODE>& Pl, const Point<N, MODE>& Pr, double w,
                                            Edge (&e)[N]) {
  const double dphi = Pr.phi - Pl.phi, dw = Pr.w - Pl.w;
  // constant convection velocity (comsol_model.py:902-903: tds.cdm1 "u" = system['flow rate']): flux + c v, i.e. the drift argument of
  // the edge loses v h_e / D_k = pe_k / w (oracle/pnp_physical.py)
  const double rw = A.convect ? 1.0 / w : 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) e[k] = edge_flux(A.qb[k] * dphi + dw - A.pe[k] * rw, Pl.c[k], Pr.c[k], w);
}

// Residual F and Jacobian blocks (L, M, U) of block row i, returned as M and X = [L | U | -F], from the point states
// Pm, P0, Pp (i-1, i, i+1) and the fluxes em (edge i-1/2) and ep (edge i+1/2)
// (oracle/pnp_physical.py: residual_and_jacobian; same scaling: species rows dx^2/D_k, Poisson row dx^2/eps).
// At the wall the left edge, at the bulk both edges are switched off by 0/1 weights (their values are finite: the
// neighbour index is clamped).
template <int NB, int MODE>
__device__ __forceinline__ void fill_row(const NewtonArgs& A, const double* c, const double* co,
                                         const double* flux, const double* wk,
                                         const double* cb, double phiM, double phiB, int i,
                                         const Point<NB - 1, MODE>& Pm, const Point<NB - 1, MODE>& P0,
                                         const Point<NB - 1, MODE>& Pp, const Edge (&em)[NB - 1], const Edge (&ep)[NB - 1],
                                         double wem, double wep, double vi, double (&M)[NB][NB],
                                         double (&X)[NB][2 * NB + 1]) {
  constexpr int N = NB - 1;
  constexpr int NC = 2 * NB + 1;
  constexpr bool MPB = MODE >= 1;
  constexpr bool REACT = MODE == 2;
  const int nx = A.nx, ldx = A.ldx;
#pragma unroll
  for (int r = 0; r < NB; ++r) {
#pragma unroll
    for (int cc = 0; cc < NB; ++cc) M[r][cc] = 0.0;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) X[r][cc] = 0.0;
  }
  const bool wall = (i == 0), bulk = (i == nx - 1);
  const double wp = bulk ? 0.0 : 1.0;
  const double wm = (wall || bulk) ? 0.0 : 1.0;
  const double ws = bulk ? 0.0 : vi;     // control volume / dx: 1 inside a uniform grid, 1/2 at the wall
  const double wf = wall ? 1.0 : 0.0;
  double rho = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double qb = A.qb[k], sg = ws * A.sig[k];
    rho = __builtin_fma(A.peq[k], P0.c[k], rho);
    const double Jp = wp * ep[k].J, Jup = wp * ep[k].Ju;
    const double Jm = wm * em[k].J, Jum = wm * em[k].Ju;
    double F = sg * (P0.c[k] - co[k * ldx + i]) + Jp - Jm - wf * (flux[k] * A.fl[k]);
    if (bulk) F = P0.c[k] - cb[k];
    X[k][2 * NB] = -F;
    M[k][k] = sg + wp * ep[k].Bp + wm * em[k].Bm + (bulk ? 1.0 : 0.0);
    X[k][NB + k] = -(wp * ep[k].Bm);
    X[k][k] = -(wm * em[k].Bp);
    M[k][N] = -qb * (Jup + Jum);
    X[k][NB + N] = Jup * qb;
    X[k][N] = Jum * qb;
    if constexpr (MPB) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        M[k][j] += -(Jup + Jum) * P0.g[j];
        X[k][NB + j] += Jup * Pp.g[j];
        X[k][j] += Jum * Pm.g[j];
      }
    }
  }
  // homogeneous reactions: mass action in activities a_j = c_j gam, every reaction summed
  // (comsol_model.py:781-867, :1064-1084; oracle/pnp_physical.py: reaction_rates): source -(dx^2/D_k) v_i R_k on the (half)
  // cell, none on the bulk Dirichlet row.  The species indices of the table are run-time values; every access is a
  // compile-time-indexed loop with a compare and the contributions go straight into M and the right-hand side.
  if constexpr (REACT) {
    const ReactionTable* rt = A.rt;
    auto pick = [&](int idx) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) v = (k == idx) ? P0.c[k] : v;
      return v;
    };
    const int nr = rt->n;
    for (int r = 0; r < nr; ++r) {
      const int nl = rt->n_lhs[r], nrh = rt->n_rhs[r];
      for (int side = 0; side < 2; ++side) {
        const int n = side == 0 ? nl : nrh;
        const int32_t* idx = side == 0 ? rt->lhs[r] : rt->rhs[r];
        const double kk = side == 0 ? rt->kf[r] : rt->kr[r];
        if (kk == 0.0) continue;             // n = 0: constant rate (the side consists of excluded species, e.g. H2O)
        double pre = kk;
        for (int a = 0; a < n; ++a) pre *= P0.gam;
        double prod = pre;
        for (int a = 0; a < n; ++a) prod *= pick(idx[a]);
        double dprod[N];
#pragma unroll
        for (int j = 0; j < N; ++j) dprod[j] = MPB ? prod * n * P0.g[j] : 0.0;     // through gam; g = 0 for point ions
        for (int a = 0; a < n; ++a) {
          double rest = pre;
          for (int b2 = 0; b2 < n; ++b2)
            if (b2 != a) rest *= pick(idx[b2]);
          const int ia = idx[a];
#pragma unroll
          for (int j = 0; j < N; ++j) dprod[j] += (j == ia) ? rest : 0.0;
        }
        const double sgn = side == 0 ? 1.0 : -1.0;                    // forward minus backward
        for (int a = 0; a < nl + nrh; ++a) {
          const bool left = a < nl;
          const int jsp = left ? rt->lhs[r][a] : rt->rhs[r][a - nl];
          const double w = left ? -sgn : sgn;                          // educts lose, products gain
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const double wr = (k == jsp) ? w * ws * A.rs[k] : 0.0;
            X[k][2 * NB] = __builtin_fma(wr, prod, X[k][2 * NB]);
#pragma unroll
            for (int jj = 0; jj < N; ++jj) M[k][jj] = __builtin_fma(-wr, dprod[jj], M[k][jj]);
          }
        }
      }
    }
  }
  // first-order surface reactions: flux into the domain nu_k K c_s(0) joins the prescribed wall flux, so the kinetics <->
  // transport fixed point of the SCF loop (calculator.py:294-406) is part of the Newton system
  if (wall && A.n_wk > 0) {
    for (int r = 0; r < A.n_wk; ++r) {
      const int sp = A.wk_species[r];
      const double kr = wk[r];
      const double cs = sp >= 0 ? c[sp * ldx] : 1.0;
      // rate law K g(c_s) E: Langmuir g = c_s/(1 + K_sat c_s), Butler-Volmer E = exp(alpha (phiM - phi(0))) in the Stern-layer
      // drop (docs/source/topics/flux_definition.rst:90-160 of the reference: exp(-(Ga + alpha F (phiM - phi - phiEq))/RT));
      // alpha = K_sat = 0 is the first-order table: g = c_s, dg = 1, same bits as before
      const double al = A.wk_alpha[r], den = 1.0 / (1.0 + A.wk_sat[r] * cs);
      const double E = al != 0.0 ? exp(al * (phiM - P0.phi)) : 1.0;
      const double g = cs * den * E, dg = den * den * E;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const double a = A.wk_nu[r][k] * kr * A.fl[k];
        X[k][2 * NB] += a * g;
#pragma unroll
        for (int j = 0; j < N; ++j) M[k][j] -= (j == sp) ? a * dg : 0.0;
        if (al != 0.0) M[k][N] += a * al * g;
      }
    }
  }
  const double p0 = P0.phi, pp = Pp.phi, pm = Pm.phi;
  if (bulk) {
    X[N][2 * NB] = -(p0 - phiB);
    M[N][N] = 1.0;
  } else if (wall) {
    if (A.wall_bc == 0) {
      X[N][2 * NB] = -(p0 - phiM);
      M[N][N] = 1.0;
    } else {
      X[N][2 * NB] = -(wep * (pp - p0) + A.stern * (phiM - A.phi_pzc - p0));
      M[N][N] = -wep - A.stern;
      X[N][NB + N] = wep;
    }
  } else {
    X[N][2 * NB] = -(wep * (pp - p0) - wem * (p0 - pm) + vi * rho);
#pragma unroll
    for (int k = 0; k < N; ++k) M[N][k] = vi * A.peq[k];
    M[N][N] = -(wep + wem);
    X[N][N] = wem;
    X[N][NB + N] = wep;
  }
}

template <int NB, int MODE>
__device__ __forceinline__ void assemble_row(const NewtonArgs& A, const double* c, const double* co,
                                             const double* phi, const double* flux,
                                             const double* wk, const double* cb, double phiM,
                                             double phiB, int i,
                                             double (&M)[NB][NB], double (&X)[NB][2 * NB + 1]) {
  constexpr int N = NB - 1;
  const int im = i > 0 ? i - 1 : 0;
  const int ip = i < A.nx - 1 ? i + 1 : A.nx - 1;
  const Point<N, MODE> Pm = load_point<N, MODE>(A, c, phi, im);
  const Point<N, MODE> P0 = load_point<N, MODE>(A, c, phi, i);
  const Point<N, MODE> Pp = load_point<N, MODE>(A, c, phi, ip);
  const double wem = A.gw[im], wep = A.gw[i < A.nx - 1 ? i : A.nx - 2];
  Edge em[N], ep[N];
  edge_fluxes<N, MODE>(A, Pm, P0, wem, em);
  edge_fluxes<N, MODE>(A, P0, Pp, wep, ep);
  fill_row<NB, MODE>(A, c, co, flux, wk, cb, phiM, phiB, i, Pm, P0, Pp, em, ep, wem, wep, A.gv[i], M, X);
}

// Workgroup barrier that orders LDS traffic only: it does not wait for outstanding global stores (the parked row) the way
// __syncthreads() does.  For hand-offs that go through LDS alone.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// One workgroup per operating point (grid-stride over the batch).  blockDim.x = T threads, thread t owns block rows
// t, t+T, ...  Dynamic LDS: the row buffer when A.work == nullptr.
// Launch bounds: the kernels of this file are compiled for at most 256 registers per thread (bound 512): two waves per SIMD.
// History: an early version of this kernel (ping-pong PCR through LDS / device memory) gave run-to-run different results when
// built for 512 registers; the present one (in-place cyclic reduction) does not -- its 512-register builds (256 VGPRs + 74...166
// accumulator registers as spill space, no scratch) are bitwise reproducible, equal to the 256-register builds bit for bit and
// unchanged under -ftrivial-auto-var-init=zero / =pattern (no uninitialised local reaches a result; tests/fuzz/trap_repro.py,
// gpurun_out/trap_repro.txt).  The register budget is an occupancy choice, not a correctness requirement; both builds of
// NB = 6, 7 are kept and compared by tests/test_gpu_newton.py::test_register_budget_does_not_change_results.
template <int NB, int TMAX, int MODE>
__global__ __launch_bounds__(TMAX) void newton_kernel(const NewtonArgs A) {
  constexpr int N = NB - 1;
  constexpr bool MPB = MODE >= 1;
  extern __shared__ double newton_lds[];
  __shared__ double red[2][16];
  const int tid = threadIdx.x, T = blockDim.x;
  const int nx = A.nx, ldx = A.ldx, RS = A.RS;
  // N >= 4: the row buffer always lives in device memory, so its accesses compile to global_load/store; a pointer that
  // may also be LDS costs flat instructions (both address paths, both wait counters) -- measured 90 % of the wave time
  // waiting at N = 6 with flat accesses
  double* buf0;
  if constexpr (NB >= 5) buf0 = A.work + (size_t)blockIdx.x * A.work_stride;
  else buf0 = A.work ? A.work + (size_t)blockIdx.x * A.work_stride : newton_lds;
  for (int64_t b = blockIdx.x; b < A.B; b += gridDim.x) {
    if (A.lane_mask && !A.lane_mask[b]) continue;       // frozen lane (block-uniform: no thread reaches a barrier)
    double* c = A.c + (size_t)b * N * ldx;
    double* co = A.c_old + (size_t)b * N * ldx;
    double* phi = A.phi + (size_t)b * ldx;
    const double* flux = A.flux + (size_t)b * N;
    const double* cb = A.cbulk + (size_t)b * N;
    const double* wk = A.wk_k + (size_t)b * PNP_MAX_WALL_REACTIONS;     // per-lane surface rate constants (unused if n_wk = 0)
    const double phiM = A.pb[b * 4 + 0], phiB = A.pb[b * 4 + 1];
    int total_it = 0, st = PNP_STATUS_OK;
    for (int step = 0; step < A.nsteps; ++step) {
      if (!A.ext_old)
        for (int e = tid; e < N * ldx; e += T) co[e] = c[e];
      __syncthreads();
      bool conv = false;
      double upd_prev = INFINITY;       // scaled update of the previous full (undamped) iteration
      int it = 1;
      for (; it <= A.maxit; ++it) {
        for (int row = tid; row < nx; row += T) {
          double M[NB][NB], X[NB][2 * NB + 1];
          assemble_row<NB, MODE>(A, c, co, phi, flux, wk, cb, phiM, phiB, row, M, X);
          block_solve<NB, 2 * NB + 1, (MODE != 0)>(M, X);      // point ions without reactions: pivot-free, see block_solve
          store_row<NB>(buf0, RS, row, X);
        }
        __syncthreads();
        // Block cyclic reduction, in place: at stride s the rows i = 2s-1 (mod 2s) absorb their neighbours i-s, i+s
        // (which nobody writes at this level and which keep their form for the back-substitution).  Total work ~2 nx
        // row operations instead of nx log2(nx) for PCR: this kernel serves the large blocks / long grids whose rows do
        // not fit on chip, where the exchange traffic through device memory is what binds.
        double* src = buf0;
        int s = 1;
        for (; s < nx; s <<= 1) {
          for (int row = 2 * s - 1 + tid * 2 * s; row < nx; row += T * 2 * s) {
            if constexpr (NB >= 5 && NB <= 7) cr_row_streamed<NB>(src, RS, row, s, nx);
            else pcr_row<NB>(src, src, RS, row, s, nx);
          }
          __syncthreads();
        }
        for (s >>= 1; s >= 1; s >>= 1) {
          for (int row = s - 1 + tid * 2 * s; row < nx; row += T * 2 * s) cr_backsub_row<NB>(src, RS, row, s, nx);
          __syncthreads();
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
        if (lam == 1.0) {
          // strict: the update itself is below tol.  With A.estimate the state is accepted as soon as the quadratic error
          // estimate of the state just computed, upd^2/upd_prev (two consecutive contracting full steps), is below tol --
          // the iteration that would only confirm it is skipped (oracle/pnp_physical.py: newton_step(estimate=True))
          if (upd < A.tol || (A.estimate && upd_prev < INFINITY && upd < 0.1 * upd_prev && upd * (upd / upd_prev) < A.tol) ||
              newton_at_rounding_floor(upd, upd_prev, A.tol)) {
            conv = true;
            break;
          }
          upd_prev = upd;
        } else {
          upd_prev = INFINITY;
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
// Fast path: every thread owns the two adjacent block rows a = 2t, b = 2t+1 (nx <= 2*blockDim.x).
//   1. row a is normalised and eliminated against its neighbours in registers (first level of cyclic reduction):
//      the T reduced b-rows form a block-tridiagonal system of half the size;
//   2. parallel cyclic reduction over the T b-rows, the thread's own row resident in registers, neighbour rows
//      exchanged through ONE element-major LDS buffer (write own / barrier / read i-s, i+s / barrier);
//   3. x_a = rt_a - Lt_a x_b[t-1] - Ut_a x_b[t].
// LDS: (2 NB^2 + NB) * T doubles (72 KiB for N = 3, nx = 512 -> two workgroups per CU).
// ------------------------------------------------------------------------------------------------
template <int NB, int TS>
__device__ __forceinline__ void lds_store_row(double* buf, int t, const double (&X)[NB][2 * NB + 1]) {
  double* p = buf + t;
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int cc = 0; cc < 2 * NB + 1; ++cc) p[(r * (2 * NB + 1) + cc) * TS] = X[r][cc];
}

// Q <- columns [c0, c0+NB) of row t's stored [NB][2NB+1] matrix
template <int NB, int TS>
__device__ __forceinline__ void lds_load_block(const double* buf, int t, int c0, double (&Q)[NB][NB]) {
  const double* p = buf + t;
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int cc = 0; cc < NB; ++cc) Q[r][cc] = p[(r * (2 * NB + 1) + c0 + cc) * TS];
}

template <int NB, int TS>
__device__ __forceinline__ void lds_load_rhs(const double* buf, int t, double (&v)[NB]) {
  const double* p = buf + t;
#pragma unroll
  for (int r = 0; r < NB; ++r) v[r] = p[(r * (2 * NB + 1) + 2 * NB) * TS];
}

// acc[:, c0:c0+NB] -= A * Q
template <int NB, int NC>
__device__ __forceinline__ void mm_sub(double (&acc)[NB][NC], int c0, const double (&A)[NB][NB], const double (&Q)[NB][NB]) {
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int cc = 0; cc < NB; ++cc)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[r][c0 + cc] = __builtin_fma(-A[r][j], Q[j][cc], acc[r][c0 + cc]);
}

// Register budget: 5 x 5 blocks (N = 4) need ~330 registers for the thread's two block rows, and their exchange buffer ((2 NB^2 + NB) TS
// doubles: 112 KB at TS = 256) leaves room for one workgroup of at most four waves per CU anyway -- one wave per SIMD, so the whole
// 512-entry file (256 VGPR + 256 AGPR) is the wave's: bound to 256 threads, the spills go to accumulator registers instead of
// scratch (740 B per lane before): 3.45e6 -> 5.24e6 Newton iterations/s at N = 4, nx = 512, B = 1024 (steric ions 2.5e6 -> 4.7e6, reactions
// 2.3e6 -> 4.2e6; profiles/r03_pair_kernel_register_budget_ab.txt).  Smaller blocks fit 256 registers and keep two waves per SIMD: the
// steric N = 3 instances (172-192 B of scratch) lose 14-24 % when given 512 registers and one wave per SIMD instead.
template <int NB, int TS, int MODE>
__global__ __launch_bounds__(NB >= 5 ? 256 : 512) void newton_pair_kernel(const NewtonArgs G) {
  // The scalar parameters (six constants per species, boundary model, tolerances) do not fit the SGPR file next to the
  // pointers: left in the kernel-argument segment they end up as spilled scalars reloaded ~120 times per Newton iteration.
  // A copy in LDS costs broadcast ds_reads instead (+6 % at batch 1024, +15 % at 8192); pointers stay kernel arguments (G)
  // so that their loads remain global_, not flat_.
  __shared__ NewtonArgs sA;
  if (threadIdx.x == 0) sA = G;
  __syncthreads();
  const NewtonArgs& A = sA;
  constexpr int N = NB - 1;
  constexpr bool MPB = MODE >= 1;
  constexpr int NC = 2 * NB + 1;
  extern __shared__ double newton_lds[];
  __shared__ double red[2][16];
  double* xch = newton_lds;
  const int tid = threadIdx.x, T = blockDim.x;
  const int nx = A.nx, ldx = A.ldx;
  // (row indices ra, rb are re-derived from an opaque copy of tid inside the Newton loop, see there)
  double* stash = G.stash + (size_t)blockIdx.x * G.stash_stride;
  for (int64_t b = blockIdx.x; b < G.B; b += gridDim.x) {
    if (G.lane_mask && !G.lane_mask[b]) continue;       // frozen lane (block-uniform: no thread reaches a barrier)
    double* c = G.c + (size_t)b * N * ldx;
    double* co = G.c_old + (size_t)b * N * ldx;
    double* phi = G.phi + (size_t)b * ldx;
    const double* flux = G.flux + (size_t)b * N;
    const double* cb = G.cbulk + (size_t)b * N;
    const double* wk = G.wk_k + (size_t)b * PNP_MAX_WALL_REACTIONS;     // per-lane surface rate constants (unused if n_wk = 0)
    const double phiM = G.pb[b * 4 + 0], phiB = G.pb[b * 4 + 1];
    int total_it = 0, st = PNP_STATUS_OK;
    for (int step = 0; step < A.nsteps; ++step) {
      if (!A.ext_old)
        for (int e = tid; e < N * ldx; e += T) co[e] = c[e];
      __syncthreads();
      bool conv = false;
      double upd_prev = INFINITY;       // scaled update of the previous full (undamped) iteration
      int it = 1;
      for (; it <= A.maxit; ++it) {
        // The thread index is made opaque once per iteration: the compiler then recomputes the ~100 global addresses that
        // depend on it (state rows at four points, parked row) where they are used, instead of precomputing all of them
        // before the loop and reloading them from spill slots -- scratch memory that misses L2 -- every iteration.
        int tv = tid;
        asm volatile("" : "+v"(tv));
        const int ra = 2 * tv, rb = 2 * tv + 1;
        // ---- rows a and b (the edge between them is evaluated once); a normalised
        double Ma[NB][NB], Xa[NB][NC];     // Xa = [Lt_a | Ut_a | rt_a] after the solve
        double Mb[NB][NB], Xb[NB][NC];
        {
          const int last = nx - 1;
          const int i0 = ra > 0 ? (ra - 1 < last ? ra - 1 : last) : 0;
          const int i1 = ra < last ? ra : last, i2 = rb < last ? rb : last, i3 = rb + 1 < last ? rb + 1 : last;
          const Point<N, MODE> P0 = load_point<N, MODE>(A, c, phi, i0);
          const Point<N, MODE> P1 = load_point<N, MODE>(A, c, phi, i1);
          const Point<N, MODE> P2 = load_point<N, MODE>(A, c, phi, i2);
          const Point<N, MODE> P3 = load_point<N, MODE>(A, c, phi, i3);
          const int le = nx - 2;
          const double w0 = G.gw[i0 < le ? i0 : le], w1 = G.gw[i1 < le ? i1 : le], w2 = G.gw[i2 < le ? i2 : le];
          Edge e0[N], e1[N], e2[N];
          edge_fluxes<N, MODE>(A, P0, P1, w0, e0);
          edge_fluxes<N, MODE>(A, P1, P2, w1, e1);
          if (ra < nx) {
            fill_row<NB, MODE>(A, c, co, flux, wk, cb, phiM, phiB, ra, P0, P1, P2, e0, e1, w0, w1, G.gv[i1], Ma, Xa);
            block_solve<NB, NC, (MODE != 0)>(Ma, Xa);
          } else {
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
              for (int cc = 0; cc < NC; ++cc) Xa[r][cc] = 0.0;
          }
          lds_store_row<NB, TS>(xch, tid, Xa);
          __builtin_amdgcn_sched_barrier(0);          // row a is complete before the work on row b starts (register pressure)
          edge_fluxes<N, MODE>(A, P2, P3, w2, e2);
          if (rb < nx) {
            fill_row<NB, MODE>(A, c, co, flux, wk, cb, phiM, phiB, rb, P1, P2, P3, e1, e2, w1, w2, G.gv[i2], Mb, Xb);
          } else {
#pragma unroll
            for (int r = 0; r < NB; ++r) {
#pragma unroll
              for (int cc = 0; cc < NB; ++cc) Mb[r][cc] = (r == cc) ? 1.0 : 0.0;
#pragma unroll
              for (int cc = 0; cc < NC; ++cc) Xb[r][cc] = 0.0;
            }
          }
        }
        lds_barrier();  
        {
          // eliminate x_a (own) and x_a' (thread t+1) from row b:  L_b x_a + M_b x_b + U_b x_a' = r_b.
          // Ordered for a small live set: the own a-row is used first and parked right away, the coupling blocks are
          // consumed row by row (no copies of L_b), then the neighbour's blocks stream through one register block.
          double Ub[NB][NB], Q[NB][NB], qv[NB];
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            double lrow[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
              lrow[j] = Xb[r][j];
              Ub[r][j] = Xb[r][NB + j];
              Xb[r][j] = 0.0;
              Xb[r][NB + j] = 0.0;
            }
            // own a-row: M_b -= L_b Ut_a ; L' = -L_b Lt_a ; r' -= L_b rt_a
#pragma unroll
            for (int j = 0; j < NB; ++j) {
#pragma unroll
              for (int cc = 0; cc < NB; ++cc) {
                Mb[r][cc] = __builtin_fma(-lrow[j], Xa[j][NB + cc], Mb[r][cc]);
                Xb[r][cc] = __builtin_fma(-lrow[j], Xa[j][cc], Xb[r][cc]);
              }
              Xb[r][2 * NB] = __builtin_fma(-lrow[j], Xa[j][2 * NB], Xb[r][2 * NB]);
            }
          }
          // park the normalised a-row in device memory (coalesced) until the back-substitution: it would otherwise cost
          // 2 NB^2 + NB registers through the rest of this phase and every PCR level
          {
            double* sp = stash + tv;
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
              for (int cc = 0; cc < NC; ++cc) sp[(size_t)(r * NC + cc) * TS] = Xa[r][cc];
          }
          __builtin_amdgcn_sched_barrier(0);
          if (tid + 1 < T) {   // a-row of the next thread: M_b -= U_b Lt_a' ; U' = -U_b Ut_a' ; r' -= U_b rt_a'
            lds_load_block<NB, TS>(xch, tid + 1, 0, Q);
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
              for (int cc = 0; cc < NB; ++cc)
#pragma unroll
                for (int j = 0; j < NB; ++j) Mb[r][cc] = __builtin_fma(-Ub[r][j], Q[j][cc], Mb[r][cc]);
            lds_load_block<NB, TS>(xch, tid + 1, NB, Q);
            mm_sub<NB, NC>(Xb, NB, Ub, Q);
            lds_load_rhs<NB, TS>(xch, tid + 1, qv);
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
              for (int j = 0; j < NB; ++j) Xb[r][2 * NB] = __builtin_fma(-Ub[r][j], qv[j], Xb[r][2 * NB]);
          }
          block_solve<NB, NC, (MODE != 0)>(Mb, Xb);
        }
        // ---- PCR over the T reduced rows, own row in registers
        for (int s = 1; s < T; s <<= 1) {
          lds_barrier();                         // everyone has finished reading the previous contents of xch
          lds_store_row<NB, TS>(xch, tid, Xb);
          lds_barrier();  
          double D[NB][NB], Y[NB][NC], Lt[NB][NB], Ut[NB][NB], Q[NB][NB], qv[NB];
#pragma unroll
          for (int r = 0; r < NB; ++r) {
#pragma unroll
            for (int cc = 0; cc < NB; ++cc) {
              D[r][cc] = (r == cc) ? 1.0 : 0.0;
              Lt[r][cc] = Xb[r][cc];
              Ut[r][cc] = Xb[r][NB + cc];
              Y[r][cc] = 0.0;
              Y[r][NB + cc] = 0.0;
            }
            Y[r][2 * NB] = Xb[r][2 * NB];
          }
          if (tid - s >= 0) {
            lds_load_block<NB, TS>(xch, tid - s, NB, Q);     // Ut[-s]
            mm_sub<NB, NB>(D, 0, Lt, Q);
            __builtin_amdgcn_sched_barrier(0);
            lds_load_block<NB, TS>(xch, tid - s, 0, Q);      // Lt[-s]
            mm_sub<NB, NC>(Y, 0, Lt, Q);
            __builtin_amdgcn_sched_barrier(0);
            lds_load_rhs<NB, TS>(xch, tid - s, qv);
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
              for (int j = 0; j < NB; ++j) Y[r][2 * NB] = __builtin_fma(-Lt[r][j], qv[j], Y[r][2 * NB]);
          }
          if (tid + s < T) {
            lds_load_block<NB, TS>(xch, tid + s, 0, Q);      // Lt[+s]
            mm_sub<NB, NB>(D, 0, Ut, Q);
            __builtin_amdgcn_sched_barrier(0);
            lds_load_block<NB, TS>(xch, tid + s, NB, Q);     // Ut[+s]
            mm_sub<NB, NC>(Y, NB, Ut, Q);
            __builtin_amdgcn_sched_barrier(0);
            lds_load_rhs<NB, TS>(xch, tid + s, qv);
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
              for (int j = 0; j < NB; ++j) Y[r][2 * NB] = __builtin_fma(-Ut[r][j], qv[j], Y[r][2 * NB]);
          }
          block_solve<NB, NC, false>(D, Y);
#pragma unroll
          for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) Xb[r][cc] = Y[r][cc];
        }
        // ---- x_b = rt_b; x_a = rt_a - Lt_a x_b[t-1] - Ut_a x_b[t]
        double dub[NB], dua[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) dub[r] = Xb[r][2 * NB];
        lds_barrier();  
#pragma unroll
        for (int r = 0; r < NB; ++r) xch[r * TS + tid] = dub[r];
        lds_barrier();  
        {
          const double* sp = stash + tv;
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            double acc = sp[(size_t)(r * NC + 2 * NB) * TS];
#pragma unroll
            for (int j = 0; j < NB; ++j) acc = __builtin_fma(-sp[(size_t)(r * NC + NB + j) * TS], dub[j], acc);
            dua[r] = acc;
          }
          if (tid > 0) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
              const double xl = xch[j * TS + tid - 1];
#pragma unroll
              for (int r = 0; r < NB; ++r) dua[r] = __builtin_fma(-sp[(size_t)(r * NC + j) * TS], xl, dua[r]);
            }
          }
        }
        // ---- damping, update, convergence (oracle/pnp_physical.py: newton_step)
        double mphi = 0.0, upd = 0.0;
        double ca[N], cbv[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
          ca[k] = ra < nx ? c[k * ldx + ra] : 1.0;
          cbv[k] = rb < nx ? c[k * ldx + rb] : 1.0;
          upd = fmax(upd, fabs(dua[k]) / (fabs(ca[k]) + fabs(cb[k]) + 1e-300));
          upd = fmax(upd, fabs(dub[k]) / (fabs(cbv[k]) + fabs(cb[k]) + 1e-300));
          if (!(dua[k] == dua[k]) || !(dub[k] == dub[k])) upd = INFINITY;
        }
        mphi = fmax(fabs(dua[N]), fabs(dub[N]));
        // fmax drops a NaN operand: test both potential updates, as the other kernels do, so that a NaN update is never accepted
        if (!(dua[N] == dua[N]) || !(dub[N] == dub[N])) mphi = INFINITY;
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
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int row = half == 0 ? ra : rb;
          if (row < nx) {
            double cn[N];
            double f_old = 0.0, f_new = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
              const double cc_ = half == 0 ? ca[k] : cbv[k];
              const double du = half == 0 ? dua[k] : dub[k];
              const double t_ = __builtin_fma(lam, du, cc_);
              const double lo = 0.1 * cc_;
              cn[k] = t_ < lo ? lo : t_;
              if constexpr (MPB) {
                f_old = __builtin_fma(A.vol[k], cc_, f_old);
                f_new = __builtin_fma(A.vol[k], cn[k], f_new);
              }
            }
            if constexpr (MPB) {
              const double free_ = 1.0 - f_old;
              const double target = fmax(0.1 * free_, 1e-12);
              if ((1.0 - f_new) < target) {
                const double theta = (free_ - target) / (f_new - f_old);
#pragma unroll
                for (int k = 0; k < N; ++k) {
                  const double cc_ = half == 0 ? ca[k] : cbv[k];
                  cn[k] = __builtin_fma(theta, cn[k] - cc_, cc_);
                }
              }
            }
#pragma unroll
            for (int k = 0; k < N; ++k) c[k * ldx + row] = cn[k];
            phi[row] = __builtin_fma(lam, half == 0 ? dua[N] : dub[N], phi[row]);
          }
        }
        __syncthreads();
        if (lam == 1.0) {
          // strict: the update itself is below tol.  With A.estimate the state is accepted as soon as the quadratic error
          // estimate of the state just computed, upd^2/upd_prev (two consecutive contracting full steps), is below tol --
          // the iteration that would only confirm it is skipped (oracle/pnp_physical.py: newton_step(estimate=True))
          if (upd < A.tol || (A.estimate && upd_prev < INFINITY && upd < 0.1 * upd_prev && upd * (upd / upd_prev) < A.tol) ||
              newton_at_rounding_floor(upd, upd_prev, A.tol)) {
            conv = true;
            break;
          }
          upd_prev = upd;
        } else {
          upd_prev = INFINITY;
        }
      }
      total_it += conv ? it : A.maxit + 1;
      if (!conv) st = PNP_STATUS_MAXIT;
    }
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
      G.status[b] = bad > 0.0 ? PNP_STATUS_NAN : st;
      G.iters[b] = total_it;
    }
    __syncthreads();
  }
}


// ------------------------------------------------------------------------------------------------
// Lane-team kernel for large blocks (N >= 5): a block row does not fit the registers of one thread (the row-per-thread kernel
// spills and spends 90 % of its time in s_waitcnt), so NB = N+1 adjacent lanes share it -- lane r of a team holds ROW r of
// [Lt | Ut | rt] (2 NB + 1 doubles), i.e. the equation of species r (r < N) or the Poisson equation (r = N).
//   * assembly: every lane evaluates its own equation (two Scharfetter-Gummel fluxes for a species lane); the few
//     quantities that couple the lanes of a point (occupied volume fraction at i-1, i, i+1, charge density) are team sums;
//   * block algebra: products with neighbour blocks read the neighbour's rows as broadcast loads (the NB lanes of a team ask
//     for the same addresses); Gauss-Jordan runs across the team, the pivot row travelling through a per-team LDS strip
//     (partial pivoting over the lanes for the raw Jacobian block, none inside cyclic reduction);
//   * the grid is reduced by in-place block cyclic reduction in a device-memory row buffer, layout [row][r][2 NB + 2].
// ~110 registers per lane (bound: 4 waves per SIMD), no scratch.  64/NB teams per wave; the leftover lanes of a wave
// form a dummy team that runs along (same control flow, no stores).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void team_sync() {   // LDS hand-off inside one wave (all lanes run the same code)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NB>
struct TeamLayout {
  static constexpr int NC = 2 * NB + 1;
  static constexpr int NCP = NC + 1;               // row record padded to an even number of doubles
  static constexpr int TPW = 64 / NB;              // teams per wave (+1 dummy team of the leftover lanes)
  static constexpr int SV = NB;                    // strip: NB scalars (reductions / pivot search) ...
  static constexpr int SL = ((NB + NB + NC) + 1) / 2 * 2;   // ... + pivot row (NB of M/D + NC of X)
  // Staging neighbour block rows in an LDS tile turns ~2 NB dependent memory round trips per row operation into 2, but the
  // tile costs occupancy: measured +15 % at N = 8 and N = 3, -15 % at N = 6 -> only where it pays
  static constexpr bool STAGE = NB >= 8 || NB <= 5;
  static constexpr int SLT = SL + (STAGE ? ((NB * NC) + 1) / 2 * 2 : 0);  // strip (+ tile for one neighbour block row)
};

// Gauss-Jordan across a team on rows Dr (NB) | Xr (NC):  X <- D^-1 X.  PIVOT: the lane with the largest |D[.][k]| among the
// lanes not used yet provides pivot row k; on return `myk` is the unknown whose solution row the lane holds.
template <int NB, int NC, bool PIVOT>
__device__ __forceinline__ void team_solve_n(double (&Dr)[NB], double (&Xr)[NC], double* strip, int r, int& myk) {
  double* sv = strip;             // [NB] scalars
  double* sp = strip + NB;        // pivot row
  bool used = false;
  myk = PIVOT ? -1 : r;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    int piv = k;
    if constexpr (PIVOT) {
      sv[r] = used ? -1.0 : fabs(Dr[k]);
      team_sync();
      double best = -2.0;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const double v = sv[j];
        if (v > best) {
          best = v;
          piv = j;
        }
      }
    }
    if (r == piv) {
      const double inv = nrcp(Dr[k]);
#pragma unroll
      for (int j = k + 1; j < NB; ++j) {
        Dr[j] *= inv;
        sp[j] = Dr[j];
      }
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        Xr[j] *= inv;
        sp[NB + j] = Xr[j];
      }
      used = true;
      myk = k;
    }
    team_sync();
    if (r != piv) {
      const double f = Dr[k];
#pragma unroll
      for (int j = k + 1; j < NB; ++j) Dr[j] = __builtin_fma(-f, sp[j], Dr[j]);
#pragma unroll
      for (int j = 0; j < NC; ++j) Xr[j] = __builtin_fma(-f, sp[NB + j], Xr[j]);
    }
    if constexpr (!PIVOT) team_sync();      // (with pivoting the next search's sync separates the strip reuse)
  }
}

template <int NB, bool PIVOT>
__device__ __forceinline__ void team_solve(double (&Dr)[NB], double (&Xr)[2 * NB + 1], double* strip, int r, int& myk) {
  team_solve_n<NB, 2 * NB + 1, PIVOT>(Dr, Xr, strip, r, myk);
}

// One block row of the Newton system, evaluated by a team: lane r returns row r of the diagonal block (Dr) and of [L | U | rhs]
// (Xr) of grid point i -- the equation of species r (r < N) or the Poisson equation (r = N); see fill_row for the formulas.
// A: scalar parameters (LDS copy), G: kernel arguments (pointers).  `strip` is the team's LDS strip (team sums).
// CARRY (sweep kernel, rows visited in order): what row i-1 evaluated at its right neighbour and on its right edge IS row i's
// centre point and left edge -- volume fraction, steric potential, activity factor, Scharfetter-Gummel edge flux -- so it is
// handed over instead of being evaluated again (same operands, same bits).
struct RowCarry {
  double f0, fp, w0, wp, inv0, invp;
  Edge ep;
};

template <int NB, int MODE, bool CARRY = false>
__device__ __forceinline__ void team_assemble_row(const NewtonArgs& A, const NewtonArgs& G, const double* c, const double* co,
                                                  const double* phi, const double* cb, const double* wk, double phiM, double phiB,
                                                  int64_t b, int i, int r, bool spec, double* strip, double (&Dr)[NB],
                                                  double (&Xr)[2 * NB + 1], RowCarry* carry = nullptr, bool carried = false) {
  constexpr int N = NB - 1, NC = 2 * NB + 1;
  constexpr bool MPB = MODE >= 1, REACT = MODE == 2;
  const int nx = A.nx, ldx = A.ldx;
    const int rs_ = spec ? r : 0;
    const double pe_r = A.pe[rs_];
    const double qb_r = A.qb[rs_], sig_r = A.sig[rs_], fl_r = A.fl[rs_], peq_r = A.peq[rs_], vol_r = A.vol[rs_], rs_r = A.rs[rs_];
    const double flux_r = G.flux[(size_t)b * N + rs_], cb_r = cb[rs_];
    const int im = i > 0 ? i - 1 : 0, ip = i < nx - 1 ? i + 1 : nx - 1;
    const bool wall = (i == 0), bulk = (i == nx - 1);
    const double pm = phi[im], p0 = phi[i], pp = phi[ip];
    const double cm = c[rs_ * ldx + im], c0 = c[rs_ * ldx + i], cp = c[rs_ * ldx + ip];
    // team sums: occupied volume fraction at the three points, scaled charge density at i
    auto team_sum = [&](double v) {
      strip[r] = v;
      team_sync();
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < NB; ++j) acc += strip[j];
      team_sync();
      return acc;
    };
    double fm = 0.0, f0 = 0.0, fp = 0.0;
    double wm_ = 0.0, w0_ = 0.0, wp_ = 0.0, invm = 1.0, inv0 = 1.0, invp = 1.0;
    if constexpr (MPB) {
      if (CARRY && carried) {
        fm = carry->f0;
        f0 = carry->fp;
        wm_ = carry->w0;
        w0_ = carry->wp;
        invm = carry->inv0;
        inv0 = carry->invp;
      } else {
        fm = team_sum(spec ? vol_r * cm : 0.0);
        f0 = team_sum(spec ? vol_r * c0 : 0.0);
        wm_ = -log1p_sc(-fm);
        w0_ = -log1p_sc(-f0);
        invm = 1.0 / (1.0 - fm);
        inv0 = 1.0 / (1.0 - f0);
      }
      fp = team_sum(spec ? vol_r * cp : 0.0);
      wp_ = -log1p_sc(-fp);
      invp = 1.0 / (1.0 - fp);
      if constexpr (CARRY) {
        carry->f0 = f0;
        carry->fp = fp;
        carry->w0 = w0_;
        carry->wp = wp_;
        carry->inv0 = inv0;
        carry->invp = invp;
      }
    }
    const double rho = team_sum(spec ? peq_r * c0 : 0.0);
    const double wem = G.gw[im], wep = G.gw[i < nx - 1 ? i : nx - 2], vi = G.gv[i];
#pragma unroll
    for (int j = 0; j < NB; ++j) Dr[j] = 0.0;
#pragma unroll
    for (int j = 0; j < NC; ++j) Xr[j] = 0.0;
    if (spec) {
      const double wpp = bulk ? 0.0 : 1.0, wmm = (wall || bulk) ? 0.0 : 1.0, ws = bulk ? 0.0 : vi;
      Edge em;
      if (CARRY && carried) em = carry->ep;
      else em = edge_flux(qb_r * (p0 - pm) + (w0_ - wm_) - (A.convect ? pe_r / wem : 0.0), cm, c0, wem);
      const Edge ep = edge_flux(qb_r * (pp - p0) + (wp_ - w0_) - (A.convect ? pe_r / wep : 0.0), c0, cp, wep);
      if constexpr (CARRY) carry->ep = ep;
      const double sg = ws * sig_r;
      const double Jp = wpp * ep.J, Jup = wpp * ep.Ju, Jm = wmm * em.J, Jum = wmm * em.Ju;
      double F = sg * (c0 - co[rs_ * ldx + i]) + Jp - Jm - (wall ? flux_r * fl_r : 0.0);
      if (bulk) F = c0 - cb_r;
      double rhs = -F;
      const double diag = sg + wpp * ep.Bp + wmm * em.Bm + (bulk ? 1.0 : 0.0);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const double own = (j == r) ? 1.0 : 0.0, pot = (j == N) ? 1.0 : 0.0;
        Dr[j] = own * diag - pot * qb_r * (Jup + Jum);
        Xr[NB + j] = -own * (wpp * ep.Bm) + pot * Jup * qb_r;
        Xr[j] = -own * (wmm * em.Bp) + pot * Jum * qb_r;
        if constexpr (MPB) {
          if (j < N) {
            const double vj = A.vol[j];
            Dr[j] += -(Jup + Jum) * vj * inv0;
            Xr[NB + j] += Jup * vj * invp;
            Xr[j] += Jum * vj * invm;
          }
        }
      }
      if constexpr (REACT) {       // mass action in activities, see fill_row
        const ReactionTable* rt = G.rt;
        double call[N];
#pragma unroll
        for (int k = 0; k < N; ++k) call[k] = c[k * ldx + i];
        auto pick = [&](int idx) {
          double v = 0.0;
#pragma unroll
          for (int k = 0; k < N; ++k) v = (k == idx) ? call[k] : v;
          return v;
        };
        const double wr = ws * rs_r;
        const int nr = rt->n;
        for (int q = 0; q < nr; ++q) {
          const int nl = rt->n_lhs[q], nrh = rt->n_rhs[q];
          // net stoichiometry of this lane's species in reaction q (products - educts)
          int mult = 0;
          for (int a = 0; a < nl; ++a) mult -= (rt->lhs[q][a] == r) ? 1 : 0;
          for (int a = 0; a < nrh; ++a) mult += (rt->rhs[q][a] == r) ? 1 : 0;
          for (int side = 0; side < 2; ++side) {
            const int n = side == 0 ? nl : nrh;
            const int32_t* idx = side == 0 ? rt->lhs[q] : rt->rhs[q];
            const double kk = side == 0 ? rt->kf[q] : rt->kr[q];
            if (kk == 0.0) continue;
            double pre = kk;
            for (int a = 0; a < n; ++a) pre *= inv0;
            double prod = pre;
            for (int a = 0; a < n; ++a) prod *= pick(idx[a]);
            const double w = (side == 0 ? 1.0 : -1.0) * mult * wr;      // forward minus backward
            rhs = __builtin_fma(w, prod, rhs);
#pragma unroll
            for (int j = 0; j < N; ++j) {
              double dp = MPB ? prod * n * A.vol[j] * inv0 : 0.0;
              for (int a = 0; a < n; ++a) {
                if (idx[a] != j) continue;
                double rest = pre;
                for (int b2 = 0; b2 < n; ++b2)
                  if (b2 != a) rest *= pick(idx[b2]);
                dp += rest;
              }
              Dr[j] = __builtin_fma(-w, dp, Dr[j]);
            }
          }
        }
      }
      if (wall && A.n_wk > 0) {   // implicit surface kinetics, see fill_row
        for (int q = 0; q < A.n_wk; ++q) {
          const int sp = A.wk_species[q];
          double nu = 0.0;
#pragma unroll
          for (int k = 0; k < N; ++k) nu = (k == r) ? A.wk_nu[q][k] : nu;
          const double a = nu * wk[q] * fl_r;
          const double cs = sp >= 0 ? c[sp * ldx] : 1.0;
          const double al = A.wk_alpha[q], den = 1.0 / (1.0 + A.wk_sat[q] * cs);
          const double E = al != 0.0 ? exp(al * (phiM - p0)) : 1.0;
          const double g = cs * den * E, dg = den * den * E;
          rhs += a * g;
#pragma unroll
          for (int j = 0; j < N; ++j) Dr[j] -= (j == sp) ? a * dg : 0.0;
          if (al != 0.0) Dr[N] += a * al * g;
        }
      }
      Xr[2 * NB] = rhs;
    } else {
      if (bulk) {
        Xr[2 * NB] = -(p0 - phiB);
        Dr[N] = 1.0;
      } else if (wall) {
        if (A.wall_bc == 0) {
          Xr[2 * NB] = -(p0 - phiM);
          Dr[N] = 1.0;
        } else {
          Xr[2 * NB] = -(wep * (pp - p0) + A.stern * (phiM - A.phi_pzc - p0));
          Dr[N] = -wep - A.stern;
          Xr[NB + N] = wep;
        }
      } else {
        Xr[2 * NB] = -(wep * (pp - p0) - wem * (p0 - pm) + vi * rho);
#pragma unroll
        for (int k = 0; k < N; ++k) Dr[k] = vi * A.peq[k];
        Dr[N] = -(wep + wem);
        Xr[N] = wem;
        Xr[NB + N] = wep;
      }
    }
}

template <int NB, int MODE>
__global__ __launch_bounds__(1024, 4) void newton_team_kernel(const NewtonArgs G) {
  __shared__ NewtonArgs sA;        // scalar parameters in LDS, pointers stay kernel arguments (see newton_pair_kernel)
  if (threadIdx.x == 0) sA = G;
  __syncthreads();
  const NewtonArgs& A = sA;
  using TL = TeamLayout<NB>;
  constexpr int N = NB - 1, NC = TL::NC, NCP = TL::NCP, TPW = TL::TPW;
  constexpr bool MPB = MODE >= 1, REACT = MODE == 2;
  extern __shared__ double newton_lds[];
  __shared__ double red[2][16];
  const int tid = threadIdx.x, T = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
  const int tw = lane / NB;                       // team inside the wave; tw == TPW: dummy team
  const int r0_ = lane - tw * NB;                 // row of the block = equation this lane owns
  const bool real_team = tw < TPW;
  const int team0_ = wave * TPW + (real_team ? tw : 0);
  const int nteams = nwaves * TPW;
  double* strip = newton_lds + (size_t)(wave * (TPW + 1) + tw) * TL::SLT;
  double* tile = strip + TL::SL;                  // NB x NC doubles: one neighbour block row
  const int nx = A.nx, ldx = A.ldx;
  double* rowbuf = G.work + (size_t)blockIdx.x * G.work_stride;
  auto REC = [&](int row, int rr) { return rowbuf + ((size_t)row * NB + rr) * NCP; };
  const bool spec = r0_ < N;
  const int rs0_ = spec ? r0_ : 0;                // clamped species index for loads
  for (int64_t b = blockIdx.x; b < G.B; b += gridDim.x) {
    if (G.lane_mask && !G.lane_mask[b]) continue;       // frozen lane (block-uniform: no thread reaches a barrier)
    double* c = G.c + (size_t)b * N * ldx;
    double* co = G.c_old + (size_t)b * N * ldx;
    double* phi = G.phi + (size_t)b * ldx;
    const double* cb = G.cbulk + (size_t)b * N;
    const double* wk = G.wk_k + (size_t)b * PNP_MAX_WALL_REACTIONS;
    const double phiM = G.pb[b * 4 + 0], phiB = G.pb[b * 4 + 1];
    int total_it = 0, st = PNP_STATUS_OK;
    for (int step = 0; step < A.nsteps; ++step) {
      if (!A.ext_old)
        for (int e = tid; e < N * ldx; e += T) co[e] = c[e];
      __syncthreads();
      bool conv = false;
      double upd_prev = INFINITY;       // scaled update of the previous full (undamped) iteration
      int it = 1;
      for (; it <= A.maxit; ++it) {
        // lane coordinates made opaque once per iteration: addresses derived from them are recomputed where they are used
        // instead of being precomputed before the loop and reloaded from spill slots (see newton_pair_kernel)
        // ---- assembly + normalisation, one block row per team -------------------------------------------------
        for (int row0 = 0; row0 < nx; row0 += nteams) {
          // lane coordinates made opaque once per pass: everything derived from them (addresses, the lane's species constants,
          // which are read from the parameter copy in LDS) is recomputed per pass instead of being kept in -- spilled --
          // registers across the whole Newton loop (see newton_pair_kernel)
          int r = r0_, team = team0_;
          const NewtonArgs* Ap = &sA;            // ... and so is the pointer to the parameter copy: its loads stay inside the pass
          asm volatile("" : "+v"(r), "+v"(team), "+v"(Ap));
          const NewtonArgs& A = *Ap;
          const int i = min(row0 + team, nx - 1);
          const bool valid = real_team && row0 + team < nx;
          double Dr[NB], Xr[NC];
          team_assemble_row<NB, MODE>(A, G, c, co, phi, cb, wk, phiM, phiB, b, i, r, spec, strip, Dr, Xr);
          int myk;
          team_solve<NB, (MODE != 0)>(Dr, Xr, strip, r, myk);
          if (valid) {
            double* o = REC(i, myk);
#pragma unroll
            for (int j = 0; j < NC; ++j) o[j] = Xr[j];
          }
          team_sync();
        }
        __syncthreads();
        // ---- block cyclic reduction in place ---------------------------------------------------------------------
        int s = 1;
        for (; s < nx; s <<= 1) {
          const int nact = (nx - (2 * s - 1) + 2 * s - 1) / (2 * s);      // rows 2s-1 + t*2s < nx
          for (int t0 = 0; t0 < nact; t0 += nteams) {
            int r = r0_, team = team0_;
            asm volatile("" : "+v"(r), "+v"(team));
            const int t = t0 + team;
            const bool valid = real_team && t < nact;
            const int row = valid ? 2 * s - 1 + t * 2 * s : 2 * s - 1;
            const bool hm = row - s >= 0, hp = row + s < nx;
            const double* own = REC(row, r);
            double Lt[NB], Ut[NB], Dr[NB], Xr[NC];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
              Lt[j] = own[j];
              Ut[j] = own[NB + j];
              Dr[j] = (j == r) ? 1.0 : 0.0;
              Xr[j] = 0.0;
              Xr[NB + j] = 0.0;
            }
            Xr[2 * NB] = own[2 * NB];
            if constexpr (!TL::STAGE) {
              if (hm) {
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                  const double* nb_ = REC(row - s, q);
                  const double lq = Lt[q];
#pragma unroll
                  for (int j = 0; j < NB; ++j) {
                    Dr[j] = __builtin_fma(-lq, nb_[NB + j], Dr[j]);       // - Lt Ut[-s]
                    Xr[j] = __builtin_fma(-lq, nb_[j], Xr[j]);            // - Lt Lt[-s]
                  }
                  Xr[2 * NB] = __builtin_fma(-lq, nb_[2 * NB], Xr[2 * NB]);
                }
              }
              if (hp) {
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                  const double* nb_ = REC(row + s, q);
                  const double uq = Ut[q];
#pragma unroll
                  for (int j = 0; j < NB; ++j) {
                    Dr[j] = __builtin_fma(-uq, nb_[j], Dr[j]);            // - Ut Lt[+s]
                    Xr[NB + j] = __builtin_fma(-uq, nb_[NB + j], Xr[NB + j]);   // - Ut Ut[+s]
                  }
                  Xr[2 * NB] = __builtin_fma(-uq, nb_[2 * NB], Xr[2 * NB]);
                }
              }
            } else {
              // neighbour block rows travel through the team's LDS tile: every lane fetches ITS row of the neighbour in one batch
              // of wide loads (one memory round trip per side instead of one per row), then the products read the tile
              // (broadcast within the team)
  #pragma unroll
              for (int side = 0; side < 2; ++side) {
                const bool have = side == 0 ? hm : hp;
                const double* src = REC(have ? (side == 0 ? row - s : row + s) : row, r);
                double tmp[NC];
  #pragma unroll
                for (int j = 0; j < NC; ++j) tmp[j] = src[j];
  #pragma unroll
                for (int j = 0; j < NC; ++j) tile[r * NC + j] = tmp[j];
                team_sync();
                if (have) {
  #pragma unroll
                  for (int q = 0; q < NB; ++q) {
                    const double* nb_ = tile + q * NC;
                    const double f = side == 0 ? Lt[q] : Ut[q];
  #pragma unroll
                    for (int j = 0; j < NB; ++j) {
                      if (side == 0) {
                        Dr[j] = __builtin_fma(-f, nb_[NB + j], Dr[j]);             // - Lt Ut[-s]
                        Xr[j] = __builtin_fma(-f, nb_[j], Xr[j]);                  // - Lt Lt[-s]
                      } else {
                        Dr[j] = __builtin_fma(-f, nb_[j], Dr[j]);                  // - Ut Lt[+s]
                        Xr[NB + j] = __builtin_fma(-f, nb_[NB + j], Xr[NB + j]);   // - Ut Ut[+s]
                      }
                    }
                    Xr[2 * NB] = __builtin_fma(-f, nb_[2 * NB], Xr[2 * NB]);
                  }
                }
                team_sync();
              }
            }
            int myk;
            team_solve<NB, false>(Dr, Xr, strip, r, myk);
            if (valid) {
              double* o = REC(row, r);
#pragma unroll
              for (int j = 0; j < NC; ++j) o[j] = Xr[j];
            }
          }
          __syncthreads();
        }
        for (s >>= 1; s >= 1; s >>= 1) {
          const int nact = (nx - (s - 1) + 2 * s - 1) / (2 * s);          // rows s-1 + t*2s < nx
          for (int t0 = 0; t0 < nact; t0 += nteams) {
            const int r = r0_, team = team0_;
            const int t = t0 + team;
            if (real_team && t < nact) {
              const int row = s - 1 + t * 2 * s;
              const double* own = REC(row, r);
              double x = own[2 * NB];
              if (row - s >= 0) {
#pragma unroll
                for (int q = 0; q < NB; ++q) x = __builtin_fma(-own[q], REC(row - s, q)[2 * NB], x);
              }
              if (row + s < nx) {
#pragma unroll
                for (int q = 0; q < NB; ++q) x = __builtin_fma(-own[NB + q], REC(row + s, q)[2 * NB], x);
              }
              REC(row, r)[2 * NB] = x;
            }
          }
          __syncthreads();
        }
        // ---- damping, update, convergence: one thread per grid point (oracle/pnp_physical.py: newton_step) ---------
        double mphi = 0.0, upd = 0.0;
        for (int row = tid; row < nx; row += T) {
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const double du = REC(row, k)[2 * NB];
            const double ck = c[k * ldx + row];
            upd = fmax(upd, fabs(du) / (fabs(ck) + fabs(cb[k]) + 1e-300));
            if (!(du == du)) upd = INFINITY;
          }
          const double a = fabs(REC(row, N)[2 * NB]);
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
        for (int w = 0; w < nwaves; ++w) {
          mphi = fmax(mphi, red[0][w]);
          upd = fmax(upd, red[1][w]);
        }
        double lam = 1.0;
        if (A.dphi_max > 0.0 && mphi > A.dphi_max) lam = A.dphi_max / mphi;
        for (int row = tid; row < nx; row += T) {
          double cn[N], cc_[N];
#pragma unroll
          for (int k = 0; k < N; ++k) {
            cc_[k] = c[k * ldx + row];
            const double t_ = __builtin_fma(lam, REC(row, k)[2 * NB], cc_[k]);
            const double lo = 0.1 * cc_[k];
            cn[k] = t_ < lo ? lo : t_;
          }
          if constexpr (MPB) {
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
          phi[row] = __builtin_fma(lam, REC(row, N)[2 * NB], phi[row]);
        }
        __syncthreads();
        if (lam == 1.0) {
          // strict: the update itself is below tol.  With A.estimate the state is accepted as soon as the quadratic error
          // estimate of the state just computed, upd^2/upd_prev (two consecutive contracting full steps), is below tol --
          // the iteration that would only confirm it is skipped (oracle/pnp_physical.py: newton_step(estimate=True))
          if (upd < A.tol || (A.estimate && upd_prev < INFINITY && upd < 0.1 * upd_prev && upd * (upd / upd_prev) < A.tol) ||
              newton_at_rounding_floor(upd, upd_prev, A.tol)) {
            conv = true;
            break;
          }
          upd_prev = upd;
        } else {
          upd_prev = INFINITY;
        }
      }
      total_it += conv ? it : A.maxit + 1;
      if (!conv) st = PNP_STATUS_MAXIT;
    }
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
      for (int w = 0; w < nwaves; ++w) bad = fmax(bad, red[0][w]);
      G.status[b] = bad > 0.0 ? PNP_STATUS_NAN : st;
      G.iters[b] = total_it;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Sweep kernel for LARGE BATCHES of large blocks: ONE TEAM (NB lanes) PER OPERATING POINT, block Thomas along the grid.
// Cyclic reduction spends a workgroup per operating point and streams every block row ~6 times per Newton iteration through
// device memory (N = 8, nx = 4096: ~60 MB per operating point and iteration -- with thousands of lanes in flight that traffic,
// not arithmetic, bounds the lane-team kernel above).  When the batch alone fills the chip the parallelism inside one grid is
// not needed: a team walks its grid once forward (assemble row i, D'_i = D_i - L_i Ut_{i-1}, [Ut_i | rt_i] = D'_i^-1 [U_i | r_i -
// L_i rt_{i-1}], one record of NB x (NB+1) doubles per row written to device memory) and once backward (x_i = rt_i - Ut_i
// x_{i+1}): O(nx) work instead of O(nx log nx), each record written and read once (~6 MB per operating point and iteration
// for N = 8, nx = 4096), no workgroup barrier anywhere -- the 64/NB teams of a wave do not interact and a wave is the workgroup.
// Same assembly (team_assemble_row), damping, update and stopping rule as the lane-team kernel.
// ------------------------------------------------------------------------------------------------
template <int NB>
struct SweepLayout {
  static constexpr int NW = (NB + 1 + 1) / 2 * 2;     // record of one block-row lane: Ut row (NB) + rt, padded to even
  static constexpr int TPW = 64 / NB;
  static constexpr int SL = ((NB + NB + NB + 1) + 1) / 2 * 2;      // strip: NB scalars + pivot row (NB of D, NB + 1 of X)
  static constexpr int TILE = NB * (NB + 1) + (NB * (NB + 1)) % 2; // [Ut | rt] of the previous row
  static constexpr int BASE = SL + TILE + NB + NB % 2;             // + the solution vector of the row below (back-substitution)
  // The teams of a wave read their tiles / strips at the SAME offset in the same instruction (one broadcast address per team).  LDS
  // banks repeat every 16 doubles: with a team pitch that is a multiple of 16 (N = 8: 128 doubles) all teams hit the same banks and
  // every such read is served team after team; a pitch of 2 (mod 16) doubles gives 8 teams disjoint banks even for 16-byte reads.
  static constexpr int PER_TEAM = BASE + ((2 - BASE % 16) + 16) % 16;
};
size_t newton_sweep_doubles(int nb, int nx) { return (size_t)nx * nb * ((nb + 2) / 2 * 2); }

template <int NB, int MODE>
__global__ __launch_bounds__(64, 2) void newton_sweep_kernel(const NewtonArgs G) {
  __shared__ NewtonArgs sA;
  if (threadIdx.x == 0) sA = G;
  __syncthreads();
  using SL_ = SweepLayout<NB>;
  constexpr int N = NB - 1, NW = SL_::NW, TPW = SL_::TPW, NY = NB + 1;
  constexpr bool MPB = MODE >= 1;
  __shared__ double sweep_lds[(TPW + 1) * SL_::PER_TEAM];
  const int lane = threadIdx.x;
  const int tw = lane / NB;
  const int r0_ = lane - tw * NB;
  const bool real_team = tw < TPW;
  double* strip = sweep_lds + (size_t)tw * SL_::PER_TEAM;
  double* tile = strip + SL_::SL;
  double* xs = tile + SL_::TILE;
  const int64_t gteam = (int64_t)blockIdx.x * TPW + (real_team ? tw : 0);
  const int64_t nteams = (int64_t)gridDim.x * TPW;
  double* W = G.sweep + (size_t)gteam * G.sweep_stride;
  const bool spec = r0_ < N;
  // CONTROL FLOW IS WAVE-UNIFORM.  The kernel is an iteration engine: every pass of the loop below is ONE Newton iteration of
  // whatever operating point each team currently holds; a team that finishes its point (all timesteps converged or out of
  // iterations) takes its next one (gteam, gteam + nteams, ...) at the top of the next pass, so the teams of a wave never wait
  // for each other's iteration counts.  A team without work (batch exhausted, masked lane, the leftover lanes of the wave) runs
  // along on valid data with its state stores switched off, and the loop ends on a wave-wide vote.  (Letting teams diverge --
  // skip rounds, leave a Newton loop early -- produced wild addresses in the instances that spill registers: values spilled
  // under a partial exec mask came back undefined for the other lanes.)
  const int nx = sA.nx, ldx = sA.ldx;
  int64_t bnext = gteam;          // the team's next operating point
  int64_t b = 0;                  // ... and its current one (valid memory even while the team has no work)
  bool have = false, fresh = false;
  int step = 0, it = 0, total_it = 0, st = PNP_STATUS_OK;
  double upd_prev = INFINITY;
#ifdef PNP_SWEEP_STAMPS
  int stamp_code = 0, stamp_cycles = 0;
#endif
  for (;;) {
    if (!have && real_team && bnext < G.B) {
      b = bnext;
      bnext += nteams;
      have = !(G.lane_mask && !G.lane_mask[b]);
      fresh = true;
      step = 0;
      total_it = 0;
      st = PNP_STATUS_OK;
    }
    if (__ballot(have) == 0ull) {
      if (__ballot(real_team && bnext < G.B) == 0ull) break;
      continue;                   // (only masked lanes were drawn: draw again)
    }
    const NewtonArgs& A = sA;
    double* c = G.c + (size_t)b * N * ldx;
    double* co = G.c_old + (size_t)b * N * ldx;
    double* phi = G.phi + (size_t)b * ldx;
    const double* cb = G.cbulk + (size_t)b * N;
    const double* wk = G.wk_k + (size_t)b * PNP_MAX_WALL_REACTIONS;
    const double phiM = G.pb[b * 4 + 0], phiB = G.pb[b * 4 + 1];
    // ---- start of a timestep: previous time level ------------------------------------------------------------------------
    if (__ballot(have && fresh) != 0ull) {
      for (int e = r0_; e < N * ldx; e += NB) {
        const double v = c[e];
        if (have && fresh && !G.ext_old) co[e] = v;
      }
      team_sync();
    }
    if (fresh) {
      it = 0;
      upd_prev = INFINITY;
      fresh = false;
    }
    it += 1;
#ifdef PNP_SWEEP_STAMPS
    const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
    // ---- forward: assemble, eliminate the sub-diagonal block with the previous row's record, solve, record ------------------
    RowCarry carry;
    for (int i = 0; i < nx; ++i) {
      int r = r0_;
      const NewtonArgs* Ap = &sA;
      asm volatile("" : "+v"(r), "+v"(Ap));      // see newton_team_kernel: keeps per-lane constants out of spill slots
      const NewtonArgs& A = *Ap;
      double Dr[NB], Xr[2 * NB + 1];
      team_assemble_row<NB, MODE, (NB >= 8)>(A, G, c, co, phi, cb, wk, phiM, phiB, b, i, r, spec, strip, Dr, Xr, &carry, i > 0);   // (measured: +4 % at N = 8, -6 % at N = 6)
      double Y[NY];
#pragma unroll
      for (int j = 0; j < NB; ++j) Y[j] = Xr[NB + j];
      Y[NB] = Xr[2 * NB];
      if (i > 0) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const double lq = Xr[q];
          const double* prev = tile + q * NY;
#pragma unroll
          for (int j = 0; j < NB; ++j) Dr[j] = __builtin_fma(-lq, prev[j], Dr[j]);
          Y[NB] = __builtin_fma(-lq, prev[NB], Y[NB]);
        }
      }
      team_sync();
      int myk;
      team_solve_n<NB, NY, (MODE != 0)>(Dr, Y, strip, r, myk);
      myk = myk < 0 ? r : myk;      // (a NaN block leaves a lane without a pivot; the lane's state is flagged at the end)
      double* rec = W + ((size_t)i * NB + myk) * NW;
#pragma unroll
      for (int j = 0; j < NY; ++j) {
        tile[myk * NY + j] = Y[j];
#ifndef PNP_SWEEP_NOSTORE
        if (real_team) rec[j] = Y[j];        // (the leftover lanes of the wave have no records of their own)
#endif
      }
      team_sync();
    }
#ifdef PNP_SWEEP_STAMPS
    const unsigned long long ts1 = __builtin_readcyclecounter();
#endif
    // ---- backward: x_i = rt_i - Ut_i x_{i+1}; the update norms on the way -------------------------------------------------
    double mphi = 0.0, upd = 0.0;
    {
      const int r = r0_;
      for (int i = nx - 1; i >= 0; --i) {
        double* rec = W + ((size_t)i * NB + r) * NW;
        double x = rec[NB];
        if (i < nx - 1) {
#pragma unroll
          for (int j = 0; j < NB; ++j) x = __builtin_fma(-rec[j], xs[j], x);
        }
        team_sync();
        xs[r] = x;
        if (real_team) rec[NB] = x;
        team_sync();
        const double ck = c[(spec ? r : 0) * ldx + i];
        const double rel = fabs(x) / (fabs(ck) + fabs(cb[spec ? r : 0]) + 1e-300);
        const double a = fabs(x);
        if (spec) {
          upd = fmax(upd, rel);
          if (!(x == x)) upd = INFINITY;
        } else {
          mphi = fmax(mphi, a);
          if (!(a == a)) mphi = INFINITY;
        }
      }
    }
    auto team_max = [&](double v) {
      strip[r0_] = v;
      team_sync();
      double m = 0.0;
#pragma unroll
      for (int j = 0; j < NB; ++j) m = fmax(m, strip[j]);
      team_sync();
      return m;
    };
    mphi = team_max(mphi);
    upd = team_max(upd);
    upd = fmax(upd, mphi * A.vt_inv);
    double lam = 1.0;
    if (A.dphi_max > 0.0 && mphi > A.dphi_max) lam = A.dphi_max / mphi;
#ifdef PNP_SWEEP_STAMPS
    const unsigned long long ts2 = __builtin_readcyclecounter();
#endif
    // ---- damping, clips, update (oracle/pnp_physical.py: newton_step), row by row -----------------------------------------
    for (int i = 0; i < nx; ++i) {
      const int r = r0_;
      const double du = W[((size_t)i * NB + r) * NW + NB];
      double cc_ = 0.0, cn = 0.0;
      if (spec) {
        cc_ = c[r * ldx + i];
        const double t_ = __builtin_fma(lam, du, cc_);
        const double lo = 0.1 * cc_;
        cn = t_ < lo ? lo : t_;
      }
      if constexpr (MPB) {
        strip[r] = cc_;
        xs[r] = cn;
        team_sync();
        double f_old = 0.0, f_new = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
          f_old = __builtin_fma(A.vol[k], strip[k], f_old);
          f_new = __builtin_fma(A.vol[k], xs[k], f_new);
        }
        team_sync();
        const double free_ = 1.0 - f_old;
        const double target = fmax(0.1 * free_, 1e-12);
        if ((1.0 - f_new) < target) {
          const double theta = (free_ - target) / (f_new - f_old);
          cn = __builtin_fma(theta, cn - cc_, cc_);
        }
      }
      if (have) {
        if (spec) c[r * ldx + i] = cn;
        else phi[i] = __builtin_fma(lam, du, phi[i]);
      }
    }
    team_sync();
#ifdef PNP_SWEEP_STAMPS
    {   // diagnosis build: per-mille shares of the forward and backward passes of the LAST iteration, cycles per row in the rest
      const unsigned long long ts3 = __builtin_readcyclecounter();
      const double tot = (double)(ts3 - ts0);
      stamp_code = (int)(1000.0 * (double)(ts1 - ts0) / tot) + 1000 * (int)(1000.0 * (double)(ts2 - ts1) / tot);
      stamp_cycles = (int)(tot / nx);
    }
#endif
    // ---- bookkeeping of the team's operating point: iteration -> timestep -> finished -------------------------------------
    bool finished = false;
    if (have) {
      bool accept = false;
      if (lam == 1.0) {
        accept = upd < A.tol || (A.estimate && upd_prev < INFINITY && upd < 0.1 * upd_prev && upd * (upd / upd_prev) < A.tol) ||
              newton_at_rounding_floor(upd, upd_prev, A.tol);
        upd_prev = upd;
      } else {
        upd_prev = INFINITY;
      }
      if (accept || it >= A.maxit) {
        total_it += accept ? it : A.maxit + 1;
        if (!accept) st = PNP_STATUS_MAXIT;
        step += 1;
        fresh = true;
        finished = step >= A.nsteps;
      }
    }
    if (__ballot(finished) != 0ull) {
      double bad = 0.0;
      for (int e = r0_; e < nx; e += NB) {
        double sacc = phi[e];
#pragma unroll
        for (int k = 0; k < N; ++k) sacc += c[k * ldx + e];
        if (!(fabs(sacc) < INFINITY)) bad = 1.0;
      }
      strip[r0_] = bad;
      team_sync();
#pragma unroll
      for (int j = 0; j < NB; ++j) bad = fmax(bad, strip[j]);
      if (finished && r0_ == 0) {
        G.status[b] = bad > 0.0 ? PNP_STATUS_NAN : st;
        G.iters[b] = total_it;
#ifdef PNP_SWEEP_STAMPS
        G.iters[b] = stamp_code;
        G.status[b] = stamp_cycles;
#endif
      }
      team_sync();
    }
    if (finished) have = false;
  }
}

// ------------------------------------------------------------------------------------------------
// Two-sided sweep ("burn at both ends", twisted block factorisation): TWO teams per operating point.  Team 0 eliminates upwards from
// the wall (rows 0 .. m-1: D'_i = D_i - L_i Ut_{i-1}, [Ut_i | rt_i] = D'_i^-1 [U_i | r_i - L_i rt_{i-1}]), team 1 downwards from the
// bulk (rows nx-1 .. m+1, the mirror image: D''_i = D_i - U_i Lt_{i+1}, [Lt_i | rt_i] = D''_i^-1 [L_i | r_i - U_i rt_{i+1}]); the
// middle row m sees both: (D_m - L_m Ut_{m-1} - U_m Lt_{m+1}) x_m = r_m - L_m rt_{m-1} - U_m rt_{m+1}; then both teams substitute
// outwards (x_i = rt_i - Ut_i x_{i+1} below m, x_i = rt_i - Lt_i x_{i-1} above).  The arithmetic is that of the one-sided sweep (no
// fill-in, no interface system beyond the one middle block), the dependent chain of a Newton iteration is half as long and a batch
// occupies twice the waves: for the batch sizes where the one-sided sweep leaves the SIMDs with one or two waves (B ~ 4-16 k at
// N = 8) that is where the time goes (66 % of a wave's life parked).  Both teams of a pair sit in the same wave
// and run the same instruction stream; which off-diagonal block is "behind" and which "ahead" is a per-lane select.  Everything
// else -- assembly, damping, update, stopping rule, wave-uniform iteration engine -- is the one-sided kernel's.
// ------------------------------------------------------------------------------------------------
template <int NB, int MODE>
__global__ __launch_bounds__(64, 2) void newton_sweep2_kernel(const NewtonArgs G) {
  __shared__ NewtonArgs sA;
  if (threadIdx.x == 0) sA = G;
  __syncthreads();
  using SL_ = SweepLayout<NB>;
  constexpr int N = NB - 1, NW = SL_::NW, TPW = SL_::TPW, PPW = TPW / 2, NY = NB + 1;
  constexpr bool MPB = MODE >= 1;
  __shared__ double sweep_lds[(TPW + 2) * SL_::PER_TEAM];
  const int lane = threadIdx.x;
  const int tw = lane / NB;
  const int r0_ = lane - tw * NB;
  const bool real_team = tw < 2 * PPW;
  const int side = tw & 1;            // 0: from the wall upwards, 1: from the bulk downwards
  double* strip = sweep_lds + (size_t)tw * SL_::PER_TEAM;
  double* tile = strip + SL_::SL;
  double* xs = tile + SL_::TILE;
  const int poff = side ? -SL_::PER_TEAM : SL_::PER_TEAM;      // the partner team's LDS block
  const int64_t gteam = (int64_t)blockIdx.x * PPW + (real_team ? (tw >> 1) : 0);
  const int64_t nteams = (int64_t)gridDim.x * PPW;
  double* W = G.sweep + (size_t)gteam * G.sweep_stride;
  const bool spec = r0_ < N;
  const int nx = sA.nx, ldx = sA.ldx;
  const int m = nx >> 1;              // middle row; team 0 owns rows [0, m), team 1 rows (m, nx)
  const int n_up = nx - 1 - m;        // rows of team 1 (m or m - 1)
  int64_t bnext = gteam;
  int64_t b = 0;
  bool have = false, fresh = false;
  int step = 0, it = 0, total_it = 0, st = PNP_STATUS_OK;
  double upd_prev = INFINITY;
  for (;;) {      // CONTROL FLOW IS WAVE-UNIFORM, see newton_sweep_kernel
    if (!have && real_team && bnext < G.B) {
      b = bnext;
      bnext += nteams;
      have = !(G.lane_mask && !G.lane_mask[b]);
      fresh = true;
      step = 0;
      total_it = 0;
      st = PNP_STATUS_OK;
    }
    if (__ballot(have) == 0ull) {
      if (__ballot(real_team && bnext < G.B) == 0ull) break;
      continue;
    }
    const NewtonArgs& A = sA;
    double* c = G.c + (size_t)b * N * ldx;
    double* co = G.c_old + (size_t)b * N * ldx;
    double* phi = G.phi + (size_t)b * ldx;
    const double* cb = G.cbulk + (size_t)b * N;
    const double* wk = G.wk_k + (size_t)b * PNP_MAX_WALL_REACTIONS;
    const double phiM = G.pb[b * 4 + 0], phiB = G.pb[b * 4 + 1];
    if (__ballot(have && fresh) != 0ull) {      // previous time level: each team of the pair copies every other stripe
      for (int e = r0_ + side * NB; e < N * ldx; e += 2 * NB) {
        const double v = c[e];
        if (have && fresh && !G.ext_old) co[e] = v;
      }
      team_sync();
    }
    if (fresh) {
      it = 0;
      upd_prev = INFINITY;
      fresh = false;
    }
    it += 1;
    // ---- elimination from both ends ---------------------------------------------------------------------------------------------
    for (int j = 0; j < m; ++j) {
      int r = r0_;
      const NewtonArgs* Ap = &sA;
      asm volatile("" : "+v"(r), "+v"(Ap));
      const NewtonArgs& A = *Ap;
      const bool act = side == 0 || j < n_up;                        // (team 1 has one row less when nx is even: it runs along)
      const int i = side == 0 ? j : nx - 1 - (act ? j : n_up - 1);
      double Dr[NB], Xr[2 * NB + 1];
      team_assemble_row<NB, MODE, false>(A, G, c, co, phi, cb, wk, phiM, phiB, b, i, r, spec, strip, Dr, Xr);
      double Y[NY];
#pragma unroll
      for (int q = 0; q < NB; ++q) Y[q] = side ? Xr[q] : Xr[NB + q];      // the block towards the middle
      Y[NB] = Xr[2 * NB];
      if (j > 0) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const double lq = side ? Xr[NB + q] : Xr[q];                    // the block towards the end already eliminated
          const double* prev = tile + q * NY;
#pragma unroll
          for (int jj = 0; jj < NB; ++jj) Dr[jj] = __builtin_fma(-lq, prev[jj], Dr[jj]);
          Y[NB] = __builtin_fma(-lq, prev[NB], Y[NB]);
        }
      }
      team_sync();
      int myk;
      team_solve_n<NB, NY, (MODE != 0)>(Dr, Y, strip, r, myk);
      myk = myk < 0 ? r : myk;
      double* rec = W + ((size_t)i * NB + myk) * NW;
#pragma unroll
      for (int jj = 0; jj < NY; ++jj) {
        if (act) tile[myk * NY + jj] = Y[jj];
        if (real_team && act) rec[jj] = Y[jj];
      }
      team_sync();
    }
    // ---- the middle row: both neighbours eliminated; both teams solve it (same operands, same bits) -----------------------------
    double mphi = 0.0, upd = 0.0;
    {
      int r = r0_;
      const NewtonArgs* Ap = &sA;
      asm volatile("" : "+v"(r), "+v"(Ap));
      const NewtonArgs& A = *Ap;
      double Dr[NB], Xr[2 * NB + 1];
      team_assemble_row<NB, MODE, false>(A, G, c, co, phi, cb, wk, phiM, phiB, b, m, r, spec, strip, Dr, Xr);
      const double* t_lo = side ? tile + poff : tile;       // [Ut | rt] of row m - 1
      const double* t_hi = side ? tile : tile + poff;       // [Lt | rt] of row m + 1
      double Y1[1] = {Xr[2 * NB]};
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const double lq = Xr[q], uq = Xr[NB + q];
        const double* lo = t_lo + q * NY;
        const double* hi = t_hi + q * NY;
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) Dr[jj] = __builtin_fma(-uq, hi[jj], __builtin_fma(-lq, lo[jj], Dr[jj]));
        Y1[0] = __builtin_fma(-uq, hi[NB], __builtin_fma(-lq, lo[NB], Y1[0]));
      }
      team_sync();
      int myk;
      team_solve_n<NB, 1, (MODE != 0)>(Dr, Y1, strip, r, myk);
      myk = myk < 0 ? r : myk;
      team_sync();
      xs[myk] = Y1[0];
      if (real_team && side == 0) W[((size_t)m * NB + myk) * NW + NB] = Y1[0];
      team_sync();
      const double x = xs[r];
      const double ck = c[(spec ? r : 0) * ldx + m];
      const double rel = fabs(x) / (fabs(ck) + fabs(cb[spec ? r : 0]) + 1e-300);
      const double a = fabs(x);
      if (spec) {
        upd = fmax(upd, rel);
        if (!(x == x)) upd = INFINITY;
      } else {
        mphi = fmax(mphi, a);
        if (!(a == a)) mphi = INFINITY;
      }
    }
    // ---- substitution outwards from the middle ------------------------------------------------------------------------------------
    {
      const int r = r0_;
      for (int j = m - 1; j >= 0; --j) {
        const bool act = side == 0 || j < n_up;
        const int i = side == 0 ? j : nx - 1 - (act ? j : 0);
        double* rec = W + ((size_t)i * NB + r) * NW;
        double x = rec[NB];
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) x = __builtin_fma(-rec[jj], xs[jj], x);
        team_sync();
        if (act) xs[r] = x;
        if (real_team && act) rec[NB] = x;
        team_sync();
        const double ck = c[(spec ? r : 0) * ldx + i];
        const double rel = fabs(x) / (fabs(ck) + fabs(cb[spec ? r : 0]) + 1e-300);
        const double a = fabs(x);
        if (act) {
          if (spec) {
            upd = fmax(upd, rel);
            if (!(x == x)) upd = INFINITY;
          } else {
            mphi = fmax(mphi, a);
            if (!(a == a)) mphi = INFINITY;
          }
        }
      }
    }
    auto pair_max = [&](double v) {
      strip[r0_] = v;
      team_sync();
      double mm = 0.0;
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) mm = fmax(mm, fmax(strip[jj], strip[poff + jj]));
      team_sync();
      return mm;
    };
    mphi = pair_max(mphi);
    upd = pair_max(upd);
    upd = fmax(upd, mphi * A.vt_inv);
    double lam = 1.0;
    if (A.dphi_max > 0.0 && mphi > A.dphi_max) lam = A.dphi_max / mphi;
    // ---- damping, clips, update: team 0 rows [0, m), team 1 rows [m, nx) ------------------------------------------------------------
    for (int idx = 0; idx < nx - m; ++idx) {
      const int r = r0_;
      const bool act = side ? true : idx < m;
      const int i = side ? m + idx : (act ? idx : 0);
      const double du = W[((size_t)i * NB + r) * NW + NB];
      double cc_ = 0.0, cn = 0.0;
      if (spec) {
        cc_ = c[r * ldx + i];
        const double t_ = __builtin_fma(lam, du, cc_);
        const double lo = 0.1 * cc_;
        cn = t_ < lo ? lo : t_;
      }
      if constexpr (MPB) {
        strip[r] = cc_;
        xs[r] = cn;
        team_sync();
        double f_old = 0.0, f_new = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
          f_old = __builtin_fma(A.vol[k], strip[k], f_old);
          f_new = __builtin_fma(A.vol[k], xs[k], f_new);
        }
        team_sync();
        const double free_ = 1.0 - f_old;
        const double target = fmax(0.1 * free_, 1e-12);
        if ((1.0 - f_new) < target) {
          const double theta = (free_ - target) / (f_new - f_old);
          cn = __builtin_fma(theta, cn - cc_, cc_);
        }
      }
      if (have && act) {
        if (spec) c[r * ldx + i] = cn;
        else phi[i] = __builtin_fma(lam, du, phi[i]);
      }
    }
    team_sync();
    bool finished = false;
    if (have) {
      bool accept = false;
      if (lam == 1.0) {
        accept = upd < A.tol || (A.estimate && upd_prev < INFINITY && upd < 0.1 * upd_prev && upd * (upd / upd_prev) < A.tol) ||
              newton_at_rounding_floor(upd, upd_prev, A.tol);
        upd_prev = upd;
      } else {
        upd_prev = INFINITY;
      }
      if (accept || it >= A.maxit) {
        total_it += accept ? it : A.maxit + 1;
        if (!accept) st = PNP_STATUS_MAXIT;
        step += 1;
        fresh = true;
        finished = step >= A.nsteps;
      }
    }
    if (__ballot(finished) != 0ull) {
      double bad = 0.0;
      for (int e = r0_; e < nx; e += NB) {
        double sacc = phi[e];
#pragma unroll
        for (int k = 0; k < N; ++k) sacc += c[k * ldx + e];
        if (!(fabs(sacc) < INFINITY)) bad = 1.0;
      }
      strip[r0_] = bad;
      team_sync();
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) bad = fmax(bad, strip[jj]);
      if (finished && r0_ == 0 && side == 0) {
        G.status[b] = bad > 0.0 ? PNP_STATUS_NAN : st;
        G.iters[b] = total_it;
      }
      team_sync();
    }
    if (finished) have = false;
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

size_t newton_exchange_doubles(int nb, int nx) {   // the row buffer of one workgroup (cyclic reduction works in place)
  const size_t rs = (size_t)(nx + 15) / 16 * 16;
  return (size_t)(2 * nb * nb + nb) * rs;
}

bool newton_exchange_in_lds(int nb, int nx, const Options& opt) {
  if (opt.newton_exchange_global) return false;            // keeps the row-per-thread kernel's buffers in device memory (tests)
  if (nb >= 5) return false;               // large blocks: always device memory (global instead of flat instructions)
  return newton_exchange_doubles(nb, nx) * sizeof(double) <= kLdsBudget;
}

// pair kernel: nx <= 2*T threads, exchange buffer (2 NB^2 + NB) * TS doubles in LDS with a compile-time row stride
// TS = 256 or 512; instantiated for N <= 4 species (larger blocks do not fit the register file, they take the
// row-per-thread kernel)
int newton_pair_threads(int nb, int nx) {
  if (nb > 5) return 0;
  const int tmax = nb <= 4 ? 512 : 256;
  const int t = ((nx + 1) / 2 + 63) / 64 * 64;
  if (t > tmax) return 0;
  const int ts = t <= 64 ? 64 : (t <= 128 ? 128 : (t <= 256 ? 256 : 512));
  if ((size_t)(2 * nb * nb + nb) * ts * sizeof(double) > kLdsBudget) return 0;
  return t;
}

// compile-time row stride of the LDS exchange buffer: the smallest of 64/128/256/512 that holds the threads (a short grid
// then leaves LDS for several workgroups per CU)
int newton_pair_stride(int nb, int nx) {
  const int t = newton_pair_threads(nb, nx);
  return t == 0 ? 0 : (t <= 64 ? 64 : (t <= 128 ? 128 : (t <= 256 ? 256 : 512)));
}

template <int NB, int TS>
static hipError_t launch_pair(const NewtonArgs& a, int blocks, int tp, hipStream_t stream) {
  const size_t lds = (size_t)(2 * NB * NB + NB) * TS * sizeof(double);
  // variants: 0 point ions, 1 steric (MPB) drift, 2 steric drift + homogeneous reactions (point ions: zero volumes)
  if (a.rt) {
    (void)hipFuncSetAttribute((const void*)newton_pair_kernel<NB, TS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((newton_pair_kernel<NB, TS, 2>), dim3(blocks), dim3(tp), lds, stream, a);
  } else if (a.mpb) {
    (void)hipFuncSetAttribute((const void*)newton_pair_kernel<NB, TS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((newton_pair_kernel<NB, TS, 1>), dim3(blocks), dim3(tp), lds, stream, a);
  } else {
    (void)hipFuncSetAttribute((const void*)newton_pair_kernel<NB, TS, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((newton_pair_kernel<NB, TS, 0>), dim3(blocks), dim3(tp), lds, stream, a);
  }
  return hipGetLastError();
}

// lane-team kernel (N >= 5): row buffer [nx][NB][2 NB + 2] doubles per workgroup, 256 threads, 64/NB teams per wave
size_t newton_team_doubles(int nb, int nx) { return (size_t)nx * nb * (2 * nb + 2); }

template <int NB>
static hipError_t launch_team(const NewtonArgs& a, int blocks, hipStream_t stream) {
  using TL = TeamLayout<NB>;
  // threads per operating point: a small batch cannot fill the chip with 256-thread workgroups (4 resident per CU), so it
  // gets wider ones (shorter passes over the rows, same registers)
  int T = a.B * 4 <= 1024 ? 1024 : (a.B * 2 <= 1024 ? 512 : 256);
  if (a.opt) {
    const int v = a.opt->newton_team_threads;
    if (v == 256 || v == 512 || v == 1024) T = v;
  }
  size_t lds = (size_t)(T / 64) * (TL::TPW + 1) * TL::SLT * sizeof(double);
  while (lds > kLdsBudget && T > 64) {
    T /= 2;
    lds /= 2;
  }
  if (lds > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)newton_team_kernel<NB, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)newton_team_kernel<NB, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)newton_team_kernel<NB, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (a.rt) hipLaunchKernelGGL((newton_team_kernel<NB, 2>), dim3(blocks), dim3(T), lds, stream, a);
  else if (a.mpb) hipLaunchKernelGGL((newton_team_kernel<NB, 1>), dim3(blocks), dim3(T), lds, stream, a);
  else hipLaunchKernelGGL((newton_team_kernel<NB, 0>), dim3(blocks), dim3(T), lds, stream, a);
  return hipGetLastError();
}

// sweep kernel: one wave per workgroup, 64/NB operating points per wave
template <int NB>
static hipError_t launch_sweep(const NewtonArgs& a, hipStream_t stream) {
  constexpr int TPW = SweepLayout<NB>::TPW;
  int64_t blocks = (a.B + TPW - 1) / TPW;
  if (blocks > a.sweep_blocks) blocks = a.sweep_blocks;
  if (a.rt) hipLaunchKernelGGL((newton_sweep_kernel<NB, 2>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
  else if (a.mpb) hipLaunchKernelGGL((newton_sweep_kernel<NB, 1>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL((newton_sweep_kernel<NB, 0>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
  return hipGetLastError();
}

// two-sided sweep: one wave per workgroup, (64/NB)/2 operating points per wave
template <int NB>
static hipError_t launch_sweep2(const NewtonArgs& a, hipStream_t stream) {
  constexpr int PPW = SweepLayout<NB>::TPW / 2;
  int64_t blocks = (a.B + PPW - 1) / PPW;
  if (blocks > a.sweep_blocks) blocks = a.sweep_blocks;
  if (a.rt) hipLaunchKernelGGL((newton_sweep2_kernel<NB, 2>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
  else if (a.mpb) hipLaunchKernelGGL((newton_sweep2_kernel<NB, 1>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL((newton_sweep2_kernel<NB, 0>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
  return hipGetLastError();
}

// Two teams per operating point instead of one: when the one-sided sweep would leave the SIMDs with fewer than ~3 waves.
bool newton_sweep_two_sided(int nb, int nx, int64_t B, int mode, const Options& opt) {
  if (nb < 6 || nx < 8) return false;
  if (opt.newton_kernel != NK_AUTO && opt.newton_kernel != NK_WORKGROUP) return opt.newton_kernel == NK_BOTH;      // "both ends" (tests, probes)
  // with homogeneous reactions (the 4096-lane CO2R sweep: stationary solves along a continuation, 3...30 iterations per lane) the
  // lane-team kernel stays ahead: 0.48 s against 0.52 s (one-sided sweep 0.83 s)
  if (mode >= 2) return false;
  // measured (tools/probe/sweep2_probe.py, profiles/r02_newton_two_sided_sweep.txt): ahead of both the lane-team kernel and the
  // one-sided sweep while the latter would have less than one wave per SIMD (N = 8, nx = 512: B = 2048 1.67e5 against 1.37e5 / 0.99e5
  // timesteps/s, B = 4096 2.77e5 against 1.40e5 / 1.90e5; N = 6, nx = 1024: B = 8192 2.64e5 against 1.07e5 / 2.05e5); level with the
  // one-sided sweep from two waves per SIMD on, behind it in between (N = 8, B = 8192: 2.72e5 against 3.05e5)
  const int64_t tpw = 64 / nb;
  const int64_t waves1 = (B + tpw - 1) / tpw, waves2 = (B + tpw / 2 - 1) / (tpw / 2);
  return waves1 < 1024 && waves2 >= 512;
}

// Large blocks and a batch that fills the chip with teams on its own (measured): the sweep kernel.
bool newton_sweep_preferred(int nb, int nx, int64_t B, int mode, const Options& opt) {
  if (opt.newton_kernel != NK_AUTO && opt.newton_kernel != NK_WORKGROUP)
    return (opt.newton_kernel == NK_SWEEP && nb >= 3) || (opt.newton_kernel == NK_BOTH && nb >= 6 && nx >= 8);
  if (newton_sweep_two_sided(nb, nx, B, mode, opt)) return true;
  // at least one wave of teams per SIMD (1024 SIMDs): below that the chip is not full and, with uniform control flow, a wave
  // waits for its slowest lane -- the CO2R example (7 species, 4096 lanes, iteration counts 3...30) took 0.84 s instead of 0.51 s
  const int64_t waves = (B + 64 / nb - 1) / (64 / nb);
  // (N = 4: the pair kernel in its 512-register build is ahead of the sweep at every batch, 1.4e6 against 0.35-0.99e6 timesteps/s
  // with steric ions at nx = 512 -- profiles/r03_pair_crossover.jsonl; round 2 sent steric / reacting N = 4 batches here from
  // 15 k lanes on because the 256-register pair kernel spilled)
  return nb >= 6 && waves >= 1024;
}

template <int NB, int TMAX>
static hipError_t launch_newton_nb(const NewtonArgs& a, int blocks, hipStream_t stream) {
  static const Options kDefaults;
  const Options& opt = a.opt ? *a.opt : kDefaults;
  const bool generic = opt.newton_kernel == NK_GENERIC;     // forces the row-per-thread kernel (tests)
  const int tp = generic ? 0 : newton_pair_threads(NB, a.nx);
  if constexpr (NB >= 3) {
    if (a.sweep && a.sweep_blocks > 0 && newton_sweep_preferred(NB, a.nx, a.B, a.rt ? 2 : (a.mpb ? 1 : 0), opt)) {
      if constexpr (NB >= 6) {
        if (newton_sweep_two_sided(NB, a.nx, a.B, a.rt ? 2 : (a.mpb ? 1 : 0), opt)) return launch_sweep2<NB>(a, stream);
      }
      return launch_sweep<NB>(a, stream);
    }
  }
  if constexpr (NB >= 3) {     // lane teams: every large block, and the N = 2..4 grids too long for the pair kernel
    if (a.work && !generic && (NB >= 6 || tp == 0 || opt.newton_kernel == NK_TEAM)) return launch_team<NB>(a, blocks, stream);
  }
  if constexpr (NB <= 5) {
    if (tp > 0) {
      if (tp <= 64) return launch_pair<NB, 64>(a, blocks, tp, stream);
      if (tp <= 128) return launch_pair<NB, 128>(a, blocks, tp, stream);
      if (tp <= 256) return launch_pair<NB, 256>(a, blocks, tp, stream);
      if constexpr (NB <= 4) return launch_pair<NB, 512>(a, blocks, tp, stream);
    }
  }
  // row-per-thread kernel: small shapes the other two do not take, and (up to N = 6) the forced cross-check of the tests;
  // for N >= 7 it is not built (it only spilled there, and its instances dominated the compile time)
  if constexpr (NB >= 8) {
    if (a.work) return launch_team<NB>(a, blocks, stream);
    return hipErrorInvalidValue;
  } else {
    const int T = newton_threads(NB, a.nx);
    const size_t lds = a.work ? 0 : newton_exchange_doubles(NB, a.nx) * sizeof(double);
    if (a.rt) {
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)newton_kernel<NB, TMAX, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((newton_kernel<NB, TMAX, 2>), dim3(blocks), dim3(T), lds, stream, a);
    } else if (a.mpb) {
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)newton_kernel<NB, TMAX, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((newton_kernel<NB, TMAX, 1>), dim3(blocks), dim3(T), lds, stream, a);
    } else {
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)newton_kernel<NB, TMAX, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((newton_kernel<NB, TMAX, 0>), dim3(blocks), dim3(T), lds, stream, a);
    }
    return hipGetLastError();
  }
}

hipError_t launch_newton(const NewtonArgs& a, int blocks, hipStream_t stream) {
  const bool regs512 = a.opt && a.opt->newton_regs == 512;
  switch (a.N + 1) {
    case 2: return launch_newton_nb<2, 512>(a, blocks, stream);
    case 3: return launch_newton_nb<3, 512>(a, blocks, stream);
    case 4: return launch_newton_nb<4, 512>(a, blocks, stream);
    case 5: return launch_newton_nb<5, 512>(a, blocks, stream);
    // (the row-per-thread kernel of these two block sizes also exists in a 512-register build -- launch bound 256, accumulator
    // registers as spill space -- selected by CATINT_NEWTON_REGS=512: tests/test_gpu_newton.py checks that it walks the same bits)
    case 6: return regs512 ? launch_newton_nb<6, 256>(a, blocks, stream) : launch_newton_nb<6, 512>(a, blocks, stream);
    case 7: return regs512 ? launch_newton_nb<7, 256>(a, blocks, stream) : launch_newton_nb<7, 512>(a, blocks, stream);
    case 8: return launch_newton_nb<8, 512>(a, blocks, stream);
    case 9: return launch_newton_nb<9, 512>(a, blocks, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace pnp
