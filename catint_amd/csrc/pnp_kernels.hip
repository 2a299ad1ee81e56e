// Hand-written gfx950 (MI355X / CDNA4) kernels for the batched 1D Poisson-Nernst-Planck timestep.
//
// Mapping: ONE 1D GRID PER WAVEFRONT.  A 64-lane wave owns one operating point;
// lane l owns P consecutive interior unknowns r = l*P .. l*P+P-1 (grid points r+1) in registers.
// Rows are streamed HBM -> (coalesced 16 B/lane) -> LDS -> (blocked, bank-conflict-free padded
// layout) -> registers, one species at a time, so HBM sees exactly the algorithmic traffic:
// N concentration rows + 1 charge row in, the same out, per timestep.
//
// Per timestep (reference: catint/calculator_old.py, time-loop bodies :512-558 (CN), :990-1023 (FTCS)):
//   1. Poisson for the lagged potential from the charge row (get_potential_and_gradient :680-819):
//      the constant-coefficient Dirichlet problem is two wave-level prefix scans (no linear solve);
//      the reference's prefix-sum branches (:787-803) are the same scans with other directions.
//   2. per species: Robin wall BC / Dirichlet bulk BC, stencil assembly with the reference's index
//      conventions (add_field :473-491, add_boundary_values :493-500, row-vector RHS product :553),
//      tridiagonal solve = per-lane substructuring (Thomas on the P-1 interior rows against the two
//      interface unknowns) + a 64-unknown parallel cyclic reduction across the wave with
//      ds_bpermute shuffles, back-substitution, and accumulation of the next step's charge row.
// No MFMA: the work is O(N*nx) fp64 VALU on streamed bytes.
#include "pnp_internal.h"

namespace pnp {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double fast_rcp(double x) {
  // v_rcp_f64 seed + two Newton steps (what hipcc's own fdiv expansion uses, minus the
  // div_scale/div_fixup range handling that well-conditioned pivots do not need)
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

// wave-private LDS hand-off: LDS operations of one wave execute in order, so only the compiler
// must be kept from moving accesses across the hand-off point.
__device__ __forceinline__ void lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double shfl_up0(double v, int s, int lane) {
  double t = __shfl_up(v, s, 64);
  return lane >= s ? t : 0.0;
}
__device__ __forceinline__ double shfl_dn0(double v, int s, int lane) {
  double t = __shfl_down(v, s, 64);
  return lane + s < 64 ? t : 0.0;
}

// padded LDS index of grid point i: one pad double per P points makes the blocked access
// (lane stride P doubles) hit 64 distinct banks for ds_read_b64 / ds_write_b64.
template <int P>
__device__ __forceinline__ int pidx(int i) {
  return i + i / P;
}

template <int P>
__host__ __device__ constexpr int rowbuf_doubles(int ldx) {
  return ((ldx + ldx / P + 2) + 1) & ~1;
}

// coalesced global row (16 B per lane, 1 KiB per wave instruction) -> padded LDS row
template <int P>
__device__ __forceinline__ void load_row(const double* __restrict__ g, double* buf, int ldx, int lane) {
  constexpr int IT = P / 2 + 1;  // ldx <= 64*P + 16
  double2 t[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = 2 * lane + 128 * it;
    if (e < ldx) t[it] = *reinterpret_cast<const double2*>(g + e);
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = 2 * lane + 128 * it;
    if (e < ldx) {
      buf[pidx<P>(e)] = t[it].x;
      buf[pidx<P>(e + 1)] = t[it].y;
    }
  }
}

template <int P>
__device__ __forceinline__ void store_row(double* __restrict__ g, const double* buf, int ldx, int lane) {
  constexpr int IT = P / 2 + 1;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int e = 2 * lane + 128 * it;
    if (e < ldx) {
      double2 t;
      t.x = buf[pidx<P>(e)];
      t.y = buf[pidx<P>(e + 1)];
      *reinterpret_cast<double2*>(g + e) = t;
    }
  }
}

// Inclusive scan of the wave's 64*P blocked values (lane-major order). REV = suffix scan.
template <int P, bool REV>
__device__ __forceinline__ void blocked_scan(double (&x)[P], int lane, double& total) {
  if (!REV) {
#pragma unroll
    for (int j = 1; j < P; ++j) x[j] += x[j - 1];
    double inc = x[P - 1];
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      double u = __shfl_up(inc, s, 64);
      if (lane >= s) inc += u;
    }
    total = __shfl(inc, 63, 64);
    double base = shfl_up0(inc, 1, lane);
#pragma unroll
    for (int j = 0; j < P; ++j) x[j] += base;
  } else {
#pragma unroll
    for (int j = P - 2; j >= 0; --j) x[j] += x[j + 1];
    double inc = x[0];
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      double u = __shfl_down(inc, s, 64);
      if (lane + s < 64) inc += u;
    }
    total = __shfl(inc, 0, 64);
    double base = shfl_dn0(inc, 1, lane);
#pragma unroll
    for (int j = 0; j < P; ++j) x[j] += base;
  }
}

// ------------------------------------------------------------------------------------------------
// Tridiagonal solve of 64*P unknowns held P per lane (rows pre-scaled to unit diagonal):
//     a[j]*x[r-1] + x[r] + c[j]*x[r+1] = d[j],   r = lane*P + j
// a of the first and c of the last row of the wave must be 0.  The solution overwrites d.
// Stage 1 (per lane, registers): Thomas elimination of the P-1 interior rows against the two
//   interface unknowns yL = x[last row of lane-1] and y = x[last row of this lane].
// Stage 2 (across the wave): the 64 interface rows form a tridiagonal system solved by
//   parallel cyclic reduction in log2(64) = 6 shuffle steps.
// Stage 3: back-substitution of the interior rows.
// ------------------------------------------------------------------------------------------------
template <int P>
__device__ __forceinline__ void tridiag_wave(const double (&a)[P], const double (&c)[P], double (&d)[P], int lane) {
  constexpr int Q = (P > 1) ? P - 1 : 1;
  double Vn[Q], Wn[Q], dn[Q];
  double ra, rc, rd;
  if constexpr (P == 1) {
    ra = a[0];
    rc = c[0];
    rd = d[0];
  } else {
    double r[Q];
    Vn[0] = a[0];
    dn[0] = d[0];
    r[0] = 1.0;
#pragma unroll
    for (int i = 1; i < P - 1; ++i) {
      const double mlt = a[i] * r[i - 1];
      const double bb = __builtin_fma(-mlt, c[i - 1], 1.0);
      dn[i] = __builtin_fma(-mlt, dn[i - 1], d[i]);
      Vn[i] = -mlt * Vn[i - 1];
      r[i] = fast_rcp(bb);
    }
    Wn[P - 2] = c[P - 2];
#pragma unroll
    for (int i = P - 3; i >= 0; --i) {
      const double mlt = c[i] * r[i + 1];
      dn[i] = __builtin_fma(-mlt, dn[i + 1], dn[i]);
      Vn[i] = __builtin_fma(-mlt, Vn[i + 1], Vn[i]);
      Wn[i] = -mlt * Wn[i + 1];
    }
#pragma unroll
    for (int i = 0; i < P - 1; ++i) {
      Vn[i] *= r[i];
      Wn[i] *= r[i];
      dn[i] *= r[i];
    }
    // first interior row of the next lane closes this lane's interface row
    const double dn0 = shfl_dn0(dn[0], 1, lane);
    const double Vn0 = shfl_dn0(Vn[0], 1, lane);
    const double Wn0 = shfl_dn0(Wn[0], 1, lane);
    const double aL = a[P - 1], cL = c[P - 1];
    double rb = __builtin_fma(-aL, Wn[P - 2], 1.0);
    rb = __builtin_fma(-cL, Vn0, rb);
    ra = -aL * Vn[P - 2];
    rc = -cL * Wn0;
    rd = __builtin_fma(-aL, dn[P - 2], d[P - 1]);
    rd = __builtin_fma(-cL, dn0, rd);
    const double rr = fast_rcp(rb);
    ra *= rr;
    rc *= rr;
    rd *= rr;
  }
  // parallel cyclic reduction over the 64 interface rows (unit diagonal kept by renormalising)
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const double aL = shfl_up0(ra, s, lane), cL = shfl_up0(rc, s, lane), dL = shfl_up0(rd, s, lane);
    const double aR = shfl_dn0(ra, s, lane), cR = shfl_dn0(rc, s, lane), dR = shfl_dn0(rd, s, lane);
    double nb = __builtin_fma(-ra, cL, 1.0);
    nb = __builtin_fma(-rc, aR, nb);
    double nd = __builtin_fma(-ra, dL, rd);
    nd = __builtin_fma(-rc, dR, nd);
    const double na = -ra * aL;
    const double nc = -rc * cR;
    const double rr = fast_rcp(nb);
    ra = na * rr;
    rc = nc * rr;
    rd = nd * rr;
  }
  const double y = rd;
  if constexpr (P > 1) {
    const double yL = shfl_up0(y, 1, lane);
#pragma unroll
    for (int i = 0; i < P - 1; ++i) {
      double t = __builtin_fma(-Vn[i], yL, dn[i]);
      d[i] = __builtin_fma(-Wn[i], y, t);
    }
  }
  d[P - 1] = y;
}

// ------------------------------------------------------------------------------------------------
// Poisson for one wave.  LV holds lapl_v by grid index (padded LDS row).  Writes grad_v by grid
// index into GV (all nx entries incl. the extrapolated ends) and, if VV != nullptr, v into VV.
// Returns v[1] (needed by the Robin wall condition, calculator_old.py:528-532).
// ------------------------------------------------------------------------------------------------
template <int P>
__device__ __forceinline__ double poisson_wave(const DevArgs& A, const double* LV, double* GV, double* VV,
                                               double vw, double vb, double gw, double gb, int lane) {
  const int nx = A.nx, m = A.m;
  const int r0 = lane * P;
  const double dx = A.dx;
  double vown[P], gown[P];
  if (A.pb_mode == PNP_PB_DD) {
    // v'' = lapl with v[0]=vw, v[nx-1]=vb (solve_poisson :716-730) as two prefix scans:
    // w_i = v_{i+1}-v_i = w_0 + H_i,  H_i = sum_{j=1..i} h_j,  h = lapl*dx^2
    // v_i = vw + i*w_0 + G_i,          G_i = sum_{j=1..i-1} H_j,  w_0 from v_{nx-1} = vb.
    double h[P], Hi[P], Hx[P];
    const double dx2 = dx * dx;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int r = r0 + j;
      h[j] = (r < m) ? LV[pidx<P>(r + 1)] * dx2 : 0.0;
      Hi[j] = h[j];
    }
    double tot1;
    blocked_scan<P, false>(Hi, lane, tot1);  // Hi[j] = H_{grid r+1}
    double G[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int r = r0 + j;
      Hx[j] = Hi[j] - h[j];                  // H_{grid r}
      G[j] = (r < m) ? Hi[j] : 0.0;
    }
    double totG;
    blocked_scan<P, false>(G, lane, totG);   // inclusive; exclusive = G[j] - own
    const double w0 = (vb - vw - totG) / (double)(nx - 1);
    const double inv2dx = 1.0 / (2 * dx);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int r = r0 + j;
      const double Gex = G[j] - ((r < m) ? Hi[j] : 0.0);   // G_{grid r+1}
      vown[j] = vw + (double)(r + 1) * w0 + Gex;
      gown[j] = inv2dx * ((w0 + Hi[j]) + (w0 + Hx[j]));    // (v[i+1]-v[i-1])/(2dx), :784
    }
  } else {
    const bool g_from_wall = (A.pb_mode == PNP_PB_GWALL_VBULK) || (A.pb_mode == PNP_PB_VWALL_GWALL);
    const bool v_from_wall = (A.pb_mode == PNP_PB_VWALL_GBULK) || (A.pb_mode == PNP_PB_VWALL_GWALL);
    double t[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int r = r0 + j;
      t[j] = (r < m) ? LV[pidx<P>(r + 1)] * dx : 0.0;
    }
    double tot;
    if (g_from_wall) {  // grad_v[i] = grad_v[i-1] + lapl_v[i]*dx, :761
      blocked_scan<P, false>(t, lane, tot);
#pragma unroll
      for (int j = 0; j < P; ++j) gown[j] = gw + t[j];
    } else {            // grad_v[i] = grad_v[i+1] - lapl_v[i]*dx, :759
      blocked_scan<P, true>(t, lane, tot);
#pragma unroll
      for (int j = 0; j < P; ++j) gown[j] = gb - t[j];
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int r = r0 + j;
      t[j] = (r < m) ? gown[j] * dx : 0.0;
    }
    if (v_from_wall) {
      blocked_scan<P, false>(t, lane, tot);
#pragma unroll
      for (int j = 0; j < P; ++j) vown[j] = vw + t[j];
    } else {
      blocked_scan<P, true>(t, lane, tot);
#pragma unroll
      for (int j = 0; j < P; ++j) vown[j] = vb - t[j];
    }
  }
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const int r = r0 + j;
    if (r < m) {
      GV[pidx<P>(r + 1)] = gown[j];
      if (VV) VV[pidx<P>(r + 1)] = vown[j];
    }
  }
  lds_sync();
  if (lane == 0) {
    // extrapolated / prescribed end values, :785-786, :790-803
    const double g1 = GV[pidx<P>(1)], g2 = GV[pidx<P>(2)];
    const double gm1 = GV[pidx<P>(nx - 2)], gm2 = GV[pidx<P>(nx - 3)];
    double g0, gl;
    if (A.pb_mode == PNP_PB_DD) {
      g0 = g1 + (g1 - g2);
      gl = gm1 + (gm1 - gm2);
    } else if (A.pb_mode == PNP_PB_GWALL_VBULK || A.pb_mode == PNP_PB_VWALL_GWALL) {
      g0 = gw;
      gl = gm1 + (gm1 - gm2);
    } else {
      gl = gb;
      g0 = g1 + (g1 - g2);
    }
    GV[pidx<P>(0)] = g0;
    GV[pidx<P>(nx - 1)] = gl;
    if (VV) {
      const double v1 = VV[pidx<P>(1)], v2 = VV[pidx<P>(2)];
      const double vm1 = VV[pidx<P>(nx - 2)], vm2 = VV[pidx<P>(nx - 3)];
      double v0, vl;
      if (A.pb_mode == PNP_PB_DD) {
        v0 = vw;
        vl = vb;
      } else if (A.pb_mode == PNP_PB_VWALL_GBULK || A.pb_mode == PNP_PB_VWALL_GWALL) {
        v0 = vw;
        vl = vm1 + (vm1 - vm2);
      } else {
        vl = vb;
        v0 = v1 + (v1 - v2);
      }
      VV[pidx<P>(0)] = v0;
      VV[pidx<P>(nx - 1)] = vl;
    }
  }
  lds_sync();
  return __shfl(vown[0], 0, 64);
}

// ------------------------------------------------------------------------------------------------
// The timestep kernel: grid = B blocks of one wave; dynamic LDS = 3 padded rows.
// ------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(64) void step_kernel(const DevArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int nx = A.nx, m = A.m, ldx = A.ldx, N = A.N;
  const int rb = rowbuf_doubles<P>(ldx);
  double* ROW = lds;
  double* GV = lds + rb;
  double* LV = lds + 2 * rb;
  const int r0 = lane * P;
  const double dx = A.dx, dt = A.dt;

  const double vw = A.pb[b * 4 + 0], vb = A.pb[b * 4 + 1], gw = A.pb[b * 4 + 2], gb = A.pb[b * 4 + 3];
  const double vz = A.vzeta[b];
  double* lin = A.lapl_a + b * (int64_t)ldx;
  double* lout = A.lapl_b + b * (int64_t)ldx;
  double* crow0 = A.c + b * (int64_t)N * ldx;
  const double inv_eps = 1.0 / A.eps;
  int bad = 0;

  for (int step = 0; step < A.nsteps; ++step) {
    // ---- 1. lagged potential ------------------------------------------------------------------
    double v1 = 0.0;
    if (A.use_mig) {
      load_row<P>(lin, LV, ldx, lane);
      lds_sync();
      v1 = poisson_wave<P>(A, LV, GV, nullptr, vw, vb, gw, gb, lane);
    }
    // species-independent stencil inputs: grad_v at grid r0-1 .. r0+P+1, lapl_v at grid r0 .. r0+P-1
    double gq[P + 3], lq[P];
#pragma unroll
    for (int t = 0; t < P + 3; ++t) {
      const int gi = r0 - 1 + t;
      gq[t] = (A.use_mig && gi >= 0 && gi < nx) ? GV[pidx<P>(gi)] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < P; ++j) lq[j] = (A.use_mig && r0 + j < nx) ? LV[pidx<P>(r0 + j)] : 0.0;
    const double g_first = A.use_mig ? GV[pidx<P>(0)] : 0.0;       // grad_v[0]
    const double g_last = A.use_mig ? GV[pidx<P>(nx - 1)] : 0.0;   // grad_v[-1]
    lds_sync();

    double acc[P];   // next step's charge row at own points
#pragma unroll
    for (int j = 0; j < P; ++j) acc[j] = 0.0;
    double acc0 = 0.0, accL = 0.0;

    // ---- 2. species ------------------------------------------------------------------------------
    for (int k = 0; k < N; ++k) {
      double* crow = crow0 + (int64_t)k * ldx;
      load_row<P>(crow, ROW, ldx, lane);
      lds_sync();
      double cc[P + 2];   // C[k] at grid r0 .. r0+P+1
#pragma unroll
      for (int t = 0; t < P + 2; ++t) {
        const int gi = r0 + t;
        cc[t] = (gi < nx) ? ROW[pidx<P>(gi)] : 0.0;
      }
      const double c1 = ROW[pidx<P>(1)];
      const double c0old = ROW[pidx<P>(0)];
      const double cLold = ROW[pidx<P>(nx - 1)];
      const double Dk = A.D[k], qk = A.q[k];
      const double muk = Dk * qk * A.beta;                 // transport.py:436
      const double flux = A.flux[b * N + k];
      const double cL = A.cbulk[b * N + k];                // C[k,-1] = C0[(k+1)*nx-1], :540 / :1008
      double x[P];
      double c0new;
      if (A.method == PNP_METHOD_CRANK_NICOLSON) {
        // Robin wall condition :528-532
        const double aa = muk * (v1 - vz);
        const double den = -2 * Dk + aa;
        c0new = (-2 * Dk - aa) / den * c1 - 2 * flux * dx / den;
        double s = Dk * dt / (dx * dx);                    // :543
        if (A.lf) s += 0.5;
        const double ee = qk * A.beta * dt * Dk;           // :547
        const double hs = 0.5 * s;
        const double rdiag = 1.0 / (1.0 + s);
        const double e4 = ee / 4. / dx;
        double ta[P], tc[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const int r = r0 + j;
          const double gm = e4 * gq[j];       // g_{r-1}
          const double g0 = e4 * gq[j + 1];   // g_r      (grad_v[r]: interior index, not r+1 -- :483-490)
          const double gp = e4 * gq[j + 2];   // g_{r+1}
          // B = C[k,1:-1] . B1  (row-vector x matrix, :553):
          //   B[r] = c[r-1]*B1[r-1,r] + c[r]*B1[r,r] + c[r+1]*B1[r+1,r]
          double left, right;
          if (r == 0) left = (hs + e4 * g_first) * (c0new + c0old);            // :496-497
          else left = cc[j] * (hs + gm);
          if (r == m - 1) right = (hs - e4 * g_last) * (cL + cLold);           // :498-499
          else right = cc[j + 2] * (hs - gp);
          const double diag = (1 - s) + ee * lq[j];
          double rhs = left + cc[j + 1] * diag + right;
          double av = (r == 0) ? 0.0 : (-hs + g0);          // A[r,r-1], :487
          double cv = (r == m - 1) ? 0.0 : (-hs - g0);      // A[r,r+1], :490
          if (r >= m) {
            av = 0.0;
            cv = 0.0;
            rhs = 0.0;
          }
          ta[j] = av * rdiag;
          tc[j] = cv * rdiag;
          x[j] = rhs * rdiag;
        }
        tridiag_wave<P>(ta, tc, x, lane);                   // np.linalg.solve(A,B), :556
      } else {
        // FTCS :1001-1023
        const double aa = muk * (v1 - vz);
        const double divisor = 2 * Dk - aa;
        c0new = ((2 * Dk + aa) * c1 + flux * 2. * dx) / divisor;
        const double s = Dk * dt / (dx * dx);
        const double dm = dt / (2. * dx) * muk;
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const int r = r0 + j;                             // grid i = r+1
          double W = s - dm * gq[j + 3] + 0.5;              // grad_v[i+1]
          double M = -2. * Dk * dt / (dx * dx);
          double E = s + dm * gq[j + 1] + 0.5;              // grad_v[i-1]
          if (!A.lf) {
            W -= 0.5;
            E -= 0.5;
            M += 1;
          }
          const double cm = (r == 0) ? c0new : cc[j];
          const double cp = (r == m - 1) ? cL : cc[j + 2];
          double val = E * cm + M * cc[j + 1] + W * cp;
          if (A.has_rates && r < m) val += A.rates[(b * N + k) * (int64_t)ldx + r + 1] * dt;
          x[j] = val;
        }
      }
      // status + next charge row
      const double qe = qk * inv_eps;
#pragma unroll
      for (int j = 0; j < P; ++j) {
        if (r0 + j < m) {
          if (!(x[j] - x[j] == 0.0)) bad |= 2;
          else if (x[j] < 0.0) bad |= 1;
          acc[j] = __builtin_fma(-x[j], qe, acc[j]);
        }
      }
      if (!(c0new - c0new == 0.0)) bad |= 2;
      else if (c0new < 0.0) bad |= 1;
      acc0 = __builtin_fma(-c0new, qe, acc0);
      accL = __builtin_fma(-cL, qe, accL);
      lds_sync();
#pragma unroll
      for (int j = 0; j < P; ++j) {
        if (r0 + j < m) ROW[pidx<P>(r0 + j + 1)] = x[j];
      }
      if (lane == 0) {
        ROW[pidx<P>(0)] = c0new;
        ROW[pidx<P>(nx - 1)] = cL;
      }
      lds_sync();
      store_row<P>(crow, ROW, ldx, lane);
      lds_sync();
    }
    // ---- 3. charge row of the new state (lagged by the next step) -----------------------------
#pragma unroll
    for (int j = 0; j < P; ++j) {
      if (r0 + j < m) LV[pidx<P>(r0 + j + 1)] = acc[j];
    }
    if (lane == 0) {
      LV[pidx<P>(0)] = acc0;
      LV[pidx<P>(nx - 1)] = accL;
    }
    lds_sync();
    store_row<P>(lout, LV, ldx, lane);
    lds_sync();
    double* tmp = lin;
    lin = lout;
    lout = tmp;
    // the wave re-reads its own rows in the next fused step: make the stores visible first
    if (step + 1 < A.nsteps) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }
  // lane status (replaces check_error / NaN test, calculator.py:409-414)
  const unsigned long long nan_mask = __ballot(bad & 2);
  const unsigned long long neg_mask = __ballot(bad & 1);
  if (lane == 0) {
    int st = PNP_STATUS_OK;
    if (neg_mask) st = PNP_STATUS_NEGATIVE;
    if (nan_mask) st = PNP_STATUS_NAN;
    if (st > A.status[b]) A.status[b] = st;   // sticky until the next pnp_set_batch
  }
}

// stand-alone Poisson (read-back of tp.potential / tp.efield): one wave per lane of the batch
template <int P>
__global__ __launch_bounds__(64) void poisson_kernel(const DevArgs A, const double* __restrict__ lapl,
                                                      double* __restrict__ v, double* __restrict__ gradv) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int ldx = A.ldx;
  const int rb = rowbuf_doubles<P>(ldx);
  double* VV = lds;
  double* GV = lds + rb;
  double* LV = lds + 2 * rb;
  // zero the pads so that the pitch tail written back is deterministic
  for (int i = lane; i < 3 * rb; i += 64) lds[i] = 0.0;
  lds_sync();
  load_row<P>(lapl + b * (int64_t)ldx, LV, ldx, lane);
  lds_sync();
  poisson_wave<P>(A, LV, GV, VV, A.pb[b * 4 + 0], A.pb[b * 4 + 1], A.pb[b * 4 + 2], A.pb[b * 4 + 3], lane);
  store_row<P>(v + b * (int64_t)ldx, VV, ldx, lane);
  store_row<P>(gradv + b * (int64_t)ldx, GV, ldx, lane);
}

// lapl[b][i] = -sum_k q_k c[b][k][i]/eps  (:767-771), thread per grid point
__global__ void charge_row_kernel(const DevArgs A, double* __restrict__ lapl) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = A.B * A.ldx;
  if (idx >= total) return;
  const int64_t b = idx / A.ldx;
  const int i = (int)(idx - b * A.ldx);
  double acc = 0.0;
  if (i < A.nx) {
    const double inv_eps = 1.0 / A.eps;
    for (int k = 0; k < A.N; ++k) acc = __builtin_fma(-A.c[(b * A.N + k) * (int64_t)A.ldx + i], A.q[k] * inv_eps, acc);
  }
  lapl[idx] = acc;
}

// get_rates (:159-208) with the reference's overwrite order, thread per grid point
__global__ void rates_kernel(const DevArgs A, const ReactionTable rt, double* __restrict__ rates) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = A.B * A.nx;
  if (idx >= total) return;
  const int64_t b = idx / A.nx;
  const int i = (int)(idx - b * A.nx);
  double cv[PNP_MAX_SPECIES], rv[PNP_MAX_SPECIES];
  for (int k = 0; k < A.N; ++k) {
    cv[k] = A.c[(b * A.N + k) * (int64_t)A.ldx + i];
    rv[k] = 0.0;
  }
  for (int r = 0; r < rt.n; ++r) {
    double pl = 1.0, pr = 1.0;
    for (int j = 0; j < rt.n_lhs[r]; ++j) pl *= cv[rt.lhs[r][j]];
    for (int j = 0; j < rt.n_rhs[r]; ++j) pr *= cv[rt.rhs[r][j]];
    for (int j = 0; j < rt.n_lhs[r]; ++j) {
      const int k = rt.lhs[r][j];
      rv[k] = 0.0;
      rv[k] -= pl * rt.kf[r];
      rv[k] += pr * rt.kr[r];
    }
    for (int j = 0; j < rt.n_rhs[r]; ++j) {
      const int k = rt.rhs[r][j];
      rv[k] = 0.0;
      rv[k] += pl * rt.kf[r];
      rv[k] -= pr * rt.kr[r];
    }
  }
  for (int k = 0; k < A.N; ++k) rates[(b * A.N + k) * (int64_t)A.ldx + i] = rv[k];
}

__global__ void surface_kernel(const DevArgs A, double* __restrict__ csurf) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= A.B * A.N) return;
  csurf[idx] = A.c[idx * (int64_t)A.ldx];
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
int points_per_lane(int nx) {
  const int m = nx - 2;
  if (nx < 5) return 0;
  for (int P : {1, 2, 4, 8, 16}) {
    if (m <= 64 * P) return P;
  }
  return 0;
}

size_t step_lds_bytes(int ldx, int P) {
  int rb = 0;
  switch (P) {
    case 1: rb = rowbuf_doubles<1>(ldx); break;
    case 2: rb = rowbuf_doubles<2>(ldx); break;
    case 4: rb = rowbuf_doubles<4>(ldx); break;
    case 8: rb = rowbuf_doubles<8>(ldx); break;
    case 16: rb = rowbuf_doubles<16>(ldx); break;
    default: return 0;
  }
  return (size_t)3 * rb * sizeof(double);
}

hipError_t launch_step(const DevArgs& a, hipStream_t stream) {
  const int P = points_per_lane(a.nx);
  const size_t lds = step_lds_bytes(a.ldx, P);
  const dim3 grid((unsigned)a.B), block(64);
  switch (P) {
    case 1: hipLaunchKernelGGL(step_kernel<1>, grid, block, lds, stream, a); break;
    case 2: hipLaunchKernelGGL(step_kernel<2>, grid, block, lds, stream, a); break;
    case 4: hipLaunchKernelGGL(step_kernel<4>, grid, block, lds, stream, a); break;
    case 8: hipLaunchKernelGGL(step_kernel<8>, grid, block, lds, stream, a); break;
    case 16: hipLaunchKernelGGL(step_kernel<16>, grid, block, lds, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_poisson(const DevArgs& a, const double* lapl, double* v, double* gradv, hipStream_t stream) {
  const int P = points_per_lane(a.nx);
  const size_t lds = step_lds_bytes(a.ldx, P);
  const dim3 grid((unsigned)a.B), block(64);
  switch (P) {
    case 1: hipLaunchKernelGGL(poisson_kernel<1>, grid, block, lds, stream, a, lapl, v, gradv); break;
    case 2: hipLaunchKernelGGL(poisson_kernel<2>, grid, block, lds, stream, a, lapl, v, gradv); break;
    case 4: hipLaunchKernelGGL(poisson_kernel<4>, grid, block, lds, stream, a, lapl, v, gradv); break;
    case 8: hipLaunchKernelGGL(poisson_kernel<8>, grid, block, lds, stream, a, lapl, v, gradv); break;
    case 16: hipLaunchKernelGGL(poisson_kernel<16>, grid, block, lds, stream, a, lapl, v, gradv); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_charge_row(const DevArgs& a, double* lapl, hipStream_t stream) {
  const int64_t total = a.B * a.ldx;
  const int threads = 256;
  const unsigned blocks = (unsigned)((total + threads - 1) / threads);
  hipLaunchKernelGGL(charge_row_kernel, dim3(blocks), dim3(threads), 0, stream, a, lapl);
  return hipGetLastError();
}

hipError_t launch_rates(const DevArgs& a, const ReactionTable& rt, double* rates, hipStream_t stream) {
  const int64_t total = a.B * a.nx;
  const int threads = 256;
  const unsigned blocks = (unsigned)((total + threads - 1) / threads);
  hipLaunchKernelGGL(rates_kernel, dim3(blocks), dim3(threads), 0, stream, a, rt, rates);
  return hipGetLastError();
}

hipError_t launch_surface(const DevArgs& a, double* csurf, hipStream_t stream) {
  const int64_t total = a.B * a.N;
  const int threads = 256;
  const unsigned blocks = (unsigned)((total + threads - 1) / threads);
  hipLaunchKernelGGL(surface_kernel, dim3(blocks), dim3(threads), 0, stream, a, csurf);
  return hipGetLastError();
}

}  // namespace pnp
