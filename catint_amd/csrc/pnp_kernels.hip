// Hand-written gfx950 (MI355X / CDNA4) kernels for the batched 1D Poisson-Nernst-Planck timestep.
//
// Mapping: ONE 1D GRID PER WAVEFRONT.  A 64-lane wave owns one tridiagonal system
// of one operating point; lane l owns P consecutive interior unknowns r = l*P .. l*P+P-1 (grid
// points r+1) in registers.  A workgroup = W waves works on ONE operating point: wave w advances
// species w, w+W, ... (the species solves of one timestep are independent because the reference
// lags the potential), so small batches still fill the chip; W = 1 for large batches.
// Rows stream HBM -> (coalesced 16 B/lane) -> LDS -> (blocked, bank-conflict-free padded layout)
// -> registers and back, one species at a time, so HBM sees exactly the algorithmic traffic:
// N concentration rows + 1 charge row in, the same out, per timestep.
//
// Per timestep (reference: catint/calculator_old.py, time-loop bodies :512-558 (CN), :990-1023 (FTCS)):
//   1. Poisson for the lagged potential from the charge row (get_potential_and_gradient :680-819):
//      the constant-coefficient Dirichlet problem is two wave-level prefix scans (no linear solve);
//      the reference's prefix-sum branches (:787-803) are the same scans with other directions.
//   2. per species: Robin wall BC / Dirichlet bulk BC, stencil assembly with the reference's index
//      conventions (add_field :473-491, add_boundary_values :493-500, row-vector RHS product :553),
//      tridiagonal solve = per-lane substructuring (Thomas on the P-1 interior rows against the two
//      interface unknowns) + a 64-unknown parallel cyclic reduction across the wave whose neighbour
//      exchange goes through LDS (ds_read2_b64 with zero guard slots), back-substitution.
//   3. cooperative epilogue: every thread stores its coalesced share of the new rows and folds them,
//      in species order, into the charge row the next step will lag.
// No MFMA: the work is O(N*nx) fp64 VALU on streamed bytes.
#include "pnp_internal.h"
#include "pnp_wave.h"

namespace pnp {

// ------------------------------------------------------------------------------------------------
// Poisson for one wave.  LV holds lapl_v by grid index (padded LDS row).  Writes grad_v by grid
// index into GV (all nx entries incl. the extrapolated ends) and, if VV != nullptr, v into VV.
// Returns v[1] (needed by the Robin wall condition, calculator_old.py:528-532).
// ------------------------------------------------------------------------------------------------
template <int P, bool WANT_V, int SHIFT>
__device__ __forceinline__ double poisson_wave(const DevArgs& A, double* LV, double* GV, double* VV, double* X,
                                               double vw, double vb, double gw, double gb, int lane) {
  const int nx = A.nx, m = A.m;
  const int r0 = lane * P;
  const double dx = A.dx;
  double vown[P], gown[P];
  double v1;   // v at grid point 1
  if constexpr (!WANT_V && P >= 2) {
    if (A.pb_mode == PNP_PB_DD) {
      // The timestep kernels' Dirichlet-Dirichlet fast path: step_kernel_rr / step_kernel_st's register arithmetic statement for
      // statement (DPP prefix scans instead of six LDS round trips), so that every kernel family walks the same bits:
      //   w_i = v_{i+1}-v_i = w_0 + H_i,  H_i = sum_{j=1..i} h_j,  h = lapl*dx^2,  G_{nx-1} = sum_r (m - r) h_r,
      //   w_0 = (vb - vw - G_{nx-1})/(nx-1),  grad_v[i] = ((w_0 + H_i) + (w_0 + H_{i-1}))/(2dx),  ends extrapolated (:785-786)
      double Hi[P];
      const double dx2 = A.dx2;
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int j = 0; j < P; ++j) {
        double h = LV[pidx<P>(r0 + j + 1)] * dx2;
        h = (r0 + j == m) ? 0.0 : h;                      // the bulk point is not part of the interior sum
        Hi[j] = h;
        s0 += h;
        s1 = __builtin_fma((double)j, h, s1);
      }
      const double wsum = __builtin_fma((double)(m - r0), s0, -s1);
      const double hm1 = pick_blocked<P>(Hi, r0, m - 1), hm2 = pick_blocked<P>(Hi, r0, m - 2);
      const double h0 = read_lane(Hi[0], 0), h1 = read_lane(Hi[1], 0);
#pragma unroll
      for (int j = 1; j < P; ++j) Hi[j] += Hi[j - 1];
      const double incT = wave_scan_incl(Hi[P - 1]);
      const double incW = wave_scan_incl(wsum);
      const double base = from_prev_lane(0.0, incT);
      const double tot1 = read_lane(incT, 63), totG = read_lane(incW, 63);
#pragma unroll
      for (int j = 0; j < P; ++j) Hi[j] += base;          // Hi[j] = H_{grid r0+j+1}
      const double w0 = (vb - vw - totG) / A.nxm1;
      const double inv2dx = A.inv2dx;
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const double Hx = (j == 0) ? base : Hi[j > 0 ? j - 1 : 0];
        GV[pidx<P>(r0 + j + 1 + SHIFT)] = inv2dx * ((w0 + Hi[j]) + (w0 + Hx));     // (v[i+1]-v[i-1])/(2dx), :784
      }
      const double g1 = inv2dx * ((w0 + h0) + (w0 + 0.0));
      const double g2 = inv2dx * ((w0 + (h0 + h1)) + (w0 + h0));
      const double g_first = g1 + (g1 - g2);
      const double Hm1 = tot1 - hm1, Hm2 = Hm1 - hm2;     // H_{nx-3}, H_{nx-4}; H_{nx-2} = tot1
      const double gm1 = inv2dx * ((w0 + tot1) + (w0 + Hm1));
      const double gm2 = inv2dx * ((w0 + Hm1) + (w0 + Hm2));
      const double g_last = gm1 + (gm1 - gm2);
      lds_sync();
      if (lane == 0) {
        GV[pidx<P>(0 + SHIFT)] = g_first;
        GV[pidx<P>(nx - 1 + SHIFT)] = g_last;
        if constexpr (SHIFT == 1) {
          GV[pidx<P>(0)] = g_first;          // interior index -1 clamps to grad_v[0]  (add_boundary_values :496)
          GV[pidx<P>(nx + 1)] = g_last;      // one past the end, read by the FTCS window of padded rows
        }
      }
      lds_sync();
      return vw + w0;
    }
  }
  if (A.pb_mode == PNP_PB_DD) {
    // v'' = lapl with v[0]=vw, v[nx-1]=vb (solve_poisson :716-730) as two prefix scans:
    // w_i = v_{i+1}-v_i = w_0 + H_i,  H_i = sum_{j=1..i} h_j,  h = lapl*dx^2
    // v_i = vw + i*w_0 + G_i,          G_i = sum_{j=1..i-1} H_j,  w_0 from v_{nx-1} = vb.
    double Hi[P];
    const double dx2 = dx * dx;
    // The scans run over all 64*P slots without per-row predicates: LV is zero beyond the row and the
    // bulk boundary entry (not part of the interior sum) is blanked here; nothing reads it afterwards.
    if (lane == 0) LV[pidx<P>(nx - 1)] = 0.0;
    lds_sync();
#pragma unroll
    for (int j = 0; j < P; ++j) Hi[j] = LV[pidx<P>(r0 + j + 1)] * dx2;
    double tot1, baseH, totG;
    const double inv2dx = 1.0 / (2 * dx);
    if constexpr (!WANT_V) {
      // only v[1] and grad_v are needed: G_{nx-1} = sum_r (m - r) h_r is a plain weighted reduction that
      // rides along with the one scan (half the dependent LDS round trips of the two-scan form)
      double wsum = 0.0;
      const double wj0 = (double)(m - r0);
#pragma unroll
      for (int j = 0; j < P; ++j) wsum = __builtin_fma(wj0 - (double)j, Hi[j], wsum);
      blocked_scan_sum<P>(Hi, wsum, X, lane, tot1, baseH, totG);
    } else {
      double G[P];
      blocked_scan<P, false>(Hi, X, lane, tot1, baseH);   // Hi[j] = H_{grid r+1}; H_{grid r} = Hi[j-1] | baseH
#pragma unroll
      for (int j = 0; j < P; ++j) G[j] = Hi[j];
      double totAll, baseG;
      blocked_scan<P, false>(G, X, lane, totAll, baseG);  // G_{grid r+1} = G[j-1] | baseG  (exclusive)
      // the 64*P - m padded slots each carried the full sum tot1: remove them from the grand total
      totG = totAll - (double)(64 * P - m) * tot1;
      const double w0v = (vb - vw - totG) / (double)(nx - 1);
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const double Gex = (j == 0) ? baseG : G[j > 0 ? j - 1 : 0];
        vown[j] = vw + (double)(r0 + j + 1) * w0v + Gex;
      }
    }
    const double w0 = (vb - vw - totG) / (double)(nx - 1);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const double Hx = (j == 0) ? baseH : Hi[j > 0 ? j - 1 : 0];
      gown[j] = inv2dx * ((w0 + Hi[j]) + (w0 + Hx));     // (v[i+1]-v[i-1])/(2dx), :784
    }
    v1 = vw + 1.0 * w0 + 0.0;
  } else {
    const bool g_from_wall = (A.pb_mode == PNP_PB_GWALL_VBULK) || (A.pb_mode == PNP_PB_VWALL_GWALL);
    const bool v_from_wall = (A.pb_mode == PNP_PB_VWALL_GBULK) || (A.pb_mode == PNP_PB_VWALL_GWALL);
    double t[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const double lv = LV[pidx<P>(r0 + j + 1)];
      t[j] = (r0 + j < m) ? lv * dx : 0.0;
    }
    double tot, base;
    if (g_from_wall) {  // grad_v[i] = grad_v[i-1] + lapl_v[i]*dx, :761
      blocked_scan<P, false>(t, X, lane, tot, base);
#pragma unroll
      for (int j = 0; j < P; ++j) gown[j] = gw + t[j];
    } else {            // grad_v[i] = grad_v[i+1] - lapl_v[i]*dx, :759
      blocked_scan<P, true>(t, X, lane, tot, base);
#pragma unroll
      for (int j = 0; j < P; ++j) gown[j] = gb - t[j];
    }
#pragma unroll
    for (int j = 0; j < P; ++j) t[j] = (r0 + j < m) ? gown[j] * dx : 0.0;
    if (v_from_wall) {
      blocked_scan<P, false>(t, X, lane, tot, base);
#pragma unroll
      for (int j = 0; j < P; ++j) vown[j] = vw + t[j];
    } else {
      blocked_scan<P, true>(t, X, lane, tot, base);
#pragma unroll
      for (int j = 0; j < P; ++j) vown[j] = vb - t[j];
    }
    v1 = __shfl(vown[0], 0, 64);
  }
#pragma unroll
  for (int j = 0; j < P; ++j) {
    // grad_v[i] lives in slot i+SHIFT.  Padded rows (r >= m) store finite don't-care values past the
    // interior; the two end entries are set right below.
    GV[pidx<P>(r0 + j + 1 + SHIFT)] = gown[j];
    if constexpr (WANT_V) VV[pidx<P>(r0 + j + 1)] = vown[j];
  }
  lds_sync();
  if (lane == 0) {
    // extrapolated / prescribed end values, :785-786, :790-803
    const double g1 = GV[pidx<P>(1 + SHIFT)], g2 = GV[pidx<P>(2 + SHIFT)];
    const double gm1 = GV[pidx<P>(nx - 2 + SHIFT)], gm2 = GV[pidx<P>(nx - 3 + SHIFT)];
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
    GV[pidx<P>(0 + SHIFT)] = g0;
    GV[pidx<P>(nx - 1 + SHIFT)] = gl;
    if constexpr (SHIFT == 1) {
      GV[pidx<P>(0)] = g0;          // interior index -1 clamps to grad_v[0]  (add_boundary_values :496)
      GV[pidx<P>(nx + 1)] = gl;     // one past the end, read by the FTCS window of padded rows
    }
    if constexpr (WANT_V) {
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
  return v1;
}

// ------------------------------------------------------------------------------------------------
// The timestep kernel: grid = B workgroups of W waves, every wave advances G species at a time;
// dynamic LDS = (2 + W*G) padded rows:  LV (lagged charge row) | GV (grad_v) | ROW[W*G] (staged rows)
// ------------------------------------------------------------------------------------------------
template <int P, int G>
constexpr int step_min_waves() {
  return P <= 2 ? 4 : (P == 4 ? (G == 1 ? 4 : 2) : (P == 8 ? (G == 1 ? 3 : 1) : 1));
}

template <int P, int W, int G>
__global__ __launch_bounds__(64 * W, (step_min_waves<P, G>())) void step_kernel(const DevArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int RB = rowbuf_doubles<P>();
  constexpr int NR = W * G;              // species rows staged per round
  constexpr int V1SLOT = RB - 2;         // v[1] broadcast slot inside GV
  constexpr int IT2 = P / (2 * W) + 1;   // coalesced 16-byte chunks owned by one thread
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = blockIdx.x;
  const int nx = A.nx, m = A.m, ldx = A.ldx, N = A.N;
  double* LV = lds;
  double* GV = lds + RB;
  double* ROWS = lds + 2 * RB;
  double* ROW0 = ROWS + wave * (G * RB);   // this wave's G staged rows: ROW0 + g*RB
  const int r0 = lane * P;
  const double dx = A.dx, dt = A.dt;

  double* lin = A.lapl_a + b * (int64_t)ldx;
  double* lout = A.lapl_b + b * (int64_t)ldx;
  double* crow0 = A.c + b * (int64_t)N * ldx;
  const bool single_round = (N <= NR);   // every species row stays staged in LDS between fused steps
  const bool cn = (A.method == PNP_METHOD_CRANK_NICOLSON);
  double chk = 0.0;                      // NaN/Inf detector: sum of (x - x)
  double mn = 0.0;                       // most negative concentration seen
  // Every LDS slot is finite from here on: rows of padded unknowns (r >= m) read past the row ends.
  for (int i = tid; i < (2 + NR) * RB; i += 64 * W) lds[i] = 0.0;
  wg_sync<W>();
  const int ls2 = pidx<P>(2 * tid);      // LDS slot of this thread's first coalesced pair
  // Wave 0 carries the Poisson scan on top of its species: it is the wave the others wait for at every barrier, so the SIMD it
  // shares with waves of other workgroups issues its instructions first -- for the whole step while the batch leaves the chip
  // latency-bound (headline shape: 5.37 -> 4.97 us per step), during the scan only once the batch oversubscribes it
  // (B = 8192: 41.2 -> 40.2 us per step; the whole-step priority costs 2 % there).
  const bool lead_all = W > 1 && wave == 0 && A.B <= 2048;
  if (lead_all) __builtin_amdgcn_s_setprio(3);

  for (int step = 0; step < A.nsteps; ++step) {
    const bool resident = single_round && step > 0;   // rows (and LV) already in LDS from the last step
    const bool last_step = step + 1 == A.nsteps;
    // ---- 0. all global reads of the first round go out together ------------------------------------
    if (!resident) {
      RowRegs<P> rl, rr[G];
      if (wave == 0 && A.use_mig) load_row_issue<P>(row_rsrc(lin, ldx), rl, lane);
      if (wave * G < N) {
#pragma unroll
        for (int g = 0; g < G; ++g)
          load_row_issue<P>(row_rsrc(crow0 + (int64_t)min(wave * G + g, N - 1) * ldx, ldx), rr[g], lane);
      }
      if (wave == 0 && A.use_mig) load_row_commit<P>(rl, LV, lane);
      if (wave * G < N) {
#pragma unroll
        for (int g = 0; g < G; ++g) load_row_commit<P>(rr[g], ROW0 + g * RB, lane);
      }
      lds_sync();
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the staging registers' live range inside the block above
    // ---- 1. lagged potential (wave 0) ------------------------------------------------------------
    if (A.use_mig) {
      if (wave == 0) {
        const double vw = A.pb[b * 4 + 0], vb = A.pb[b * 4 + 1], gw = A.pb[b * 4 + 2], gb = A.pb[b * 4 + 3];
        // grad_v[i] is kept in slot i+1 (slot 0 duplicates grad_v[0]) so that every stencil window
        // below is an affine, clamp-free LDS address
        // (the scans use the head of GV as their exchange strip: grad_v of the previous step is dead by now)
        // The lane index is made opaque once per step: the scan's lane-dependent weights ((double)(m - r0 - j), ...) are then
        // recomputed where they are used (one conversion each) instead of being hoisted out of the time loop and -- in the
        // instances that sit at their register budget -- reloaded from spill slots, four dependent scratch loads per step.
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        if (W > 1 && !lead_all) __builtin_amdgcn_s_setprio(3);
        const double v1w = poisson_wave<P, false, 1>(A, LV, GV, nullptr, GV, vw, vb, gw, gb, lane_o);
        if (W > 1 && !lead_all) __builtin_amdgcn_s_setprio(0);
        if (lane == 0) {
          GV[V1SLOT] = v1w;
          // CN indexes grad_v with the interior index r <= nx-3 plus, for the bulk boundary term,
          // grad_v[-1]: park the latter in the otherwise unused entry nx-2 (see the stencil below)
          if (cn) GV[pidx<P>(nx - 2 + 1)] = GV[pidx<P>(nx - 1 + 1)];
        }
      }
    }
    wg_sync<W>();
    const double v1 = A.use_mig ? GV[V1SLOT] : 0.0;
    const double vz = A.vzeta[b];

    d2 accp[IT2];   // next step's charge row at this thread's coalesced positions
#pragma unroll
    for (int it = 0; it < IT2; ++it) accp[it] = (d2)(0.0);

    // ---- 2. species, W*G at a time ------------------------------------------------------------------
    for (int k0 = 0; k0 < N; k0 += NR) {
      const int kw = k0 + wave * G;          // first species of this wave in this round
      if (kw < N) {
        int kg[G];
#pragma unroll
        for (int g = 0; g < G; ++g) kg[g] = min(kw + g, N - 1);   // a short last group recomputes species N-1
        if (k0 > 0) {   // later rounds: the first round's rows were loaded above
          RowRegs<P> rr[G];
#pragma unroll
          for (int g = 0; g < G; ++g) load_row_issue<P>(row_rsrc(crow0 + (int64_t)kg[g] * ldx, ldx), rr[g], lane);
#pragma unroll
          for (int g = 0; g < G; ++g) load_row_commit<P>(rr[g], ROW0 + g * RB, lane);
          lds_sync();
        }
        // ---- wall / bulk boundary values ----------------------------------------------------------
        double c0new[G], cLv[G], pat0[G], patL[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const double* ROW = ROW0 + g * RB;
          const SpecConst& S = A.spec[kg[g]];
          const double flux = A.flux[b * N + kg[g]];
          const double cL = A.cbulk[b * N + kg[g]];            // C[k,-1] = C0[(k+1)*nx-1], :540 / :1008
          const double c1 = ROW[pidx<P>(1)];
          const double c0old = ROW[pidx<P>(0)];
          const double cLold = ROW[pidx<P>(nx - 1)];
          const double aa = S.mu * (v1 - vz);
          if (cn) {   // Robin wall condition :528-532
            const double rden = fast_rcp2(-S.twoD + aa);
            c0new[g] = (-S.twoD - aa) * rden * c1 - 2 * flux * dx * rden;
          } else {    // :1003-1006
            c0new[g] = ((S.twoD + aa) * c1 + flux * 2. * dx) * fast_rcp2(S.twoD - aa);
          }
          cLv[g] = cL;
          // Patch the two boundary slots so that the stencil below needs no per-row special cases:
          //   CN  : add_boundary_values (:496-499) multiplies (C0+C0_old) resp. (C1+C1_old)
          //   FTCS: the interior update reads the freshly set boundary values (:1010-1011, :1022)
          pat0[g] = cn ? (c0new[g] + c0old) : c0new[g];
          patL[g] = cn ? (cL + cLold) : cL;
        }
        lds_sync();
        if (lane == 0) {
#pragma unroll
          for (int g = 0; g < G; ++g) {
            ROW0[g * RB + pidx<P>(0)] = pat0[g];
            ROW0[g * RB + pidx<P>(nx - 1)] = patL[g];
          }
        }
        lds_sync();
        double x[G][P];
        if (cn) {
          double ta[G][P], tc[G][P];
          {
            // stencil inputs shared by the G species: grad_v at interior index r0-1 .. r0+P (entry shifted by
            // one), lapl_v at interior index r0 .. r0+P-1.  grad_v is indexed with the INTERIOR index r (not
            // r+1) as in add_field :483-490; the entry of interior index nx-2 holds grad_v[-1] (see the
            // Poisson section), index -1 holds grad_v[0]: exactly what add_boundary_values uses.
            const int jmu = (m - 1) % P;              // the last real row is row jmu of lane lm (both wave-uniform)
            const bool in_lm = lane == (m - 1) / P;
#pragma unroll
            for (int g = 0; g < G; ++g) {
              const double* ROW = ROW0 + g * RB;
              const SpecConst& S = A.spec[kg[g]];
              // every row is divided by the constant diagonal 1+s up front: the scaled constants come
              // from the species table, so the unit-diagonal rows cost no extra multiplies
              const double hsr = S.hsr, e4r = S.e4r, eer = S.eer, omsr = S.omsr;
              double cc[P + 2], g4[P + 2];
#pragma unroll
              for (int t = 0; t < P + 2; ++t) cc[t] = ROW[pidx<P>(r0 + t)];
#pragma unroll
              for (int t = 0; t < P + 2; ++t) g4[t] = e4r * GV[pidx<P>(r0 + t)];
#pragma unroll
              for (int j = 0; j < P; ++j) {
                // B = C[k,1:-1] . B1  (row-vector x matrix, :553):
                //   B[r] = c[r-1]*B1[r-1,r] + c[r]*B1[r,r] + c[r+1]*B1[r+1,r]
                const double lq = LV[pidx<P>(r0 + j)];
                const double left = cc[j] * (hsr + g4[j]);
                const double right = cc[j + 2] * (hsr - g4[j + 2]);
                x[g][j] = left + cc[j + 1] * (omsr + eer * lq) + right;
                ta[g][j] = -hsr + g4[j + 1];                               // A[r,r-1], :487
                tc[g][j] = -hsr - g4[j + 1];                               // A[r,r+1], :490
                if (j == jmu) tc[g][j] = in_lm ? 0.0 : tc[g][j];           // ... which the last real row lacks
                // rows r >= m only see finite inputs and row m-1 has no super-diagonal, so they form a
                // benign trailing block that never feeds back into the real unknowns
              }
              if (lane == 0) ta[g][0] = 0.0;                               // first row has no sub-diagonal
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          lds_sync();
          tridiag_wave<P, G>(ta, tc, x, ROW0, RB, lane);       // np.linalg.solve(A,B), :556
        } else {
          // FTCS :1012-1023
          double gq[P + 2];   // grad_v at grid r0 .. r0+P+1
#pragma unroll
          for (int t = 0; t < P + 2; ++t) gq[t] = A.use_mig ? GV[pidx<P>(r0 + t + 1)] : 0.0;
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const double* ROW = ROW0 + g * RB;
            const SpecConst& S = A.spec[kg[g]];
            const double sf = S.sf, dm = S.dm, Mf = S.Mf;
            double cc[P + 2];
#pragma unroll
            for (int t = 0; t < P + 2; ++t) cc[t] = ROW[pidx<P>(r0 + t)];
#pragma unroll
            for (int j = 0; j < P; ++j) {
              double Wt = sf - dm * gq[j + 2] + 0.5;            // grad_v[i+1], grid i = r0+j+1
              double Et = sf + dm * gq[j] + 0.5;                // grad_v[i-1]
              if (!A.lf) {
                Wt -= 0.5;
                Et -= 0.5;
              }
              double val = Et * cc[j] + Mf * cc[j + 1] + Wt * cc[j + 2];
              if (A.has_rates) val += A.rates[(b * N + kg[g]) * (int64_t)ldx + min(r0 + j + 1, nx - 1)] * dt;
              x[g][j] = val;
            }
          }
          lds_sync();
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          double* ROW = ROW0 + g * RB;
#pragma unroll
          for (int j = 0; j < P; ++j) ROW[pidx<P>(r0 + j + 1)] = x[g][j];   // padded rows land past the row (don't care)
        }
        lds_sync();
#pragma unroll
        for (int g = 0; g < G; ++g) {
          double* ROW = ROW0 + g * RB;
          if (lane == 0) {
            ROW[pidx<P>(0)] = c0new[g];
            ROW[pidx<P>(nx - 1)] = cLv[g];
          }
          // the pitch tail [nx, ldx) shares LDS with the reduction's exchange area: keep it zero
          if (lane < ldx - nx) ROW[pidx<P>(nx + lane)] = 0.0;
        }
      }
      wg_sync<W>();
      // ---- 3. cooperative epilogue: store the round's rows, fold them into the next charge row ----
      const int nk = min(NR, N - k0);
      for (int w2 = 0; w2 < nk; ++w2) {
        const double* R2 = ROWS + w2 * RB;
        double* grow = crow0 + (int64_t)(k0 + w2) * ldx;
        const double qe = A.spec[k0 + w2].qe;
        const __amdgpu_buffer_rsrc_t rs = row_rsrc(grow, ldx);
#pragma unroll
        for (int it = 0; it < IT2; ++it) {
          // threads past the staged row (W > 1: the workgroup spans more than one row buffer) read a
          // clamped, don't-care slot; their global store is dropped by the range check
          const int sl = min(pair_slot<P>(ls2, W * it), RB - 4);
          d2 t;
          t.x = R2[sl];
          t.y = R2[sl + PAIR_STEP<P>];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, t), rs, tid * 16 + it * (1024 * W), 0, 0);
          accp[it].x = __builtin_fma(-t.x, qe, accp[it].x);
          accp[it].y = __builtin_fma(-t.y, qe, accp[it].y);
          // beyond the pitch the staged row holds the don't-care values of the padded unknowns
          if (last_step) {   // status of the state this launch leaves behind (the reference tests after the solve, calculator.py:409-414)
            const bool inrow = 2 * tid + 128 * W * it < ldx;
            const double sx = inrow ? t.x : 0.0, sy = inrow ? t.y : 0.0;
            chk += (sx - sx) + (sy - sy);
            mn = fmin(mn, fmin(sx, sy));
          }
        }
      }
      wg_sync<W>();
    }
    // ---- 4. charge row of the new state (lagged by the next step) -------------------------------
    const bool keep = single_round && (step + 1 < A.nsteps);
    {
      const __amdgpu_buffer_rsrc_t rs = row_rsrc(lout, ldx);
#pragma unroll
      for (int it = 0; it < IT2; ++it) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, accp[it]), rs, tid * 16 + it * (1024 * W), 0, 0);
        // LV stays zero beyond the row (the Poisson scans run unmasked): only in-row pairs are written
        if (keep && 2 * tid + 128 * W * it < ldx) {
          LV[pair_slot<P>(ls2, W * it)] = accp[it].x;
          LV[pair_slot<P>(ls2, W * it) + PAIR_STEP<P>] = accp[it].y;
        }
      }
    }
    double* tmp = lin;
    lin = lout;
    lout = tmp;
    if (step + 1 < A.nsteps) {
      if (single_round) {
        wg_sync<W>();
      } else {
        // rows are re-read from global memory by other threads of this workgroup
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if constexpr (W > 1) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
    }
  }
  // lane status (replaces check_error / NaN test, calculator.py:409-414)
  const unsigned long long nan_mask = __ballot(chk != chk);
  const unsigned long long neg_mask = __ballot(mn < 0.0);
  if (lane == 0) {
    int st = PNP_STATUS_OK;
    if (neg_mask) st = PNP_STATUS_NEGATIVE;
    if (nan_mask) st = PNP_STATUS_NAN;
    if (st) atomicMax(&A.status[b], st);   // sticky until the next pnp_set_batch
#ifdef PNP_HWID_PROBE
    // diagnosis build (tools/probe/hwid_probe.py): which SIMD ran each wave of the workgroup -- HW_ID[5:4] -- and which CU / SE
    // wave 0 ran on -- HW_ID[11:8], [15:13]
    const unsigned simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
    unsigned v = 1u << (8 + 4 * wave + simd);
    if (wave == 0) v |= (__builtin_amdgcn_s_getreg((3 << 11) | (8 << 6) | 4) << 20) | (__builtin_amdgcn_s_getreg((2 << 11) | (13 << 6) | 4) << 24);
    atomicOr((unsigned*)&A.status[b], v);
#endif
  }
}

// ================================================================================================
// Register-resident kernel family (step_kernel_rr) for the Dirichlet/Dirichlet Poisson branch.
//   * lane l loads its window of P+2 grid points (own P interior rows plus one halo point on either
//     side) of a row straight into registers with 16-byte buffer loads (each lane reads a contiguous,
//     16-B aligned piece; the windows tile the row, so HBM traffic stays algorithmic);
//   * the Poisson prefix scan and every neighbour exchange use DPP (row_shr / row_bcast / wave_shr /
//     wave_shl) -- no LDS round trips; LDS is used only for the 6-step cyclic-reduction strips and the
//     once-per-step meeting of the waves of a workgroup;
//   * in fused launches with one species per wave the state never leaves the registers between
//     timesteps (the new rows are still written to HBM every step).
// ================================================================================================
// ================================================================================================
// step_kernel_rr<P,W,CN>: W waves per operating point.
// One wave issues an fp64 VALU instruction only every ~8.5 cycles while the SIMD can take one every ~2.2
// (tools/probe/f64_rate.hip), so throughput needs >= 4 waves per SIMD and small batches want their work
// spread over several waves.  Here wave w advances species
// w, w+W, ... on its own: every wave loads the lagged charge window, runs the (cheap, DPP-only) Poisson
// scan redundantly, solves its species, and the waves meet exactly once per timestep to add up their
// contributions to the next charge row through a ping-pong LDS buffer (one s_barrier, LDS-only fences).
// ================================================================================================
template <int P, int W>
constexpr int step_rr_min_waves() {
  return P <= 4 ? 4 : (P == 8 ? 3 : 1);
}

template <int P, int W, bool CN>
__global__ __launch_bounds__(64 * W, (step_rr_min_waves<P, W>())) void step_kernel_rr(const DevArgs A) {
  static_assert(P >= 2 && P % 2 == 0, "window loads need an even P");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int XS = 384;                      // cyclic-reduction strip of one wave
  constexpr int PS = 64 * P + 2;               // one wave's partial charge row: 64*P own rows + wall + bulk entry
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = blockIdx.x;
  const int nx = A.nx, m = A.m, ldx = A.ldx, N = A.N;
  const int r0 = lane * P;
  const double dx = A.dx;
  constexpr bool cn = CN;
  double* strip = lds + wave * XS;
  double* part = lds + W * XS;                 // [2][W][PS]
  double* lin = A.lapl_a + b * (int64_t)ldx;
  double* lout = A.lapl_b + b * (int64_t)ldx;
  double* crow0 = A.c + b * (int64_t)N * ldx;
  const double vw = A.pb[b * 4 + 0], vb = A.pb[b * 4 + 1];
  const double vz = A.vzeta[b];
  const bool single_round = (N <= W);          // one species per wave: it never leaves the registers
  double chk = 0.0, mn = 0.0;

  double lw[P + 2];        // lagged charge row window: lapl_v[r0 + t]
  double cc[P + 2];        // concentration window of the species in flight: C[k][r0 + t]
  for (int step = 0; step < A.nsteps; ++step) {
    const bool last_step = step + 1 == A.nsteps;
    const bool resident = single_round && step > 0;
    if (!resident) {
      if (A.use_mig) load_window<P>(row_rsrc(lin, ldx), lw, lane);
      load_window<P>(row_rsrc(crow0 + (int64_t)min(wave, N - 1) * ldx, ldx), cc, lane);
    }
    // ---- 1. lagged potential: v'' = lapl, v[0] = vw, v[nx-1] = vb  (calculator_old.py:716-730, :780-786) ----
    double gx[P + 3];   // grad_v[r0 - 1 + t]
    double v1 = 0.0;
    if (A.use_mig) {
      double Hi[P];
      const double dx2 = dx * dx;
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int j = 0; j < P; ++j) {
        double h = lw[j + 1] * dx2;                       // grid r0+j+1; pads and out-of-range are 0
        h = (r0 + j == m) ? 0.0 : h;                      // the bulk point is not part of the interior sum
        Hi[j] = h;
        s0 += h;
        s1 = __builtin_fma((double)j, h, s1);
      }
      // G_{nx-1} = sum_r (m - r) h_r, this lane's share: (m - r0) * sum_j h_j - sum_j j*h_j
      const double wsum = __builtin_fma((double)(m - r0), s0, -s1);
      const double hm1 = pick_blocked<P>(Hi, r0, m - 1), hm2 = pick_blocked<P>(Hi, r0, m - 2);
      const double h0 = read_lane(Hi[0], 0), h1 = read_lane(Hi[1], 0);
#pragma unroll
      for (int j = 1; j < P; ++j) Hi[j] += Hi[j - 1];
      const double incT = wave_scan_incl(Hi[P - 1]);
      const double incW = wave_scan_incl(wsum);
      const double base = from_prev_lane(0.0, incT);
      const double tot1 = read_lane(incT, 63), totG = read_lane(incW, 63);
#pragma unroll
      for (int j = 0; j < P; ++j) Hi[j] += base;          // Hi[j] = H_{grid r0+j+1}
      const double w0 = (vb - vw - totG) / (double)(nx - 1);
      v1 = vw + w0;
      const double inv2dx = 1.0 / (2 * dx);
      double gown[P];                                     // grad_v[r0+j+1] = (v[i+1]-v[i-1])/(2dx), :784
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const double Hx = (j == 0) ? base : Hi[j > 0 ? j - 1 : 0];
        gown[j] = inv2dx * ((w0 + Hi[j]) + (w0 + Hx));
      }
      // extrapolated ends :785-786 from the first / last two interior charges
      const double g1 = inv2dx * ((w0 + h0) + (w0 + 0.0));
      const double g2 = inv2dx * ((w0 + (h0 + h1)) + (w0 + h0));
      const double g_first = g1 + (g1 - g2);
      const double Hm1 = tot1 - hm1, Hm2 = Hm1 - hm2;     // H_{nx-3}, H_{nx-4}; H_{nx-2} = tot1
      const double gm1 = inv2dx * ((w0 + tot1) + (w0 + Hm1));
      const double gm2 = inv2dx * ((w0 + Hm1) + (w0 + Hm2));
      const double g_last = gm1 + (gm1 - gm2);
#pragma unroll
      for (int t = 2; t < P + 2; ++t) gx[t] = gown[t - 2];
      gx[1] = from_prev_lane(g_first, gown[P - 1]);       // grad_v[r0]    (lane 0: grad_v[0])
      gx[0] = from_prev_lane(g_first, gown[P - 2]);       // grad_v[r0-1]  (lane 0: index -1 -> grad_v[0], :496)
      gx[P + 2] = from_next_lane(0.0, gown[0]);           // grad_v[r0+P+1]
      // the bulk boundary term uses grad_v[-1] (:498): CN reads it at interior index nx-2, FTCS at grid nx-1
#pragma unroll
      for (int t = 2; t < P + 3; ++t) {
        const int idx = r0 - 1 + t;                        // gx[t] = grad_v[idx]
        gx[t] = ((cn && idx == nx - 2) || idx == nx - 1) ? g_last : gx[t];
      }
    } else {
#pragma unroll
      for (int t = 0; t < P + 3; ++t) gx[t] = 0.0;
    }

    double acc[P];            // this wave's contribution to the next charge row at the own rows
#pragma unroll
    for (int j = 0; j < P; ++j) acc[j] = 0.0;
    double acc0 = 0.0, accL = 0.0;

    // ---- 2. this wave's species: wave, wave+W, ... ------------------------------------------------------
    for (int k = wave; k < N; k += W) {
      if (k != wave) load_window<P>(row_rsrc(crow0 + (int64_t)k * ldx, ldx), cc, lane);
      const SpecConst& S = A.spec[k];
      const double flux = A.flux[b * N + k];
      const double cL = A.cbulk[b * N + k];                // C[k,-1] = C0[(k+1)*nx-1] (:540 / :1008) -- also COLD[k,-1]
      const double c0old = read_lane(cc[0], 0);
      const double c1 = read_lane(cc[1], 0);
      const double aa = S.mu * (v1 - vz);
      double c0new;
      if (cn) {   // Robin wall condition :528-532
        const double rden = fast_rcp2(-S.twoD + aa);
        c0new = (-S.twoD - aa) * rden * c1 - 2 * flux * dx * rden;
      } else {    // :1003-1006
        c0new = ((S.twoD + aa) * c1 + flux * 2. * dx) * fast_rcp2(S.twoD - aa);
      }
      const double qe = S.qe;
      // boundary entries of the window: CN multiplies (C0+C0_old) / (C1+C1_old) (:496-499); FTCS reads the
      // freshly set boundary values (:1010-1011, :1022)
      const double pat0 = cn ? (c0new + c0old) : c0new;
      const double patL = cn ? (cL + cL) : cL;
      cc[0] = (lane == 0) ? pat0 : cc[0];
#pragma unroll
      for (int t = 2; t < P + 2; ++t) cc[t] = (r0 + t == nx - 1) ? patL : cc[t];
      double x[1][P];
      if (cn) {
        double ta[1][P], tc[1][P];
        // rows are divided by the constant diagonal 1+s up front (constants pre-scaled on the host)
        const double hsr = S.hsr, e4r = S.e4r, eer = S.eer, omsr = S.omsr;
#pragma unroll
        for (int j = 0; j < P; ++j) {
          // grad_v / lapl_v carry the INTERIOR index r (add_field :483-490); RHS = C[k,1:-1] . B1 (:553):
          //   B[r] = c[r-1]*B1[r-1,r] + c[r]*B1[r,r] + c[r+1]*B1[r+1,r]
          const double gm = e4r * gx[j], g0 = e4r * gx[j + 1], gp = e4r * gx[j + 2];
          const double left = cc[j] * (hsr + gm);
          const double right = cc[j + 2] * (hsr - gp);
          x[0][j] = left + cc[j + 1] * (omsr + eer * lw[j]) + right;
          ta[0][j] = -hsr + g0;                                          // A[r,r-1], :487
          tc[0][j] = (r0 + j == m - 1) ? 0.0 : (-hsr - g0);              // A[r,r+1], :490; none in the last real row
        }
        ta[0][0] = (lane == 0) ? 0.0 : ta[0][0];                         // the first row has no sub-diagonal
        tridiag_wave<P, 1>(ta, tc, x, strip, XS, lane);                  // np.linalg.solve(A,B), :556
      } else {
        const double sf = S.sf, dm = S.dm, Mf = S.Mf;
#pragma unroll
        for (int j = 0; j < P; ++j) {                                    // grid i = r0+j+1, :1012-1022
          double Wt = sf - dm * gx[j + 3] + 0.5;                         // grad_v[i+1]
          double Et = sf + dm * gx[j + 1] + 0.5;                         // grad_v[i-1]
          if (!A.lf) {
            Wt -= 0.5;
            Et -= 0.5;
          }
          x[0][j] = Et * cc[j] + Mf * cc[j + 1] + Wt * cc[j + 2];
        }
      }
      // ---- results: HBM, charge contribution, status, (fused) next window ------------------------------
      double* grow = crow0 + (int64_t)k * ldx;
      store_rows<P>(__builtin_amdgcn_make_buffer_rsrc(grow, 0, (nx - 1) * 8, 0x00020000), x[0], lane);
      if (lane == 0) grow[0] = c0new;
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const double xv = (r0 + j < m) ? x[0][j] : 0.0;
        acc[j] = __builtin_fma(-xv, qe, acc[j]);
        if (last_step) {   // status of the state this launch leaves behind (calculator.py:409-414 tests after the solve)
          chk += xv - xv;
          mn = fmin(mn, xv);
        }
      }
      if (last_step) {
        chk += c0new - c0new;
        mn = fmin(mn, c0new);
      }
      acc0 = __builtin_fma(-c0new, qe, acc0);
      accL = __builtin_fma(-cL, qe, accL);
      if (single_round) {   // next step's window from registers: own rows + one DPP hop for each halo point
#pragma unroll
        for (int j = 0; j < P; ++j) cc[j + 1] = x[0][j];
        cc[0] = from_prev_lane(c0new, x[0][P - 1]);
        cc[P + 1] = from_next_lane(0.0, x[0][0]);
      }
    }
    // ---- 3. the waves add up their charge contributions (the one meeting point of a timestep) -----------
    if constexpr (W > 1) {
      double* mine = part + ((step & 1) * W + wave) * PS;
#pragma unroll
      for (int j = 0; j < P; ++j) mine[r0 + j] = acc[j];
      if (lane == 0) {
        mine[64 * P] = acc0;
        mine[64 * P + 1] = accL;
      }
      wg_sync<W>();
      const double* all = part + (step & 1) * W * PS;
#pragma unroll
      for (int j = 0; j < P; ++j) {
        double t = all[r0 + j];
#pragma unroll
        for (int w2 = 1; w2 < W; ++w2) t += all[w2 * PS + r0 + j];
        acc[j] = t;
      }
      double t0 = all[64 * P], tL = all[64 * P + 1];
#pragma unroll
      for (int w2 = 1; w2 < W; ++w2) {
        t0 += all[w2 * PS + 64 * P];
        tL += all[w2 * PS + 64 * P + 1];
      }
      acc0 = t0;
      accL = tL;
    }
    // ---- 4. charge row of the new state ------------------------------------------------------------------
    if (wave == 0) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(lout, 0, (nx - 1) * 8, 0x00020000);
      store_rows<P>(rs, acc, lane);
      if (lane == 0) {
        lout[0] = acc0;
        lout[nx - 1] = accL;
      }
    }
    if (single_round) {
#pragma unroll
      for (int j = 0; j < P; ++j) lw[j + 1] = acc[j];
      lw[0] = from_prev_lane(acc0, acc[P - 1]);
      lw[P + 1] = 0.0;
    }
    double* tmp = lin;
    lin = lout;
    lout = tmp;
    if (!single_round && step + 1 < A.nsteps) {
      // the next step re-reads rows written by this workgroup (other waves' rows, other lanes' pieces)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if constexpr (W > 1) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
  const unsigned long long nan_mask = __ballot(chk != chk);
  const unsigned long long neg_mask = __ballot(mn < 0.0);
  if (lane == 0) {
    int st = PNP_STATUS_OK;
    if (neg_mask) st = PNP_STATUS_NEGATIVE;
    if (nan_mask) st = PNP_STATUS_NAN;
    if (st) atomicMax(&A.status[b], st);
  }
}

// ================================================================================================
// Systems spanning several waves (nx up to 64*16*4 + 2 = 4098): step_kernel_mw<P,WY>.
// A workgroup of WY waves owns one operating point; thread t = wave*64 + lane owns rows t*P .. t*P+P-1 of
// EVERY tridiagonal system (one species at a time).  The algorithm is step_kernel's with the wave-level
// pieces lifted to the workgroup: rows are staged cooperatively, the blocked scans add the totals of the
// preceding waves, and the cyclic reduction runs over the 64*WY interface rows through one shared,
// ping-pong LDS strip with an s_barrier per step.  Throughput is secondary here (one workgroup per CU);
// this kernel exists so that no grid size of the reference is out of reach.
// ================================================================================================
template <int P, int WY>
__host__ __device__ constexpr int rowbuf_mw_doubles() {
  constexpr int CAP = 128 * WY * (P / 2 + 1);
  return (CAP + CAP / P + 4 + 1) & ~1;
}
template <int WY>
__host__ __device__ constexpr int xch_doubles() {      // [2 ping-pong][3 arrays][guard | 64*WY | guard]
  return 2 * 3 * (2 * 64 * WY);
}

// workgroup-wide inclusive blocked scan: per-wave scan, then the totals of the other waves
template <int P, int WY, bool REV>
__device__ __forceinline__ void blocked_scan_wg(double (&x)[P], double* strip_w, double* tots, int lane, int wave,
                                                double& total, double& base) {
  double wt, wb;
  blocked_scan<P, REV>(x, strip_w, lane, wt, wb);
  if (lane == 0) tots[wave] = wt;
  wg_sync<2>();
  double off = 0.0, all = 0.0;
#pragma unroll
  for (int w2 = 0; w2 < WY; ++w2) {
    const double t = tots[w2];
    all += t;
    off += (REV ? (w2 > wave) : (w2 < wave)) ? t : 0.0;
  }
#pragma unroll
  for (int j = 0; j < P; ++j) x[j] += off;
  total = all;
  base = wb + off;
  wg_sync<2>();
}

// Poisson for a workgroup (all five boundary combinations); mirrors poisson_wave.
template <int P, int WY, bool WANT_V, int SHIFT>
__device__ __forceinline__ double poisson_wg(const DevArgs& A, double* LV, double* GV, double* VV, double* strip_w,
                                             double* scal, double vw, double vb, double gw, double gb, int lane, int wave) {
  const int nx = A.nx, m = A.m;
  const int tid = wave * 64 + lane;
  const int r0 = tid * P;
  const double dx = A.dx;
  double vown[P], gown[P];
  double* tots = scal;          // WY doubles
  if (A.pb_mode == PNP_PB_DD) {
    double Hi[P], G[P];
    const double dx2 = dx * dx;
    if (tid == 0) LV[pidx<P>(nx - 1)] = 0.0;    // the bulk point is not part of the interior sums
    wg_sync<2>();
#pragma unroll
    for (int j = 0; j < P; ++j) Hi[j] = LV[pidx<P>(r0 + j + 1)] * dx2;
    double tot1, baseH;
    blocked_scan_wg<P, WY, false>(Hi, strip_w, tots, lane, wave, tot1, baseH);
#pragma unroll
    for (int j = 0; j < P; ++j) G[j] = Hi[j];
    double totAll, baseG;
    blocked_scan_wg<P, WY, false>(G, strip_w, tots, lane, wave, totAll, baseG);
    const double totG = totAll - (double)(64 * WY * P - m) * tot1;
    const double w0 = (vb - vw - totG) / (double)(nx - 1);
    const double inv2dx = 1.0 / (2 * dx);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const double Hx = (j == 0) ? baseH : Hi[j > 0 ? j - 1 : 0];
      const double Gex = (j == 0) ? baseG : G[j > 0 ? j - 1 : 0];
      gown[j] = inv2dx * ((w0 + Hi[j]) + (w0 + Hx));
      vown[j] = vw + (double)(r0 + j + 1) * w0 + Gex;
    }
  } else {
    const bool g_from_wall = (A.pb_mode == PNP_PB_GWALL_VBULK) || (A.pb_mode == PNP_PB_VWALL_GWALL);
    const bool v_from_wall = (A.pb_mode == PNP_PB_VWALL_GBULK) || (A.pb_mode == PNP_PB_VWALL_GWALL);
    double t[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const double lv = LV[pidx<P>(r0 + j + 1)];
      t[j] = (r0 + j < m) ? lv * dx : 0.0;
    }
    double tot, base;
    if (g_from_wall) {
      blocked_scan_wg<P, WY, false>(t, strip_w, tots, lane, wave, tot, base);
#pragma unroll
      for (int j = 0; j < P; ++j) gown[j] = gw + t[j];
    } else {
      blocked_scan_wg<P, WY, true>(t, strip_w, tots, lane, wave, tot, base);
#pragma unroll
      for (int j = 0; j < P; ++j) gown[j] = gb - t[j];
    }
#pragma unroll
    for (int j = 0; j < P; ++j) t[j] = (r0 + j < m) ? gown[j] * dx : 0.0;
    if (v_from_wall) {
      blocked_scan_wg<P, WY, false>(t, strip_w, tots, lane, wave, tot, base);
#pragma unroll
      for (int j = 0; j < P; ++j) vown[j] = vw + t[j];
    } else {
      blocked_scan_wg<P, WY, true>(t, strip_w, tots, lane, wave, tot, base);
#pragma unroll
      for (int j = 0; j < P; ++j) vown[j] = vb - t[j];
    }
  }
#pragma unroll
  for (int j = 0; j < P; ++j) {
    GV[pidx<P>(r0 + j + 1 + SHIFT)] = gown[j];
    if constexpr (WANT_V) VV[pidx<P>(r0 + j + 1)] = vown[j];
  }
  if (tid == 0) scal[WY] = vown[0];     // v[1]
  wg_sync<2>();
  if (tid == 0) {
    const double g1 = GV[pidx<P>(1 + SHIFT)], g2 = GV[pidx<P>(2 + SHIFT)];
    const double gm1 = GV[pidx<P>(nx - 2 + SHIFT)], gm2 = GV[pidx<P>(nx - 3 + SHIFT)];
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
    GV[pidx<P>(0 + SHIFT)] = g0;
    GV[pidx<P>(nx - 1 + SHIFT)] = gl;
    if constexpr (SHIFT == 1) {
      GV[pidx<P>(0)] = g0;
      GV[pidx<P>(nx + 1)] = gl;
    }
    if constexpr (WANT_V) {
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
  wg_sync<2>();
  return scal[WY];
}

// tridiagonal solve of 64*WY*P unknowns, P per thread (rows pre-scaled to unit diagonal), in place.
template <int P, int WY>
__device__ __forceinline__ void tridiag_wg(double (&a)[P], double (&c)[P], double (&d)[P], double* XCH, int tid) {
  constexpr int NL = 64 * WY;          // interface rows
  constexpr int ARR = 2 * NL;          // guard | NL | guard, guards of NL/2
  constexpr int GD = NL / 2;
  auto buf = [&](int e, int arr) { return XCH + ((e & 1) * 3 + arr) * ARR + GD + tid; };
  {  // zero guards of both ping-pong buffers: NL threads cover the 2*GD guard slots of each array
    const int gofs = (tid < GD) ? -GD : GD;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int arr = 0; arr < 3; ++arr) buf(e, arr)[gofs] = 0.0;
  }
  int e = 0;
#pragma unroll
  for (int i = 1; i < P - 1; ++i) {
    const double ai = a[i];
    const double bb = __builtin_fma(-ai, c[i - 1], 1.0);
    const double dd = __builtin_fma(-ai, d[i - 1], d[i]);
    const double vv = -ai * a[i - 1];
    const double r = fast_rcp(bb);
    a[i] = vv * r;
    d[i] = dd * r;
    c[i] = c[i] * r;
  }
#pragma unroll
  for (int i = P - 3; i >= 0; --i) {
    const double cs = c[i];
    d[i] = __builtin_fma(-cs, d[i + 1], d[i]);
    a[i] = __builtin_fma(-cs, a[i + 1], a[i]);
    c[i] = -cs * c[i + 1];
  }
  buf(e, 0)[0] = a[0];
  buf(e, 1)[0] = c[0];
  buf(e, 2)[0] = d[0];
  wg_sync<2>();
  const double Vn0 = buf(e, 0)[1], Wn0 = buf(e, 1)[1], dn0 = buf(e, 2)[1];
  ++e;
  double ra, rc, rd;
  {
    const double aL = a[P - 1], cL = c[P - 1];
    double rb = __builtin_fma(-aL, c[P - 2], 1.0);
    rb = __builtin_fma(-cL, Vn0, rb);
    const double rr = fast_rcp(rb);
    double t = __builtin_fma(-aL, d[P - 2], d[P - 1]);
    t = __builtin_fma(-cL, dn0, t);
    ra = (-aL * a[P - 2]) * rr;
    rc = (-cL * Wn0) * rr;
    rd = t * rr;
  }
#pragma unroll
  for (int s = 1; s < NL; s <<= 1) {
    buf(e, 0)[0] = ra;
    buf(e, 1)[0] = rc;
    buf(e, 2)[0] = rd;
    wg_sync<2>();
    const double aL = buf(e, 0)[-s], aR = buf(e, 0)[s];
    const double cL = buf(e, 1)[-s], cR = buf(e, 1)[s];
    const double dL = buf(e, 2)[-s], dR = buf(e, 2)[s];
    ++e;
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
  buf(e, 0)[0] = rd;
  wg_sync<2>();
  const double yL = buf(e, 0)[-1];
#pragma unroll
  for (int i = 0; i < P - 1; ++i) {
    const double t = __builtin_fma(-a[i], yL, d[i]);
    d[i] = __builtin_fma(-c[i], rd, t);
  }
  d[P - 1] = rd;
  wg_sync<2>();   // the strips are free again
}

template <int P, int WY>
__global__ __launch_bounds__(64 * WY, 1) void step_kernel_mw(const DevArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int RB = rowbuf_mw_doubles<P, WY>();
  constexpr int IT = P / 2 + 1;          // coalesced 16-byte chunks owned by one thread
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = blockIdx.x;
  const int nx = A.nx, m = A.m, ldx = A.ldx, N = A.N;
  double* LV = lds;
  double* GV = lds + RB;
  double* ROW = lds + 2 * RB;
  double* XCH = lds + 3 * RB;
  double* strip_w = XCH + xch_doubles<WY>() + wave * 256;     // per-wave scan strip
  double* scal = XCH + xch_doubles<WY>() + WY * 256;          // WY wave totals + v[1]
  const int r0 = tid * P;
  const double dx = A.dx, dt = A.dt;
  double* lin = A.lapl_a + b * (int64_t)ldx;
  double* lout = A.lapl_b + b * (int64_t)ldx;
  double* crow0 = A.c + b * (int64_t)N * ldx;
  const bool cn = (A.method == PNP_METHOD_CRANK_NICOLSON);
  const double vw = A.pb[b * 4 + 0], vb = A.pb[b * 4 + 1], gw = A.pb[b * 4 + 2], gb = A.pb[b * 4 + 3];
  const double vz = A.vzeta[b];
  double chk = 0.0, mn = 0.0;
  for (int i = tid; i < 3 * RB + xch_doubles<WY>() + WY * 256 + WY + 2; i += 64 * WY) lds[i] = 0.0;
  wg_sync<2>();
  const int ls2 = pidx<P>(2 * tid);

  for (int step = 0; step < A.nsteps; ++step) {
    const bool last_step = step + 1 == A.nsteps;
    // ---- 1. lagged potential ---------------------------------------------------------------------
    double v1 = 0.0;
    if (A.use_mig) {
      const __amdgpu_buffer_rsrc_t rl = row_rsrc(lin, ldx);
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const d2 t = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rl, tid * 16 + it * (1024 * WY), 0, 0));
        LV[pair_slot<P>(ls2, WY * it)] = t.x;
        LV[pair_slot<P>(ls2, WY * it) + PAIR_STEP<P>] = t.y;
      }
      wg_sync<2>();
      v1 = poisson_wg<P, WY, false, 1>(A, LV, GV, nullptr, strip_w, scal, vw, vb, gw, gb, lane, wave);
      if (tid == 0 && cn) GV[pidx<P>(nx - 2 + 1)] = GV[pidx<P>(nx - 1 + 1)];
      wg_sync<2>();
    }
    d2 accp[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) accp[it] = (d2)(0.0);

    // ---- 2. species, one at a time, all waves together ------------------------------------------------
    for (int k = 0; k < N; ++k) {
      double* grow = crow0 + (int64_t)k * ldx;
      const __amdgpu_buffer_rsrc_t rs = row_rsrc(grow, ldx);
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const d2 t = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16 + it * (1024 * WY), 0, 0));
        ROW[pair_slot<P>(ls2, WY * it)] = t.x;
        ROW[pair_slot<P>(ls2, WY * it) + PAIR_STEP<P>] = t.y;
      }
      wg_sync<2>();
      const SpecConst& S = A.spec[k];
      const double flux = A.flux[b * N + k];
      const double cL = A.cbulk[b * N + k];
      const double c1 = ROW[pidx<P>(1)], c0old = ROW[pidx<P>(0)], cLold = ROW[pidx<P>(nx - 1)];
      const double aa = S.mu * (v1 - vz);
      double c0new;
      if (cn) {
        const double rden = fast_rcp2(-S.twoD + aa);
        c0new = (-S.twoD - aa) * rden * c1 - 2 * flux * dx * rden;
      } else {
        c0new = ((S.twoD + aa) * c1 + flux * 2. * dx) * fast_rcp2(S.twoD - aa);
      }
      wg_sync<2>();
      if (tid == 0) {
        ROW[pidx<P>(0)] = cn ? (c0new + c0old) : c0new;
        ROW[pidx<P>(nx - 1)] = cn ? (cL + cLold) : cL;
      }
      wg_sync<2>();
      double x[P];
      if (cn) {
        const double hsr = S.hsr, e4r = S.e4r, eer = S.eer, omsr = S.omsr;
        double ta[P], tc[P], cc[P + 2], g4[P + 2];
#pragma unroll
        for (int t = 0; t < P + 2; ++t) cc[t] = ROW[pidx<P>(r0 + t)];
#pragma unroll
        for (int t = 0; t < P + 2; ++t) g4[t] = e4r * GV[pidx<P>(r0 + t)];
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const double lq = LV[pidx<P>(r0 + j)];
          const double left = cc[j] * (hsr + g4[j]);
          const double right = cc[j + 2] * (hsr - g4[j + 2]);
          x[j] = left + cc[j + 1] * (omsr + eer * lq) + right;
          ta[j] = -hsr + g4[j + 1];
          tc[j] = (r0 + j == m - 1) ? 0.0 : (-hsr - g4[j + 1]);
        }
        ta[0] = (tid == 0) ? 0.0 : ta[0];
        wg_sync<2>();
#ifndef MW_NOSOLVE      // (diagnosis build, tools/probe/mw_ceiling.sh: the memory-side ceiling of this kernel's access pattern; results are wrong)
        tridiag_wg<P, WY>(ta, tc, x, XCH, tid);
#endif
      } else {
        const double sf = S.sf, dm = S.dm, Mf = S.Mf;
        double cc[P + 2], gq[P + 2];
#pragma unroll
        for (int t = 0; t < P + 2; ++t) cc[t] = ROW[pidx<P>(r0 + t)];
#pragma unroll
        for (int t = 0; t < P + 2; ++t) gq[t] = A.use_mig ? GV[pidx<P>(r0 + t + 1)] : 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
          double Wt = sf - dm * gq[j + 2] + 0.5;
          double Et = sf + dm * gq[j] + 0.5;
          if (!A.lf) {
            Wt -= 0.5;
            Et -= 0.5;
          }
          double val = Et * cc[j] + Mf * cc[j + 1] + Wt * cc[j + 2];
          if (A.has_rates) val += A.rates[(b * N + k) * (int64_t)ldx + min(r0 + j + 1, nx - 1)] * dt;
          x[j] = val;
        }
        wg_sync<2>();
      }
#pragma unroll
      for (int j = 0; j < P; ++j) ROW[pidx<P>(r0 + j + 1)] = x[j];
      wg_sync<2>();
      if (tid == 0) {
        ROW[pidx<P>(0)] = c0new;
        ROW[pidx<P>(nx - 1)] = cL;
      }
      if (tid < ldx - nx) ROW[pidx<P>(nx + tid)] = 0.0;
      wg_sync<2>();
      const double qe = S.qe;
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int sl = min(pair_slot<P>(ls2, WY * it), RB - 4);
        d2 t;
        t.x = ROW[sl];
        t.y = ROW[sl + PAIR_STEP<P>];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, t), rs, tid * 16 + it * (1024 * WY), 0, 0);
        accp[it].x = __builtin_fma(-t.x, qe, accp[it].x);
        accp[it].y = __builtin_fma(-t.y, qe, accp[it].y);
        if (last_step) {
          const bool inrow = 2 * tid + 128 * WY * it < ldx;
          const double sx = inrow ? t.x : 0.0, sy = inrow ? t.y : 0.0;
          chk += (sx - sx) + (sy - sy);
          mn = fmin(mn, fmin(sx, sy));
        }
      }
      wg_sync<2>();
    }
    // ---- 3. charge row of the new state --------------------------------------------------------------
    {
      const __amdgpu_buffer_rsrc_t rs = row_rsrc(lout, ldx);
#pragma unroll
      for (int it = 0; it < IT; ++it)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, accp[it]), rs, tid * 16 + it * (1024 * WY), 0, 0);
    }
    double* tmp = lin;
    lin = lout;
    lout = tmp;
    if (step + 1 < A.nsteps) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
  const unsigned long long nan_mask = __ballot(chk != chk);
  const unsigned long long neg_mask = __ballot(mn < 0.0);
  if (lane == 0) {
    int st = PNP_STATUS_OK;
    if (neg_mask) st = PNP_STATUS_NEGATIVE;
    if (nan_mask) st = PNP_STATUS_NAN;
    if (st) atomicMax(&A.status[b], st);
  }
}

// read-back of v / grad_v for the multi-wave grid sizes
template <int P, int WY>
__global__ __launch_bounds__(64 * WY, 1) void poisson_kernel_mw(const DevArgs A, const double* __restrict__ lapl,
                                                               double* __restrict__ v, double* __restrict__ gradv) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int RB = rowbuf_mw_doubles<P, WY>();
  constexpr int IT = P / 2 + 1;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = blockIdx.x;
  const int ldx = A.ldx, nx = A.nx;
  double* LV = lds;
  double* GV = lds + RB;
  double* VV = lds + 2 * RB;
  double* XCH = lds + 3 * RB;
  double* strip_w = XCH + xch_doubles<WY>() + wave * 256;
  double* scal = XCH + xch_doubles<WY>() + WY * 256;
  for (int i = tid; i < 3 * RB + xch_doubles<WY>() + WY * 256 + WY + 2; i += 64 * WY) lds[i] = 0.0;
  wg_sync<2>();
  const int ls2 = pidx<P>(2 * tid);
  const __amdgpu_buffer_rsrc_t rl = row_rsrc(lapl + b * (int64_t)ldx, ldx);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const d2 t = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rl, tid * 16 + it * (1024 * WY), 0, 0));
    LV[pair_slot<P>(ls2, WY * it)] = t.x;
    LV[pair_slot<P>(ls2, WY * it) + PAIR_STEP<P>] = t.y;
  }
  wg_sync<2>();
  poisson_wg<P, WY, true, 0>(A, LV, GV, VV, strip_w, scal, A.pb[b * 4 + 0], A.pb[b * 4 + 1], A.pb[b * 4 + 2], A.pb[b * 4 + 3],
                             lane, wave);
  if (tid < ldx - nx) {
    VV[pidx<P>(nx + tid)] = 0.0;
    GV[pidx<P>(nx + tid)] = 0.0;
  }
  wg_sync<2>();
  const __amdgpu_buffer_rsrc_t rv = row_rsrc(v + b * (int64_t)ldx, ldx), rg = row_rsrc(gradv + b * (int64_t)ldx, ldx);
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int sl = min(pair_slot<P>(ls2, WY * it), RB - 4);
    d2 t;
    t.x = VV[sl];
    t.y = VV[sl + PAIR_STEP<P>];
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, t), rv, tid * 16 + it * (1024 * WY), 0, 0);
    t.x = GV[sl];
    t.y = GV[sl + PAIR_STEP<P>];
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, t), rg, tid * 16 + it * (1024 * WY), 0, 0);
  }
}

// stand-alone Poisson (read-back of tp.potential / tp.efield): one wave per lane of the batch
template <int P>
__global__ __launch_bounds__(64) void poisson_kernel(const DevArgs A, const double* __restrict__ lapl,
                                                      double* __restrict__ v, double* __restrict__ gradv) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int RB = rowbuf_doubles<P>();
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int ldx = A.ldx;
  double* VV = lds;
  double* GV = lds + RB;
  double* LV = lds + 2 * RB;
  // zero the pads so that the pitch tail written back is deterministic
  for (int i = lane; i < 4 * RB; i += 64) lds[i] = 0.0;
  lds_sync();
  load_row<P>(lapl + b * (int64_t)ldx, LV, ldx, lane);
  lds_sync();
  poisson_wave<P, true, 0>(A, LV, GV, VV, lds + 3 * RB, A.pb[b * 4 + 0], A.pb[b * 4 + 1], A.pb[b * 4 + 2], A.pb[b * 4 + 3], lane);
  // padded rows left don't-care values past the row: blank the pitch tail [nx, ldx)
  if (lane < ldx - A.nx) {
    VV[pidx<P>(A.nx + lane)] = 0.0;
    GV[pidx<P>(A.nx + lane)] = 0.0;
  }
  lds_sync();
  store_row<P>(v + b * (int64_t)ldx, VV, ldx, lane);
  store_row<P>(gradv + b * (int64_t)ldx, GV, ldx, lane);
}

// ------------------------------------------------------------------------------------------------
// Method-of-lines right-hand side, ode_func (calculator_old.py:827-935): dydt[b][k][i] for a batch of
// states y[b][k][i].  One wave per operating point, rows staged through LDS like step_kernel; every
// Poisson boundary combination is supported.  LDS = 3 padded rows (LV | GV | ROW).
// ------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(64) void mol_rhs_kernel(const DevArgs A, const double* __restrict__ y,
                                                      double* __restrict__ dydt) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int RB = rowbuf_doubles<P>();
  constexpr int IT = RowRegs<P>::IT;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int nx = A.nx, m = A.m, ldx = A.ldx, N = A.N;
  double* LV = lds;
  double* GV = lds + RB;
  double* ROW = lds + 2 * RB;
  const int r0 = lane * P;
  const double dx = A.dx, dt = A.dt;
  const double* yrow0 = y + b * (int64_t)N * ldx;
  double* orow0 = dydt + b * (int64_t)N * ldx;
  for (int i = lane; i < 3 * RB; i += 64) lds[i] = 0.0;
  lds_sync();
  const int ls = pidx<P>(2 * lane);
  if (A.use_mig) {
    // charge row of the state (:767-771), accumulated in species order at the coalesced positions
    d2 acc[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) acc[it] = (d2)(0.0);
    for (int k = 0; k < N; ++k) {
      RowRegs<P> rr;
      load_row_issue<P>(row_rsrc(yrow0 + (int64_t)k * ldx, ldx), rr, lane);
      const double qe = A.spec[k].qe;
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        acc[it].x = __builtin_fma(-rr.t[it].x, qe, acc[it].x);
        acc[it].y = __builtin_fma(-rr.t[it].y, qe, acc[it].y);
      }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      if (2 * lane + 128 * it < ldx) {
        LV[pair_slot<P>(ls, it)] = acc[it].x;
        LV[pair_slot<P>(ls, it) + PAIR_STEP<P>] = acc[it].y;
      }
    }
    lds_sync();
    poisson_wave<P, false, 1>(A, LV, GV, nullptr, ROW, A.pb[b * 4 + 0], A.pb[b * 4 + 1], A.pb[b * 4 + 2], A.pb[b * 4 + 3], lane);
  }
  double gq[P + 2];   // grad_v at grid r0 .. r0+P+1 (slot = index+1)
#pragma unroll
  for (int t = 0; t < P + 2; ++t) gq[t] = A.use_mig ? GV[pidx<P>(r0 + t + 1)] : 0.0;
  const double g1 = A.use_mig ? GV[pidx<P>(1 + 1)] : 0.0;   // grad_v[1]
  lds_sync();
  for (int k = 0; k < N; ++k) {
    load_row<P>(yrow0 + (int64_t)k * ldx, ROW, ldx, lane);
    lds_sync();
    double cc[P + 2];
#pragma unroll
    for (int t = 0; t < P + 2; ++t) cc[t] = ROW[pidx<P>(r0 + t)];
    const double c0 = ROW[pidx<P>(0)], c1 = ROW[pidx<P>(1)], c2 = ROW[pidx<P>(2)];
    lds_sync();
    const SpecConst& S = A.spec[k];
    const double flux = A.flux[b * N + k];
    const double bq = A.beta * S.q;
    double out[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {   // grid i = r0+j+1, :916-927
      const double d2c = (cc[j + 2] - 2 * cc[j + 1] + cc[j]) / (dx * dx);
      const double dcg = (cc[j + 2] * gq[j + 2] - cc[j] * gq[j]) / (2. * dx);
      const double corr = A.lf ? d2c * (dx * dx) / dt / 2. : 0.0;
      double v = corr + S.D * (d2c + bq * dcg);
      if (A.has_rates) v += A.rates[(b * N + k) * (int64_t)ldx + min(r0 + j + 1, nx - 1)];
      out[j] = v;
    }
    // wall cell :897-915 (no rate term), bulk point :886
    const double corr0 = A.lf ? (c1 - c0) / dt : 0.0;
    const double w0 = corr0 + (S.D * ((c2 - c0) / (2. * dx) + bq * c1 * g1) - flux) / dx;
#pragma unroll
    for (int j = 0; j < P; ++j) ROW[pidx<P>(r0 + j + 1)] = out[j];
    lds_sync();
    if (lane == 0) {
      ROW[pidx<P>(0)] = w0;
      ROW[pidx<P>(nx - 1)] = 0.0;
    }
    if (lane < ldx - nx) ROW[pidx<P>(nx + lane)] = 0.0;
    lds_sync();
    store_row<P>(orow0 + (int64_t)k * ldx, ROW, ldx, lane);
    lds_sync();
  }
  (void)m;
}

// ode_func for grids beyond one wave (nx > 1026): the same right-hand side evaluated point by point from a gradient row that
// launch_poisson produced (any grid length, any Poisson branch).  One thread per (lane, species, grid point).
__global__ void mol_rhs_pointwise_kernel(const DevArgs A, const double* __restrict__ y, const double* __restrict__ gradv,
                                         double* __restrict__ dydt) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = A.B * A.N * (int64_t)A.ldx;
  if (idx >= total) return;
  const int ldx = A.ldx, nx = A.nx, N = A.N;
  const int i = (int)(idx % ldx);
  const int64_t bk = idx / ldx;
  const int k = (int)(bk % N);
  const int64_t b = bk / N;
  const double* c = y + bk * (int64_t)ldx;
  const double* g = gradv + b * (int64_t)ldx;
  const double dx = A.dx, dt = A.dt;
  const SpecConst& S = A.spec[k];
  const double bq = A.beta * S.q;
  double out = 0.0;                      // bulk point (:886) and the pitch tail
  if (i == 0) {                          // wall cell :897-915 (no rate term)
    const double g1 = A.use_mig ? g[1] : 0.0;
    const double corr0 = A.lf ? (c[1] - c[0]) / dt : 0.0;
    out = corr0 + (S.D * ((c[2] - c[0]) / (2. * dx) + bq * c[1] * g1) - A.flux[b * N + k]) / dx;
  } else if (i < nx - 1) {               // :916-927
    const double gp = A.use_mig ? g[i + 1] : 0.0, gm = A.use_mig ? g[i - 1] : 0.0;
    const double d2c = (c[i + 1] - 2 * c[i] + c[i - 1]) / (dx * dx);
    const double dcg = (c[i + 1] * gp - c[i - 1] * gm) / (2. * dx);
    const double corr = A.lf ? d2c * (dx * dx) / dt / 2. : 0.0;
    out = corr + S.D * (d2c + bq * dcg);
    if (A.has_rates) out += A.rates[bk * (int64_t)ldx + i];
  }
  dydt[idx] = out;
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
    for (int k = 0; k < A.N; ++k) acc = __builtin_fma(-A.c[(b * A.N + k) * (int64_t)A.ldx + i], A.spec[k].qe, acc);
  }
  lapl[idx] = acc;
}

// Upload path of pnp_set_batch: the host state arrives contiguous ([B][N][nx]) in a staging buffer and THIS kernel writes the
// pitched state rows (pads zero), the bulk Dirichlet values (last grid point, calculator_old.py:540) and a zeroed second charge row.
// One contiguous host-to-device copy and one kernel replace two fill blits and two strided 2-D copies.
__global__ void unpack_state_kernel(const DevArgs A, const double* __restrict__ stage, double* __restrict__ cbulk,
                                    double* __restrict__ lapl_zero, int32_t* __restrict__ iters_zero) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t total = A.B * A.N * A.ldx;
  if (idx >= total) return;
  const int64_t row = idx / A.ldx;           // b*N + k
  const int i = (int)(idx - row * A.ldx);
  double v = 0.0;
  if (i < A.nx) {
    v = stage[row * A.nx + i];
    if (i == A.nx - 1) cbulk[row] = v;
  }
  A.c[idx] = v;
  if (lapl_zero && row % A.N == 0) lapl_zero[(row / A.N) * A.ldx + i] = 0.0;
  if (i == 0 && row % A.N == 0) {              // per-lane flags start clean
    A.status[row / A.N] = PNP_STATUS_OK;
    if (iters_zero) iters_zero[row / A.N] = 0;
  }
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
  if (m <= 64 * 16 * 4) return 16;   // several waves per system (waves_per_system)
  return 0;
}

int waves_per_system(int nx) {
  const int m = nx - 2;
  if (m <= 64 * 16) return 1;
  return m <= 64 * 16 * 2 ? 2 : 4;
}

template <int WY>
static size_t mw_lds_bytes() {
  return (size_t)(3 * rowbuf_mw_doubles<16, WY>() + xch_doubles<WY>() + WY * 256 + WY + 2) * sizeof(double);
}

hipError_t launch_step_mw(const DevArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)a.B);
  static bool attr_set = false;
  if (!attr_set) {   // more than the default 64 KiB of dynamic LDS
    (void)hipFuncSetAttribute((const void*)step_kernel_mw<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mw_lds_bytes<2>());
    (void)hipFuncSetAttribute((const void*)step_kernel_mw<16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mw_lds_bytes<4>());
    (void)hipFuncSetAttribute((const void*)poisson_kernel_mw<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mw_lds_bytes<2>());
    (void)hipFuncSetAttribute((const void*)poisson_kernel_mw<16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mw_lds_bytes<4>());
    attr_set = true;
  }
  if (waves_per_system(a.nx) == 2) hipLaunchKernelGGL((step_kernel_mw<16, 2>), grid, dim3(128), mw_lds_bytes<2>(), stream, a);
  else hipLaunchKernelGGL((step_kernel_mw<16, 4>), grid, dim3(256), mw_lds_bytes<4>(), stream, a);
  return hipGetLastError();
}

hipError_t launch_poisson_mw(const DevArgs& a, const double* lapl, double* v, double* gradv, hipStream_t stream) {
  const dim3 grid((unsigned)a.B);
  (void)launch_step_mw;   // attributes are set by the first step launch; set them here too for read-back-first use
  (void)hipFuncSetAttribute((const void*)poisson_kernel_mw<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mw_lds_bytes<2>());
  (void)hipFuncSetAttribute((const void*)poisson_kernel_mw<16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mw_lds_bytes<4>());
  if (waves_per_system(a.nx) == 2) hipLaunchKernelGGL((poisson_kernel_mw<16, 2>), grid, dim3(128), mw_lds_bytes<2>(), stream, a, lapl, v, gradv);
  else hipLaunchKernelGGL((poisson_kernel_mw<16, 4>), grid, dim3(256), mw_lds_bytes<4>(), stream, a, lapl, v, gradv);
  return hipGetLastError();
}

void choose_step_config(int N, int64_t B, int P, bool fused, int* W, int* G) {
  int w = 1, g = 1;
  if (fused) {
    // Launches that advance many timesteps (measured per shape with tools/probe/step_config_probe.py):
    // up to 8 points per lane one species per wave and 3 (N <= 4: N) waves per operating point -- 0.75-0.78 of the roofline
    // on the headline shape against 0.65 for three species interleaved in one wave, at every batch size, because a wave
    // issues an fp64 instruction only every ~8.5 cycles and the SIMD wants four of them; 16 points per lane: one wave, two
    // species interleaved when they pair up (the rows no longer fit LDS three waves wide at useful occupancy).
    if (P >= 16) g = (N % 2 == 0) ? 2 : 1;
    else if (P >= 4) w = N <= 4 ? N : 3;
  } else {
    // One timestep per launch.  Small batches: every SIMD should own a few independent dependency chains -- species in
    // different waves (W) and/or interleaved inside a wave (G).  Large batches: one wave per operating point.
    if (P <= 8) g = (N % 3 == 0) ? 3 : ((N % 2 == 0) ? 2 : 1);
    if (g > N) g = N;
    const int wmax = (g == 3) ? 1 : (g == 2 ? 2 : 4);   // instantiated (W,G) pairs: see launch_step_p
    while (w < wmax && w * g < N && B * w < 2048) ++w;
    if (N == 3 && P == 8 && B < 2048) {   // measured (PROBE_SPL=1 tools/probe/step_config_probe.py): 0.34 against 0.32
      w = 3;
      g = 1;
    }
  }
  *W = w;
  *G = g;
}

template <int P>
static constexpr size_t lds_bytes_for(int rows) {
  return (size_t)(2 + rows) * rowbuf_doubles<P>() * sizeof(double);
}

size_t step_lds_bytes(int P, int W, int G) {
  switch (P) {
    case 1: return lds_bytes_for<1>(W * G);
    case 2: return lds_bytes_for<2>(W * G);
    case 4: return lds_bytes_for<4>(W * G);
    case 8: return lds_bytes_for<8>(W * G);
    case 16: return lds_bytes_for<16>(W * G);
    default: return 0;
  }
}

template <int P>
static hipError_t launch_step_p(const DevArgs& a, int W, int G, hipStream_t stream) {
  const dim3 grid((unsigned)a.B);
  const size_t lds = lds_bytes_for<P>(W * G);
  const int key = W * 10 + G;
  switch (key) {
    case 11: hipLaunchKernelGGL((step_kernel<P, 1, 1>), grid, dim3(64), lds, stream, a); break;
    case 12: hipLaunchKernelGGL((step_kernel<P, 1, 2>), grid, dim3(64), lds, stream, a); break;
    case 13: hipLaunchKernelGGL((step_kernel<P, 1, 3>), grid, dim3(64), lds, stream, a); break;
    case 21: hipLaunchKernelGGL((step_kernel<P, 2, 1>), grid, dim3(128), lds, stream, a); break;
    case 22: hipLaunchKernelGGL((step_kernel<P, 2, 2>), grid, dim3(128), lds, stream, a); break;
    case 31: hipLaunchKernelGGL((step_kernel<P, 3, 1>), grid, dim3(192), lds, stream, a); break;
    case 41: hipLaunchKernelGGL((step_kernel<P, 4, 1>), grid, dim3(256), lds, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

bool step_config_supported(int W, int G) {
  const int key = W * 10 + G;
  return key == 11 || key == 12 || key == 13 || key == 21 || key == 22 || key == 31 || key == 41;
}

// register-resident kernel: Dirichlet/Dirichlet Poisson, even P, no FTCS rate term
bool step_rr_applicable(const DevArgs& a) {
  const int P = points_per_lane(a.nx);
  return a.pb_mode == PNP_PB_DD && P >= 2 && !a.has_rates && a.nx >= 6;
}

template <int P>
static hipError_t launch_step_rr_p(const DevArgs& a, int W, hipStream_t stream) {
  const dim3 grid((unsigned)a.B);
  const size_t lds = (size_t)(W * 384 + 2 * W * (64 * P + 2)) * sizeof(double);
  const bool cn = a.method == PNP_METHOD_CRANK_NICOLSON;
  switch (W * 2 + (cn ? 1 : 0)) {
    case 3: hipLaunchKernelGGL((step_kernel_rr<P, 1, true>), grid, dim3(64), lds, stream, a); break;
    case 5: hipLaunchKernelGGL((step_kernel_rr<P, 2, true>), grid, dim3(128), lds, stream, a); break;
    case 7: hipLaunchKernelGGL((step_kernel_rr<P, 3, true>), grid, dim3(192), lds, stream, a); break;
    case 9: hipLaunchKernelGGL((step_kernel_rr<P, 4, true>), grid, dim3(256), lds, stream, a); break;
    case 2: hipLaunchKernelGGL((step_kernel_rr<P, 1, false>), grid, dim3(64), lds, stream, a); break;
    case 4: hipLaunchKernelGGL((step_kernel_rr<P, 2, false>), grid, dim3(128), lds, stream, a); break;
    case 6: hipLaunchKernelGGL((step_kernel_rr<P, 3, false>), grid, dim3(192), lds, stream, a); break;
    case 8: hipLaunchKernelGGL((step_kernel_rr<P, 4, false>), grid, dim3(256), lds, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_step_rr(const DevArgs& a, int W, hipStream_t stream) {
  switch (points_per_lane(a.nx)) {
    case 2: return launch_step_rr_p<2>(a, W, stream);
    case 4: return launch_step_rr_p<4>(a, W, stream);
    case 8: return launch_step_rr_p<8>(a, W, stream);
    case 16: return launch_step_rr_p<16>(a, W, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_step(const DevArgs& a, int W, int G, hipStream_t stream) {
  switch (points_per_lane(a.nx)) {
    case 1: return launch_step_p<1>(a, W, G, stream);
    case 2: return launch_step_p<2>(a, W, G, stream);
    case 4: return launch_step_p<4>(a, W, G, stream);
    case 8: return launch_step_p<8>(a, W, G, stream);
    case 16: return launch_step_p<16>(a, W, G, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_poisson(const DevArgs& a, const double* lapl, double* v, double* gradv, hipStream_t stream) {
  if (waves_per_system(a.nx) > 1) return launch_poisson_mw(a, lapl, v, gradv, stream);
  const int P = points_per_lane(a.nx);
  const dim3 grid((unsigned)a.B), block(64);
  switch (P) {
    case 1: hipLaunchKernelGGL(poisson_kernel<1>, grid, block, lds_bytes_for<1>(2), stream, a, lapl, v, gradv); break;
    case 2: hipLaunchKernelGGL(poisson_kernel<2>, grid, block, lds_bytes_for<2>(2), stream, a, lapl, v, gradv); break;
    case 4: hipLaunchKernelGGL(poisson_kernel<4>, grid, block, lds_bytes_for<4>(2), stream, a, lapl, v, gradv); break;
    case 8: hipLaunchKernelGGL(poisson_kernel<8>, grid, block, lds_bytes_for<8>(2), stream, a, lapl, v, gradv); break;
    case 16: hipLaunchKernelGGL(poisson_kernel<16>, grid, block, lds_bytes_for<16>(2), stream, a, lapl, v, gradv); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_mol_rhs_pointwise(const DevArgs& a, const double* y, const double* gradv, double* dydt, hipStream_t stream) {
  const int64_t total = a.B * a.N * (int64_t)a.ldx;
  const int threads = 256;
  hipLaunchKernelGGL(mol_rhs_pointwise_kernel, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0, stream, a, y, gradv, dydt);
  return hipGetLastError();
}

hipError_t launch_mol_rhs(const DevArgs& a, const double* y, double* dydt, hipStream_t stream) {
  const int P = points_per_lane(a.nx);
  const dim3 grid((unsigned)a.B), block(64);
  switch (P) {
    case 1: hipLaunchKernelGGL(mol_rhs_kernel<1>, grid, block, lds_bytes_for<1>(1), stream, a, y, dydt); break;
    case 2: hipLaunchKernelGGL(mol_rhs_kernel<2>, grid, block, lds_bytes_for<2>(1), stream, a, y, dydt); break;
    case 4: hipLaunchKernelGGL(mol_rhs_kernel<4>, grid, block, lds_bytes_for<4>(1), stream, a, y, dydt); break;
    case 8: hipLaunchKernelGGL(mol_rhs_kernel<8>, grid, block, lds_bytes_for<8>(1), stream, a, y, dydt); break;
    case 16: hipLaunchKernelGGL(mol_rhs_kernel<16>, grid, block, lds_bytes_for<16>(1), stream, a, y, dydt); break;
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

hipError_t launch_unpack_state(const DevArgs& a, const double* stage, double* cbulk, double* lapl_zero, int32_t* iters_zero,
                               hipStream_t stream) {
  const int64_t total = a.B * a.N * a.ldx;
  const int threads = 256;
  const unsigned blocks = (unsigned)((total + threads - 1) / threads);
  hipLaunchKernelGGL(unpack_state_kernel, dim3(blocks), dim3(threads), 0, stream, a, stage, cbulk, lapl_zero, iters_zero);
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
