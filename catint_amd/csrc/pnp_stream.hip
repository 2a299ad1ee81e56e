// Streaming timestep kernel for batches whose state does not stay on chip (one launch per timestep over hundreds of MB
// to GB of state: every row comes from HBM and goes back to HBM -- the regime where the HBM roofline is the real bound).
//
// step_kernel_st<P,CN,GL>: the register-resident mapping of step_kernel_rr (one 1D grid per wavefront, lane l owns P
// consecutive interior unknowns, window loads straight into registers, DPP scans, cyclic-reduction strips in LDS), made
//   * PERSISTENT: a workgroup is one wave; the grid holds as many waves as the chip keeps resident and wave w walks the
//     operating points w, w + grid, w + 2 grid, ... -- no launch/teardown bubble per operating point;
//   * SOFTWARE-PIPELINED: the window of the next species row (at the last species: the charge row and the first species row of
//     the wave's NEXT operating point) is requested right after the stencil assembly has consumed the current window, so the
//     HBM latency of every row hides behind the tridiagonal solve of the row before it.  The destination registers are the ones
//     the assembly has just freed (the concentration window, the lagged charge window), so the prefetch costs no registers;
//   * GL = true keeps the lagged charge row and the potential gradient of the step in LDS instead of registers (transposed,
//     conflict-free): 16 points per lane (nx up to 1026) then fit two waves per SIMD instead of one.
// Arithmetic is step_kernel_rr's statement for statement (same results bit for bit); reference: catint/calculator_old.py
// :512-558 (Crank-Nicolson), :990-1023 (FTCS), Poisson :716-730, :780-786.
#include "pnp_internal.h"
#include "pnp_wave.h"

namespace pnp {

// L2 touches two rows ahead (one dword per 64 bytes): measured NEGATIVE on MI355X (0.518 -> 0.482 plain, 0.501 with sc1 loads at
// N = 3, nx = 512, B = 32768: the kernel is not waiting for HBM latency but for HBM bandwidth) -- kept behind a switch, off.
#ifndef ST_TOUCH
#define ST_TOUCH 0
#endif
// Issue priority of a wave by phase (s_setprio; waves of one SIMD are arbitrated by priority, then age).  Measured on one GPU's
// share of configs[3] / N = 3, nx = 512, B = 32768, one launch per step (tools/probe/stream_probe.py, gpurun_out/stream_probe_prio*.txt):
// (assembly, store) = (0,0) 0.551 / 0.552 -> (2,1) 0.584 / 0.575 -> (2,3) 0.601 / 0.592: the phases that talk to memory go first, the
// long tridiagonal solve fills the gaps.
#ifndef ST_PRIO_ASM
#define ST_PRIO_ASM 2
#endif
#ifndef ST_PRIO_STORE
#define ST_PRIO_STORE 3
#endif
// Cache policy of the row traffic (buffer instruction aux bits; 2 = nt).  The state is read once and overwritten in place per
// launch; tools/probe/access_pattern2.hip: an in-place streaming copy runs at 0.66 of 8 TB/s with the default policy and 0.70 with
// nt loads + nt stores (out of place: 0.68 either way).
#ifndef ST_AUX_LOAD
#define ST_AUX_LOAD 0
#endif
#ifndef ST_AUX_STORE
#define ST_AUX_STORE 0
#endif
#ifndef ST_TOUCH_AUX
#define ST_TOUCH_AUX 16   // sc1: served by L2, no allocation in the (32 KiB) vector L1
#endif

template <int P, int GL>
constexpr int step_st_min_waves() {
#ifndef ST_MINW8
#define ST_MINW8 3
#endif
#ifndef ST_MINW8GL
#define ST_MINW8GL 3
#endif
  return P <= 4 ? 4 : (P == 8 ? (GL ? ST_MINW8GL : ST_MINW8) : (GL ? 2 : 1));   // P = 16: two waves per SIMD need a row in LDS
}

template <int P>
constexpr int st_gl_doubles() {   // two transposed arrays [P][guard | 64 lanes | guard]
  return 2 * P * 66;
}

// LDS of one wave (= one workgroup): ROW | ACCS | (GL: GS | LS)
//   ROW   one padded row (pidx layout of the LDS-staged kernels): the new row of a species is transposed through it so that it
//         leaves as coalesced 16-byte-per-lane stores (1 KiB per instruction -> whole 64-byte L2 writes; the blocked register
//         layout stored directly makes every lane's 16 bytes an L2 write request of its own: 4x the requests, measured as the
//         limit of step_kernel_rr at large batch).  Its head doubles as the strip of the cyclic reduction.
//   ACCS  the charge row under construction, transposed [j][lane] (conflict-free, own entries only)
template <int P, int GL>
constexpr int st_lds_doubles() {
  return rowbuf_doubles<P>() + 64 * P + (GL == 1 ? st_gl_doubles<P>() : (GL == 2 ? st_gl_doubles<P>() / 2 : 0));
}

template <int P, bool CN, int GL>
__global__ __launch_bounds__(64, (step_st_min_waves<P, GL>())) void step_kernel_st(const DevArgs A) {
  static_assert(P >= 2 && P % 2 == 0, "window loads need an even P");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int RB = rowbuf_doubles<P>();
  constexpr int XS = RB;                       // distance between strips (one system per call)
  double* ROW = lds;
  double* strip = lds;                         // cyclic-reduction strip: the head of ROW
  double* ACCS = lds + RB;                     // ACCS[j*64 + lane] = charge row entry of grid lane*P + j + 1
  double* GS = ACCS + 64 * P;                  // GL: grad_v own rows,  GS[j*66 + 1 + lane] = grad_v[lane*P + j + 1]
  double* LS = GS + P * 66;                    // GL: lapl_v own rows,  LS[j*66 + 1 + lane] = lapl_v[lane*P + j + 1]
  const int lane = threadIdx.x;
  const int nx = A.nx, m = A.m, ldx = A.ldx, N = A.N;
  const int r0 = lane * P;
  const double dx = A.dx;
  constexpr bool cn = CN;
  const int64_t stride = gridDim.x;
  int64_t bi = blockIdx.x;
  if (bi >= A.B) return;
  int64_t b = A.reverse ? A.B - 1 - bi : bi;

  int pf0 = 0, pf1 = 0;    // L2 touches in flight (see the species loop)
  double lw[P + 2];        // lagged charge row window: lapl_v[r0 + t]
  double cw[P + 2];        // concentration window in flight: C[k][r0 + t]
  // prologue: the first operating point's charge row and first species row
  if (A.use_mig) load_window<P, ST_AUX_LOAD>(row_rsrc(A.lapl_a + b * (int64_t)ldx, ldx), lw, lane);
  load_window<P, ST_AUX_LOAD>(row_rsrc(A.c + b * (int64_t)N * ldx, ldx), cw, lane);

  for (; bi < A.B; bi += stride) {
    b = A.reverse ? A.B - 1 - bi : bi;
    const int64_t bin = (bi + stride < A.B) ? bi + stride : bi;  // next operating point of this wave (the last one re-reads itself)
    const int64_t bn = A.reverse ? A.B - 1 - bin : bin;
    double* lin = A.lapl_a + b * (int64_t)ldx;
    double* lout = A.lapl_b + b * (int64_t)ldx;
    double* crow0 = A.c + b * (int64_t)N * ldx;
    const double vw = A.pb[b * 4 + 0], vb = A.pb[b * 4 + 1];
    const double vz = A.vzeta[b];
    double chk = 0.0, mn = 0.0;
    for (int step = 0; step < A.nsteps; ++step) {
      const bool last_step = step + 1 == A.nsteps;
      // (lane index opaque once per step, as in the species loop below: the ~40 row predicates of the scan and the window offsets
      // are recomputed here instead of living in scalar registers -- spilled through v_writelane / v_readlane -- or spilled
      // vector registers across the whole kernel)
      int lane_p = lane;
      asm volatile("" : "+v"(lane_p));
      if (step > 0) {
        // fused launches: the next step re-reads rows this wave has just written; the charge window came over in registers
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        load_window<P, ST_AUX_LOAD>(row_rsrc(crow0, ldx), cw, lane_p);
      }
      // ---- 1. lagged potential: v'' = lapl, v[0] = vw, v[nx-1] = vb  (calculator_old.py:716-730, :780-786) ----
      double gx[GL ? 1 : P + 3];   // grad_v[r0 - 1 + t]
      double v1 = 0.0;
      double g_last = 0.0;
      const int r0 = lane_p * P;
      if (A.use_mig) {
        double Hi[P];
        const double dx2 = A.dx2;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
          double h = lw[j + 1] * dx2;                       // grid r0+j+1; pads and out-of-range are 0
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
        v1 = vw + w0;
        const double inv2dx = A.inv2dx;
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
        g_last = gm1 + (gm1 - gm2);
        if constexpr (GL != 0) {
          lds_sync();                                       // the previous step's readers of GS / LS are done
#pragma unroll
          for (int j = 0; j < P; ++j) {
            GS[j * 66 + 1 + lane_p] = gown[j];
            if constexpr (GL == 1) LS[j * 66 + 1 + lane_p] = lw[j + 1];
          }
          if (lane_p == 0) {                                  // what "lane_p -1" would hold: grad_v[0] twice (:496), lapl_v[0]
            GS[(P - 2) * 66] = g_first;
            GS[(P - 1) * 66] = g_first;
            if constexpr (GL == 1) LS[(P - 1) * 66] = lw[0];
            GS[65] = 0.0;                                   // "lane_p 64", row 0 (FTCS window of the last lane_p)
          }
          lds_sync();
        } else {
#pragma unroll
          for (int t = 2; t < P + 2; ++t) gx[t] = gown[t - 2];
          gx[1] = from_prev_lane(g_first, gown[P - 1]);       // grad_v[r0]    (lane_p 0: grad_v[0])
          gx[0] = from_prev_lane(g_first, gown[P - 2]);       // grad_v[r0-1]  (lane_p 0: index -1 -> grad_v[0], :496)
          gx[P + 2] = from_next_lane(0.0, gown[0]);           // grad_v[r0+P+1]
          // the bulk boundary term uses grad_v[-1] (:498): CN reads it at interior index nx-2, FTCS at grid nx-1
#pragma unroll
          for (int t = 2; t < P + 3; ++t) {
            const int idx = r0 - 1 + t;                        // gx[t] = grad_v[idx]
            gx[t] = ((cn && idx == nx - 2) || idx == nx - 1) ? g_last : gx[t];
          }
        }
      } else {
        if constexpr (GL != 0) {
          lds_sync();
#pragma unroll
          for (int j = 0; j < P; ++j) {
            GS[j * 66 + 1 + lane_p] = 0.0;
            if constexpr (GL == 1) LS[j * 66 + 1 + lane_p] = 0.0;
          }
          if (lane_p == 0) {
            GS[(P - 2) * 66] = 0.0;
            GS[(P - 1) * 66] = 0.0;
            if constexpr (GL == 1) LS[(P - 1) * 66] = 0.0;
            GS[65] = 0.0;
          }
          lds_sync();
        } else {
#pragma unroll
          for (int t = 0; t < P + 3; ++t) gx[t] = 0.0;
        }
      }
      // grad_v[r0 - 1 + t] and lapl_v[r0 + j] as the stencil wants them
      auto grad_at = [&](int t, int r0o) -> double {
        if constexpr (GL != 0) {
          const int lo = r0o / P;      // (opaque) lane
          double g = (t == 0) ? GS[(P - 2) * 66 + lo] : (t == 1) ? GS[(P - 1) * 66 + lo]
                     : (t == P + 2) ? GS[2 + lo] : GS[(t - 2) * 66 + 1 + lo];
          if (t >= 2) {
            const int idx = r0o - 1 + t;
            g = ((cn && idx == nx - 2) || idx == nx - 1) ? g_last : g;
          }
          return g;
        } else {
          return gx[t];
        }
      };
      auto lapl_at = [&](int j, int lo) -> double {
        if constexpr (GL == 1) return (j == 0) ? LS[(P - 1) * 66 + lo] : LS[(j - 1) * 66 + 1 + lo];
        else return lw[j];
      };

      // the next charge row, accumulated species by species: own rows in LDS (ACCS), the two boundary entries in registers
      double acc0 = 0.0, accL = 0.0;

      // ---- 2. the species, one row at a time; the next row is on its way while this one is solved -------------------
      for (int k = 0; k < N; ++k) {
        const SpecConst& S = A.spec[k];
        const double flux = A.flux[b * N + k];
        const double cL = A.cbulk[b * N + k];                // C[k,-1] = C0[(k+1)*nx-1] (:540 / :1008) -- also COLD[k,-1]
        // The lane_o index is made opaque once per species: the row predicates below (first / last real row, padded rows) are
        // then recomputed where they are used (one v_cmp each) instead of being hoisted out of the species loop into ~60
        // scalar registers that spill through v_writelane / v_readlane.
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        // a wave that is about to request memory issues first on its SIMD: the stencil assembly and the window requests behind
        // it run at raised priority, the long tridiagonal solve at the default one
        __builtin_amdgcn_s_setprio(ST_PRIO_ASM);
        const int r0 = lane_o * P;
        double (&cc)[P + 2] = cw;                            // patched in place: the prefetch below overwrites it anyway
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
        const double pat0 = cn ? (c0new + c0old) : c0new;
        const double patL = cn ? (cL + cL) : cL;
        cc[0] = (lane_o == 0) ? pat0 : cc[0];
#pragma unroll
        for (int t = 2; t < P + 2; ++t) cc[t] = (r0 + t == nx - 1) ? patL : cc[t];
        double x[1][P];
        double ta[1][P], tc[1][P];
        if (cn) {
          // rows are divided by the constant diagonal 1+s up front (constants pre-scaled on the host)
          const double hsr = S.hsr, e4r = S.e4r, eer = S.eer, omsr = S.omsr;
          double gm = e4r * grad_at(0, r0), g0 = e4r * grad_at(1, r0);
#pragma unroll
          for (int j = 0; j < P; ++j) {
            // grad_v / lapl_v carry the INTERIOR index r (add_field :483-490); RHS = C[k,1:-1] . B1 (:553)
            const double gp = e4r * grad_at(j + 2, r0);
            const double left = cc[j] * (hsr + gm);
            const double right = cc[j + 2] * (hsr - gp);
            x[0][j] = left + cc[j + 1] * (omsr + eer * lapl_at(j, lane_o)) + right;
            ta[0][j] = -hsr + g0;                                          // A[r,r-1], :487
            tc[0][j] = (r0 + j == m - 1) ? 0.0 : (-hsr - g0);              // A[r,r+1], :490; none in the last real row
            gm = g0;
            g0 = gp;
          }
          ta[0][0] = (lane_o == 0) ? 0.0 : ta[0][0];                       // the first row has no sub-diagonal
        } else {
          const double sf = S.sf, dm = S.dm, Mf = S.Mf;
#pragma unroll
          for (int j = 0; j < P; ++j) {                                    // grid i = r0+j+1, :1012-1022
            double Wt = sf - dm * grad_at(j + 3, r0) + 0.5;                    // grad_v[i+1]
            double Et = sf + dm * grad_at(j + 1, r0) + 0.5;                    // grad_v[i-1]
            if (!A.lf) {
              Wt -= 0.5;
              Et -= 0.5;
            }
            x[0][j] = Et * cc[j] + Mf * cc[j + 1] + Wt * cc[j + 2];
          }
        }
        // ---- the windows are consumed: request the next ones into the registers they leave behind -----------------
        if (k + 1 < N) {
          load_window<P, ST_AUX_LOAD>(row_rsrc(crow0 + (int64_t)(k + 1) * ldx, ldx), cw, lane_o);
        } else if (last_step) {
          if (A.use_mig) load_window<P, ST_AUX_LOAD>(row_rsrc(A.lapl_a + bn * (int64_t)ldx, ldx), lw, lane_o);
          load_window<P, ST_AUX_LOAD>(row_rsrc(A.c + bn * (int64_t)N * ldx, ldx), cw, lane_o);
        }
        // ... and pull the row(s) AFTER those from HBM into L2 with one dword per 64 bytes (one instruction covers a 4 KiB
        // row): under load an HBM miss takes longer than one row's solve, so the window request above hits L2 only if the row
        // was asked for a whole row-time earlier.  The dwords are never used; they are kept until the end of the NEXT species'
        // iteration (vmcnt retires in order, so consuming them earlier would wait for the miss).
        const int tp0 = pf0, tp1 = pf1;
        if (ST_TOUCH) {
          const int toff = lane_o * 64;
          if (k + 2 < N) {
            pf0 = __builtin_amdgcn_raw_buffer_load_b32(row_rsrc(crow0 + (int64_t)(k + 2) * ldx, ldx), toff, 0, ST_TOUCH_AUX);
          } else if (last_step) {
            const double* nrow = A.c + bn * (int64_t)N * ldx;
            if (k + 2 == N) {
              if (A.use_mig) pf1 = __builtin_amdgcn_raw_buffer_load_b32(row_rsrc(A.lapl_a + bn * (int64_t)ldx, ldx), toff, 0, ST_TOUCH_AUX);
              pf0 = __builtin_amdgcn_raw_buffer_load_b32(row_rsrc(nrow, ldx), toff, 0, ST_TOUCH_AUX);
            } else if (N > 1) {
              pf0 = __builtin_amdgcn_raw_buffer_load_b32(row_rsrc(nrow + ldx, ldx), toff, 0, ST_TOUCH_AUX);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
#ifndef ST_NOCOMPUTE   // (diagnosis build: the memory-side ceiling of this access pattern, tools/probe: results are wrong)
        if (cn) tridiag_wave<P, 1>(ta, tc, x, strip, XS, lane_o);            // np.linalg.solve(A,B), :556
#endif
        // ---- results: charge contribution, status, then the row leaves through LDS as coalesced stores ----------------
        __builtin_amdgcn_s_setprio(ST_PRIO_STORE);
#pragma unroll
        for (int j = 0; j < P; ++j) {
          const double xv = (r0 + j < m) ? x[0][j] : 0.0;
          const double prev = (k == 0) ? 0.0 : ACCS[j * 64 + lane_o];
          ACCS[j * 64 + lane_o] = __builtin_fma(-xv, qe, prev);
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
#pragma unroll
        for (int j = 0; j < P; ++j) ROW[pidx<P>(r0 + j + 1)] = x[0][j];     // padded rows land past the row (don't care)
        lds_sync();
        if (lane_o == 0) {
          ROW[pidx<P>(0)] = c0new;
          ROW[pidx<P>(nx - 1)] = cL;
        }
        if (lane_o < ldx - nx) ROW[pidx<P>(nx + lane_o)] = 0.0;             // the pitch tail stays zero
        lds_sync();
        // one launch per step on 16 points per lane: the rows written are not read again before they have left every cache -> nt
        // (measured, one GPU's share of configs[3]: 0.597 -> 0.607 of the roofline per step; fused launches re-read them: 0.715 -> 0.664,
        // and 8 points per lane loses 1.5 %, so only here)
        if (GL == 2 && A.nsteps == 1) store_row<P, 2>(crow0 + (int64_t)k * ldx, ROW, ldx, lane_o);
        else store_row<P, ST_AUX_STORE>(crow0 + (int64_t)k * ldx, ROW, ldx, lane_o);
        lds_sync();
        if (ST_TOUCH) asm volatile("" ::"v"(tp0), "v"(tp1));   // the touches of the PREVIOUS iteration retire here
      }
      // ---- 3. charge row of the new state ------------------------------------------------------------------------------
      {
        int lane_c = lane;
        asm volatile("" : "+v"(lane_c));
        const int r0 = lane_c * P;
        double acc[P];
#pragma unroll
        for (int j = 0; j < P; ++j) acc[j] = ACCS[j * 64 + lane_c];
#pragma unroll
        for (int j = 0; j < P; ++j) ROW[pidx<P>(r0 + j + 1)] = (r0 + j < m) ? acc[j] : 0.0;
        lds_sync();
        if (lane_c == 0) {
          ROW[pidx<P>(0)] = acc0;
          ROW[pidx<P>(nx - 1)] = accL;
        }
        if (lane_c < ldx - nx) ROW[pidx<P>(nx + lane_c)] = 0.0;
        lds_sync();
        if (GL == 2 && A.nsteps == 1) store_row<P, 2>(lout, ROW, ldx, lane_c);
        else store_row<P, ST_AUX_STORE>(lout, ROW, ldx, lane_c);
        lds_sync();
        if (!last_step) {   // the next step's lagged charge window from registers: own rows + one DPP hop for the halo
#pragma unroll
          for (int j = 0; j < P; ++j) lw[j + 1] = acc[j];
          lw[0] = from_prev_lane(acc0, acc[P - 1]);
          lw[P + 1] = 0.0;
        }
      }
      double* tmp = lin;
      lin = lout;
      lout = tmp;
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
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int P, bool CN, int GL>
static hipError_t launch_st_inst(const DevArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)st_lds_doubles<P, GL>() * sizeof(double);
  static int blocks_per_cu = 0;
  static int cus = 0;
  if (!blocks_per_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    cus = prop.multiProcessorCount;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)step_kernel_st<P, CN, GL>, 64, lds) != hipSuccess || nb < 1) nb = 4;
    blocks_per_cu = nb;
  }
  int64_t grid = (int64_t)blocks_per_cu * cus;
  if (a.st_waves_per_cu >= 1 && a.st_waves_per_cu <= blocks_per_cu) grid = (int64_t)a.st_waves_per_cu * cus;      // tuning / tests: fewer resident waves
  if (grid > a.B) grid = a.B;
  hipLaunchKernelGGL((step_kernel_st<P, CN, GL>), dim3((unsigned)grid), dim3(64), lds, stream, a);
  return hipGetLastError();
}

template <int P>
static hipError_t launch_st_p(const DevArgs& a, int gl, hipStream_t stream) {
  const bool cn = a.method == PNP_METHOD_CRANK_NICOLSON;
  if (gl == 1) return cn ? launch_st_inst<P, true, 1>(a, stream) : launch_st_inst<P, false, 1>(a, stream);
  if (gl == 2) return cn ? launch_st_inst<P, true, 2>(a, stream) : launch_st_inst<P, false, 2>(a, stream);
  return cn ? launch_st_inst<P, true, 0>(a, stream) : launch_st_inst<P, false, 0>(a, stream);
}

// mode: 0 registers only, 1 charge + gradient rows of the step in LDS, 2 gradient row in LDS
hipError_t launch_step_st(const DevArgs& a, int mode, hipStream_t stream) {
  switch (points_per_lane(a.nx)) {
    case 2: return launch_st_p<2>(a, mode, stream);
    case 4: return launch_st_p<4>(a, mode, stream);
    case 8: return launch_st_p<8>(a, mode, stream);
    case 16: return launch_st_p<16>(a, mode, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace pnp
