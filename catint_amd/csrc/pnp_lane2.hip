// Physical mode, mid-size batches: the LANE-PAIR kernel -- the lane kernel of pnp_lane.hip with every block row shared by TWO lanes.
//
// Why: a lane of pnp_lane.hip advances its operating point at a fixed pace whatever the batch (N = 8, nx = 512: 3.4 ms per Newton
// iteration; 67 % of it the forward pass at ~4 200 instructions per grid row), and a batch of 8192 operating points puts one wave on a
// quarter of the SIMDs.  Here four lanes work on one operating point -- two sweep directions (as before) x two halves of the block
// row -- so a wave holds 16 points, a batch gives twice the waves, and a wave's critical path per row is about half as long:
//   * the augmented block [D' | Ah | r] is distributed by COLUMNS: unknown j belongs to half j & 1 (local index j >> 1); the right-hand
//     side is column N+1 of [Ah | r].  D' column j = D column j - Bk T[:, j] needs only T's column j -- which the same lane produced
//     in the previous row -- so the elimination is formed without any exchange;
//   * Gauss-Jordan over the distributed columns: at pivot k the owner's column k travels to its partner by two DPP moves per double
//     (quad_perm broadcast inside the lane quad, no LDS, no select), both lanes compute the same multipliers and update their own
//     columns; the N+2 right-hand columns come out solved (T and t) where they will be needed;
//   * species assembly is split: of each species pair (2kk, 2kk+1) half 0 evaluates the first edge flux, half 1 the second, and the
//     four edge quantities are exchanged by DPP;
//   * the back-substitution sums each lane's partial products over its columns and adds the partner's; the update pass splits the
//     ROWS between the halves.
// Same mathematics, damping, stopping rule, batch-innermost 16-byte layouts and software pipelining as pnp_lane.hip (see there and
// DESIGN.md section 7a); same restrictions (point or steric ions, no homogeneous reactions).  Lane = 4 * point + 2 * direction + half.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "pnp_lane_common.h"

namespace pnp {

using namespace lane;

namespace {

constexpr int OG = 16;      // operating points per wave (four lanes each)

// quad_perm controls: value of lane (perm[i]) of the quad for lane i
constexpr int DPP_FROM_HALF0 = 0 | (0 << 2) | (2 << 4) | (2 << 6);     // both halves read half 0 of their direction
constexpr int DPP_FROM_HALF1 = 1 | (1 << 2) | (3 << 4) | (3 << 6);
constexpr int DPP_SWAP_HALF = 1 | (0 << 2) | (3 << 4) | (2 << 6);      // the partner half
constexpr int DPP_SWAP_SIDE = 2 | (3 << 2) | (0 << 4) | (1 << 6);      // the same half of the other direction

template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);      // (mov_dpp: no "old" operand to materialise -- every lane of a quad is a valid source)
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int Q>
__device__ __forceinline__ double dpp_from_quad_lane(double v) {
  return dpp<Q | (Q << 2) | (Q << 4) | (Q << 6)>(v);
}

}  // namespace

// per group of 16 operating points
size_t newton_lane2_rec_doubles(int nb, int nx) { return (size_t)nx * 2 * (size_t)((((nb + 2) / 2) * nb + 1) / 2 * 2) * OG; }
size_t newton_lane2_state_doubles(int nb, int nx) { return (size_t)nx * (size_t)(2 * ((nb + 1) / 2 * 2) + 2 * (nb / 2 * 2)) * OG; }

// (BDF: the BDF2 history inside the launch -- a template flag here: as a run-time flag it cost the backward-Euler instances 3-8 % in
// registers moved around, tools/probe/ab_old_new.sh)
template <int NB, int MODE, bool BDF>
__global__ __launch_bounds__(64) void newton_lane2_kernel(const NewtonArgs G) {
  constexpr int N = NB - 1;
  constexpr int CL = (NB + 2) / 2;           // local columns of [Ah | r] (NB + 1 columns: unknown j < NB, right-hand side j = NB)
  constexpr int CLD = (NB + 1) / 2;          // local columns of D'
  constexpr int NREC = CL * NB;              // record doubles per lane and row
  constexpr int RP = (NREC + 1) / 2;         // ... in 16-byte pairs
  constexpr int VP = (NB + 1) / 2, CP = (N + 1) / 2;
  constexpr bool MPB = MODE >= 1;
  constexpr bool FULL = MODE == 2;           // + homogeneous reactions and a constant convection velocity (see pnp_lane.hip)
  __shared__ double s_cb[N][OG];
  __shared__ LaneParams sP;
  // MODE 2 only (no LDS in the other instances): the flattened mass-action table and the per-row values its slots point into
  __shared__ double s_react[FULL ? sizeof(ReactionSides) / sizeof(double) + RC_ROWS * 64 : 1];
  ReactionSides& sS = *(ReactionSides*)s_react;
  double (*s_rc)[64] = (double (*)[64])(s_react + sizeof(ReactionSides) / sizeof(double));
  const int lane = threadIdx.x, o = lane >> 2;
  const int h = lane & 1;                     // half of the block row: owns the columns j with j & 1 == h
  const bool side = (lane & 2) != 0;          // false: from the wall upwards; true: from the bulk downwards
  const double sgn = side ? -1.0 : 1.0;
  const int nx = G.nx;
  const int m = (nx - 1) >> 1;
  const int n_dn = nx - 2 - m;
  const int64_t g = blockIdx.x;
  const int64_t slot = (G.lane_group0 + g) * OG + o;
  const int64_t slot_c = slot < G.B ? slot : G.B - 1;
  const int64_t b = G.lane_perm ? (int64_t)G.lane_perm[slot_c] : slot_c;      // the operating point these four lanes hold
  const bool valid = slot < G.B && !(G.lane_mask && !G.lane_mask[b]);
  d2* ts = (d2*)G.lane_ts + (size_t)g * (size_t)nx * VP * OG + o;
  d2* xs = (d2*)G.lane_xs + (size_t)g * (size_t)nx * VP * OG + o;
  d2* tco = (d2*)G.lane_tco + (size_t)g * (size_t)nx * CP * OG + o;
  d2* tcn = (d2*)G.lane_tcn + (size_t)g * (size_t)nx * CP * OG + o;      // BDF2: the time level before the previous one
  d2* rec = (d2*)G.lane_rec + (size_t)g * (size_t)nx * 2 * RP * OG + (size_t)h * RP * OG + o;      // this half's records
  auto TS = [&](int i, int p) -> d2& { return ts[((size_t)i * VP + p) * OG]; };
  auto XS = [&](int i, int p) -> d2& { return xs[((size_t)i * VP + p) * OG]; };
  auto CO = [&](int i, int p) -> d2& { return tco[((size_t)i * CP + p) * OG]; };
  auto CN = [&](int i, int p) -> d2& { return tcn[((size_t)i * CP + p) * OG]; };
  auto REC = [&](int i, int p) -> d2& { return rec[((size_t)i * 2 * RP + p) * OG]; };
  const double phiM = G.pb[b * 4 + 0], phiB = G.pb[b * 4 + 1];
  if ((lane & 3) == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) s_cb[k][o] = G.cbulk[(size_t)b * N + k];
  }
  if (lane < PNP_NEWTON_MAX_SPECIES) {
    sP.sig[lane] = G.sig[lane];
    sP.peq[lane] = G.peq[lane];
    sP.pe[lane] = G.pe[lane];
    sP.rs[lane] = G.rs[lane];
    sP.qb[lane] = G.qb[lane];
  }
  if constexpr (FULL) {
    lane_stage_reaction_sides(sS, G.sides, lane);
    lane_reaction_init(s_rc, lane);
  }
  __syncthreads();
  auto fwd_row = [&](int s) { return side ? (s < n_dn ? nx - 2 - s : m + 1) : (s < m ? s : m); };
  // kind of the local column jj of this lane: unknown j = 2 jj + h
  auto col_j = [&](const int jj) { return 2 * jj + h; };

  bool have = valid, fresh = true;
  int step = 0, it = 0, total_it = 0, st = PNP_STATUS_OK;
  double upd_prev = INFINITY;
  double alarm = 0.0;            // pivot monitor (sticky)

  for (;;) {
    if (__ballot(have) == 0ull) break;
    const NewtonArgs& A = G;
    // first iteration of a timestep: the previous time level is the state itself -- unless the caller prepared it (BDF2: G.ext_old)
    const bool first = fresh && !G.ext_old;
    // BDF2 inside a launch of several timesteps (G.bdf2, lane kernels): a step has a history -- the time level before the previous one,
    // kept in CN -- if the launch started with one or it is not the operating point's first step; then the previous-level value is the
    // combination (4 c_n - c_n-1) / 3 and 1/dt carries 3/2 (pnp_capi.hip: newton_timesteps; comsol_model.py:518-531, maxorder 2)
    const bool hist = BDF && have && (G.bdf_hist0 || step > 0);
    const double sgs = (BDF && hist) ? 1.5 : 1.0;
    if (fresh) {
      it = 0;
      upd_prev = INFINITY;
      fresh = false;
    }
    it += 1;
    // =========================== forward ===================================================================================
    int poff = 0;
    asm volatile("" : "+v"(poff));
    const LaneParams* P = (const LaneParams*)((const char*)&sP + poff);
    double hc[N], hphi, hw = 0.0, hinv = 1.0;
    double bphi = 0.0, binv = 1.0;
    double eJ[N], eBd[N], eBn[N], eJu[N];                 // behind edge, every species, in both halves
    double Tl[CL][NB];                                    // this lane's columns of the behind record [T | t]
#pragma unroll
    for (int jj = 0; jj < CL; ++jj)
#pragma unroll
      for (int r = 0; r < NB; ++r) Tl[jj][r] = 0.0;
    d2 p_a[VP], p_co[CP];
    // (first iteration of a BDF2 step with a history: the previous-level slot of the prefetch carries the level BEFORE the previous one
    //  -- the previous level of such an iteration is formed from the state itself and not read; one pointer chosen per iteration)
    const d2* tprev = (BDF && first && hist) ? tcn : tco;
    double p_vi, p_wea, p_web;
    auto request = [&](int s) {
      const int i = fwd_row(s);
      const int ia = side ? i - 1 : i + 1;
#pragma unroll
      for (int p = 0; p < VP; ++p) p_a[p] = TS(ia, p);
#pragma unroll
      for (int p = 0; p < CP; ++p) p_co[p] = tprev[((size_t)i * CP + p) * OG];
      p_vi = G.gv[i];
      p_wea = G.gw[side ? i - 1 : i];
      p_web = G.gw[side ? i : (i > 0 ? i - 1 : 0)];
    };
    double mphi = 0.0;
    {
      const int i = side ? nx - 2 : 0;
      d2 h2[VP], b2[VP];
#pragma unroll
      for (int p = 0; p < VP; ++p) {
        h2[p] = TS(i, p);
        b2[p] = TS(nx - 1, p);
      }
      request(0);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        hc[k] = h2[k >> 1][k & 1];
        eJ[k] = 0.0;
        eBd[k] = 0.0;
        eBn[k] = 0.0;
        eJu[k] = 0.0;
      }
      hphi = h2[N >> 1][N & 1];
      if constexpr (MPB) {
        double f = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) f = __builtin_fma(G.vol[k], hc[k], f);
        hw = -log1p_sc(-f);
        hinv = 1.0 / (1.0 - f);
      }
      if (side) {        // the bulk row is the downward pair's initial state (see pnp_lane.hip); both halves evaluate it alike
        double bc[N], bw = 0.0, tb[NB];
#pragma unroll
        for (int k = 0; k < N; ++k) bc[k] = b2[k >> 1][k & 1];
        bphi = b2[N >> 1][N & 1];
        if constexpr (MPB) {
          double f = 0.0;
#pragma unroll
          for (int k = 0; k < N; ++k) f = __builtin_fma(G.vol[k], bc[k], f);
          bw = -log1p_sc(-f);
          binv = 1.0 / (1.0 - f);
        }
        const double we = G.gw[nx - 2];
        const double dphi = bphi - hphi, dw = bw - hw;
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const LEdge e = lane_edge_flux(__builtin_fma(G.qb[k], dphi, dw) - (FULL ? sP.pe[k] / we : 0.0), hc[k], bc[k], we);
          eJ[k] = e.J;
          eBd[k] = e.Bp;
          eBn[k] = e.Bm;
          eJu[k] = e.Ju;
          tb[k] = -(bc[k] - s_cb[k][o]);
        }
        tb[N] = -(bphi - phiB);
        mphi = fabs(tb[N]);
        if (!(mphi == mphi)) mphi = INFINITY;
        // t is column NB of [T | t]: it lives in the half NB & 1, local index NB >> 1
        if (h == (NB & 1)) {
#pragma unroll
          for (int r = 0; r < NB; ++r) Tl[NB >> 1][r] = tb[r];
        }
        if (!h) {
#pragma unroll
          for (int p = 0; p < VP; ++p) {
            d2 v;
            v[0] = tb[2 * p];
            v[1] = 2 * p + 1 < NB ? tb[2 * p + 1 < NB ? 2 * p + 1 : 0] : 0.0;
            XS(nx - 1, p) = v;
          }
          if (first) {
#pragma unroll
            for (int p = 0; p < CP; ++p) {
              d2 v;
              v[0] = bc[2 * p];
              v[1] = 2 * p + 1 < N ? bc[2 * p + 1 < N ? 2 * p + 1 : 0] : 0.0;
              CO(nx - 1, p) = v;
              if (BDF && have) CN(nx - 1, p) = v;
            }
          }
        }
      }
    }
    const int S = (n_dn > m ? n_dn : m) + 1;
    for (int s = 0; s < S; ++s) {
      const bool last = s == S - 1;
      const bool act = last ? !side : (side ? s < n_dn : s < m);
      double ac[N], aphi, co[N];
#pragma unroll
      for (int k = 0; k < N; ++k) {
        ac[k] = p_a[k >> 1][k & 1];
        co[k] = p_co[k >> 1][k & 1];
      }
      aphi = p_a[N >> 1][N & 1];
      const double vi = p_vi, wea = p_wea, web = p_web;
      if (!last) request(s + 1);
      if (last) {       // the middle row reads the downward pair's last record back from device memory (see pnp_lane.hip)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
      }
      if (act) {
        asm volatile("" : "+v"(poff));
        P = (const LaneParams*)((const char*)&sP + poff);
        const int i = fwd_row(s);
        const bool wall = s == 0 && !side;
        double aw = 0.0, ainv = 1.0;
        if constexpr (MPB) {
          double f = 0.0;
#pragma unroll
          for (int k = 0; k < N; ++k) f = __builtin_fma(G.vol[k], ac[k], f);
          aw = -log1p_sc(-f);
          ainv = 1.0 / (1.0 - f);
        }
        // ---- ahead edge: of each species pair this half evaluates one, the partner the other ----------------------------------
        double aJ[N], aBd[N], aBn[N], aJu[N];
        {
          const double dphi = aphi - hphi, dw = aw - hw;
          const double rwea = FULL ? 1.0 / wea : 0.0;
#pragma unroll
          for (int kk = 0; kk < (N + 1) / 2; ++kk) {
            const int k0 = 2 * kk, k1 = 2 * kk + 1 < N ? 2 * kk + 1 : 2 * kk;
            const double qb_ = P->qb[h ? k1 : k0];
            const double hc_ = h ? hc[k1] : hc[k0], ac_ = h ? ac[k1] : ac[k0];
            const double u = sgn * __builtin_fma(qb_, dphi, dw) - (FULL ? P->pe[h ? k1 : k0] * rwea : 0.0);
            const double cl = side ? ac_ : hc_, cr = side ? hc_ : ac_;
            const LEdge e = lane_edge_flux(u, cl, cr, wea);
            const double mJ = sgn * e.J, mBd = side ? e.Bm : e.Bp, mBn = side ? e.Bp : e.Bm, mJu = e.Ju;
            const double oJ = dpp<DPP_SWAP_HALF>(mJ), oBd = dpp<DPP_SWAP_HALF>(mBd), oBn = dpp<DPP_SWAP_HALF>(mBn),
                         oJu = dpp<DPP_SWAP_HALF>(mJu);
            aJ[k0] = h ? oJ : mJ;
            aBd[k0] = h ? oBd : mBd;
            aBn[k0] = h ? oBn : mBn;
            aJu[k0] = h ? oJu : mJu;
            if (2 * kk + 1 < N) {
              aJ[k1] = h ? mJ : oJ;
              aBd[k1] = h ? mBd : oBd;
              aBn[k1] = h ? mBn : oBn;
              aJu[k1] = h ? mJu : oJu;
            }
          }
        }
        // ---- right-hand side and the diagonal block's ingredients (every species, both halves) --------------------------------------
        double rhs[NB], diag[N], Js[N];
        double rho = 0.0;
        double cs_[N];      // the previous-level value of this step: the state itself (backward Euler) or the BDF2 combination
#pragma unroll
        for (int k = 0; k < N; ++k) cs_[k] = hc[k];
        if (BDF && first && hist) {      // (under its own branch: eight divisions that only the first iteration of a BDF2 step needs)
#pragma unroll
          for (int k = 0; k < N; ++k) cs_[k] = (4.0 * hc[k] - co[k]) / 3.0;
        }
        if (first && !h) {
#pragma unroll
          for (int p = 0; p < CP; ++p) {
            d2 v;
            v[0] = cs_[2 * p];
            v[1] = 2 * p + 1 < N ? cs_[2 * p + 1 < N ? 2 * p + 1 : 0] : 0.0;
            CO(i, p) = v;
            if (BDF && have) {      // (a finished point runs along with its wave: its history stays)
              d2 w_;
              w_[0] = hc[2 * p];
              w_[1] = 2 * p + 1 < N ? hc[2 * p + 1 < N ? 2 * p + 1 : 0] : 0.0;
              CN(i, p) = w_;
            }
          }
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const double cok = first ? cs_[k] : co[k];
          const double sg = vi * P->sig[k] * sgs;
          rho = __builtin_fma(P->peq[k], hc[k], rho);
          double F = sg * (hc[k] - cok) + aJ[k] + eJ[k];
          if (wall) F -= G.flux[(size_t)b * N + k] * A.fl[k];
          rhs[k] = -F;
          diag[k] = sg + aBd[k] + eBd[k];
          Js[k] = aJu[k] + eJu[k];
        }
        double dNN, ahNN;
        if (wall) {
          if (A.wall_bc == 0) {
            rhs[N] = -(hphi - phiM);
            dNN = 1.0;
            ahNN = 0.0;
          } else {
            rhs[N] = -(wea * (aphi - hphi) + A.stern * (phiM - A.phi_pzc - hphi));
            dNN = -wea - A.stern;
            ahNN = wea;
          }
        } else {
          rhs[N] = -((wea * (aphi - hphi) + web * (bphi - hphi)) + vi * rho);
          dNN = -(wea + web);
          ahNN = wea;
        }
        const double pq = wall ? 0.0 : vi;
        // ---- this lane's columns of D' = D - Bk T and of [Ah | r - Bk t] ---------------------------------------------------------------
        double Dl[NB][CLD], Xl[NB][CL];           // [row][local column]
        // (Bk col)[k] = -eBn_k col[k] + eJu_k (qb_k col[N] + binv sum_q vol_q col[q]),  (Bk col)[N] = web col[N]
        auto minus_bk = [&](const double (&col)[NB], double (&v)[NB], const double (&bn)[N], const double (&ju)[N], double inv_, double wN) {
          double sj = 0.0;
          if constexpr (MPB) {
#pragma unroll
            for (int q = 0; q < N; ++q) sj = __builtin_fma(G.vol[q], col[q], sj);
            sj *= inv_;
          }
#pragma unroll
          for (int k = 0; k < N; ++k) {
            v[k] = __builtin_fma(bn[k], col[k], v[k]);
            v[k] = __builtin_fma(-ju[k], __builtin_fma(G.qb[k], col[N], sj), v[k]);
          }
          v[N] = __builtin_fma(-wN, col[N], v[N]);
        };
#pragma unroll
        for (int jj = 0; jj < CL; ++jj) {
          const int j = col_j(jj);                      // (run-time parity, compile-time jj)
          const bool isrhs = j == NB, isphi = j == N, isspec = j < N;
          double volj = 0.0;
          if constexpr (MPB) volj = (2 * jj + 1 < N) ? (h ? G.vol[2 * jj + 1 < N ? 2 * jj + 1 : 0] : G.vol[2 * jj < N ? 2 * jj : 0])
                                                     : ((2 * jj < N && !h) ? G.vol[2 * jj < N ? 2 * jj : 0] : 0.0);
          double peqj = (2 * jj + 1 < N) ? (h ? P->peq[2 * jj + 1 < N ? 2 * jj + 1 : 0] : P->peq[2 * jj < N ? 2 * jj : 0])
                                         : ((2 * jj < N && !h) ? P->peq[2 * jj < N ? 2 * jj : 0] : 0.0);
          double v[NB];
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const bool dk = (k == 2 * jj && !h) || (k == 2 * jj + 1 && h);          // k == j
            double d = MPB ? -Js[k] * (volj * hinv) : 0.0;                            // (volj = 0 outside the species columns)
            d = dk ? d + diag[k] : d;
            d = isphi ? -G.qb[k] * Js[k] : d;
            v[k] = isrhs ? rhs[k] : d;
          }
          v[N] = isrhs ? rhs[N] : (isphi ? dNN : pq * peqj);
          double col[NB];
#pragma unroll
          for (int r = 0; r < NB; ++r) col[r] = Tl[jj][r];
          minus_bk(col, v, eBn, eJu, binv, web);
          (void)isspec;
          // D' has columns j < NB, the augmented block gets the right-hand side (j == NB) here and the Ah columns below
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            if (jj < CLD) Dl[r][jj < CLD ? jj : 0] = isrhs ? 0.0 : v[r];
            Xl[r][jj] = v[r];            // (kept where j == NB; overwritten with the Ah column otherwise)
          }
        }
        // ---- implicit wall kinetics (see pnp_lane.hip / fill_row) -------------------------------------------------------------------------
        if (wall && A.n_wk > 0) {
          for (int q = 0; q < A.n_wk; ++q) {
            const int sp = A.wk_species[q];
            double cs = 1.0;
#pragma unroll
            for (int k = 0; k < N; ++k) cs = (k == sp) ? hc[k] : cs;
            const double kr = G.wk_k[(size_t)b * PNP_MAX_WALL_REACTIONS + q];
            const double al = A.wk_alpha[q], den = 1.0 / (1.0 + A.wk_sat[q] * cs);
            const double E = al != 0.0 ? exp(al * (phiM - hphi)) : 1.0;
            const double gq = cs * den * E, dg = den * den * E;
#pragma unroll
            for (int k = 0; k < N; ++k) {
              const double a = A.wk_nu[q][k] * kr * A.fl[k];
#pragma unroll
              for (int jj = 0; jj < CL; ++jj) {
                const int j = col_j(jj);
                if (jj < CLD) {
                  if (j < N && j == sp) Dl[k][jj < CLD ? jj : 0] -= a * dg;
                  if (j == N && al != 0.0) Dl[k][jj < CLD ? jj : 0] += a * al * gq;
                }
                if (j == NB) Xl[k][jj] += a * gq;
              }
            }
          }
        }
        // ---- homogeneous reactions (see pnp_lane.hip): every lane evaluates the rates, each half keeps its own columns -----------------------
        if constexpr (FULL) {
          const int ns = __builtin_amdgcn_readfirstlane(sS.n);
          if (ns > 0) {
            double vrs[N];
#pragma unroll
            for (int k = 0; k < N; ++k) vrs[k] = vi * P->rs[k];
            lane_reaction_fill<N>(s_rc, lane, hc, hinv);
            // (two sides per pass: their LDS round trips -- table entry, then the values it points at -- overlap)
            for (int sd = 0; sd < ns; sd += 2) {
              const ReactionSides::Side& Sa = sS.side[sd];
              const ReactionSides::Side& Sb = sS.side[sd + 1];
              const LaneSide ra = lane_reaction_side<MPB>(Sa, s_rc, lane);
              const LaneSide rb = lane_reaction_side<MPB>(Sb, s_rc, lane);
              double dla[CL], dlb[CL];           // d prod / d c_j of this lane's columns
#pragma unroll
              for (int jj = 0; jj < CL; ++jj) {
                double volj = 0.0;
                if constexpr (MPB) volj = (2 * jj + 1 < N) ? (h ? G.vol[2 * jj + 1 < N ? 2 * jj + 1 : 0] : G.vol[2 * jj < N ? 2 * jj : 0])
                                                           : ((2 * jj < N && !h) ? G.vol[2 * jj < N ? 2 * jj : 0] : 0.0);
                dla[jj] = col_j(jj) < N ? lane_side_dprod(ra, col_j(jj), volj) : 0.0;
                dlb[jj] = col_j(jj) < N ? lane_side_dprod(rb, col_j(jj), volj) : 0.0;
              }
#pragma unroll
              for (int k = 0; k < N; ++k) {
                const double wa = Sa.w[k] * vrs[k], wb = Sb.w[k] * vrs[k];
#pragma unroll
                for (int jj = 0; jj < CL; ++jj) {
                  if (jj < CLD) Dl[k][jj < CLD ? jj : 0] = __builtin_fma(-wb, dlb[jj], __builtin_fma(-wa, dla[jj], Dl[k][jj < CLD ? jj : 0]));
                  if (col_j(jj) == NB) Xl[k][jj] = __builtin_fma(wb, rb.prod, __builtin_fma(wa, ra.prod, Xl[k][jj]));
                }
              }
            }
          }
        }
        if (last) {
          // ---- middle row: the downward pair's record of row m+1 (same half, same local columns) enters with the ahead block ----------------
          const d2* other = (const d2*)G.lane_rec + (size_t)g * (size_t)nx * 2 * RP * OG + (size_t)h * RP * OG + o;
#pragma unroll
          for (int jj = 0; jj < CL; ++jj) {
            const bool isrhs = col_j(jj) == NB;
            double col[NB], v[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
              const int e = jj * NB + r;
              col[r] = __hip_atomic_load((const double*)&other[((size_t)(m + 1) * 2 * RP + (e >> 1)) * OG] + (e & 1), __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
              v[r] = isrhs ? Xl[r][jj] : (jj < CLD ? Dl[r][jj < CLD ? jj : 0] : 0.0);
            }
            minus_bk(col, v, aBn, aJu, ainv, ahNN);
#pragma unroll
            for (int r = 0; r < NB; ++r) {
              if (jj < CLD) Dl[r][jj < CLD ? jj : 0] = isrhs ? 0.0 : v[r];
              if (isrhs) Xl[r][jj] = v[r];
            }
          }
        }
        // ---- the Ah columns of the augmented block: Ah[k][j] = -aBn_k [j == k] + aJu_k (qb_k [j == N] + vol_j ainv), Ah[N][N] = ahNN ---------
#pragma unroll
        for (int jj = 0; jj < CL; ++jj) {
          const int j = col_j(jj);
          const bool isrhs = j == NB, isphi = j == N;
          double volj = 0.0;
          if constexpr (MPB) volj = (2 * jj + 1 < N) ? (h ? G.vol[2 * jj + 1 < N ? 2 * jj + 1 : 0] : G.vol[2 * jj < N ? 2 * jj : 0])
                                                     : ((2 * jj < N && !h) ? G.vol[2 * jj < N ? 2 * jj : 0] : 0.0);
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const bool dk = (k == 2 * jj && !h) || (k == 2 * jj + 1 && h);
            double y = MPB ? aJu[k] * (volj * ainv) : 0.0;
            y = dk ? y - aBn[k] : y;
            y = isphi ? aJu[k] * G.qb[k] : y;
            y = (j > NB || last) ? 0.0 : y;                 // (no such column: the partner's spare slot; middle row: only the right-hand side)
            Xl[k][jj] = isrhs ? Xl[k][jj] : y;
          }
          Xl[N][jj] = isrhs ? Xl[N][jj] : ((isphi && !last) ? ahNN : 0.0);
        }
        // ---- Gauss-Jordan over the distributed columns ------------------------------------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          double pc[NB];
#pragma unroll
          for (int r = 0; r < NB; ++r) pc[r] = (k & 1) ? dpp<DPP_FROM_HALF1>(Dl[r][k >> 1]) : dpp<DPP_FROM_HALF0>(Dl[r][k >> 1]);
          {   // pivot monitor (pnp_lane_common.h): rows k+1 .. of the pivot column against the pivot
            double cmax = 0.0;
#pragma unroll
            for (int r = k + 1; r < NB; ++r) cmax = fmax(cmax, fabs(pc[r]));
            alarm = (fabs(pc[k]) * G.lane_pivot_limit < cmax) ? 1.0 : alarm;
          }
          const double inv = nrcp(pc[k]);
#pragma unroll
          for (int jj = 0; jj < CLD; ++jj) {
            if (2 * jj + 1 <= k) continue;          // columns j <= k are finished in both halves (the pivot column is not needed again)
            const double t_ = Dl[k][jj] * inv;
#pragma unroll
            for (int r = 0; r < NB; ++r) Dl[r][jj] = r == k ? t_ : __builtin_fma(-pc[r], t_, Dl[r][jj]);
          }
#pragma unroll
          for (int jj = 0; jj < CL; ++jj) {
            const double t_ = Xl[k][jj] * inv;
#pragma unroll
            for (int r = 0; r < NB; ++r) Xl[r][jj] = r == k ? t_ : __builtin_fma(-pc[r], t_, Xl[r][jj]);
          }
        }
        // The next row's inputs (requested at the top of this row) are made to ARRIVE here, before this row's record stores are issued:
        // vector memory operations retire in order and the compiler's wait-count bookkeeping is conservative across the loop's back
        // edge -- left to itself it consumes these loads after the stores with s_waitcnt vmcnt(0), i.e. every row waits for its own
        // record stores to reach memory (+4 k cycles per row at 1024 waves, tools/probe/lane4_stamps.sh with -DL4_NO_REC_STORE).
        // After a whole row of arithmetic the loads have long landed; the stores then drain behind the next row.  (A sum that needs
        // every loaded register, handed to an empty asm: values are only READ here -- redefining them under this
        // block's partial execution mask loses them for the resting lanes.)
        {
          double touch = (p_vi + p_wea) + p_web;
#pragma unroll
          for (int p = 0; p < VP; ++p) touch += p_a[p][0];
#pragma unroll
          for (int p = 0; p < CP; ++p) touch += p_co[p][0];
          asm volatile("" ::"v"(touch));
        }
        // ---- the record: this lane's columns, in 16-byte pairs ----------------------------------------------------------------------------------
        double held = 0.0;
#pragma unroll
        for (int jj = 0; jj < CL; ++jj)
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            const int e = jj * NB + r;
            Tl[jj][r] = Xl[r][jj];
            if ((e & 1) == 0) {
              held = Xl[r][jj];
              if (e == NREC - 1) {
                d2 pr;
                pr[0] = held;
                pr[1] = 0.0;
                __builtin_nontemporal_store(pr, &REC(i, e >> 1));
              }
            } else {
              d2 pr;
              pr[0] = held;
              pr[1] = Xl[r][jj];
              __builtin_nontemporal_store(pr, &REC(i, e >> 1));
            }
          }
        // (middle row: x_m = the solved right-hand side now sits in its owner's t slot Tl[NB >> 1], where the hand-over below takes it;
        //  the zero Ah columns of that row leave zeros in the other slots)
        bphi = hphi;
        binv = hinv;
        hphi = aphi;
        hw = aw;
        hinv = ainv;
#pragma unroll
        for (int k = 0; k < N; ++k) {
          hc[k] = ac[k];
          eJ[k] = -aJ[k];
          eBn[k] = aBd[k];
          eBd[k] = aBn[k];
          eJu[k] = aJu[k];
        }
      }
    }
    // =========================== backward ========================================================================================================
    // x_m sits in the upward pair's half NB & 1 (local column NB >> 1): quad lane NB & 1; every lane of the quad takes it from there
    double x[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) x[r] = dpp_from_quad_lane<(NB & 1)>(Tl[NB >> 1][r]);
    if (!side) {
      const double a = fabs(x[N]);
      mphi = a;
      if (!(a == a)) mphi = INFINITY;
      if (h == 0) {
#pragma unroll
        for (int p = 0; p < VP; ++p) {
          d2 v;
          v[0] = x[2 * p];
          v[1] = 2 * p + 1 < NB ? x[2 * p + 1 < NB ? 2 * p + 1 : 0] : 0.0;
          XS(m, p) = v;
        }
      }
    }
    {
      const int nb_ = n_dn > m ? n_dn : m;
      auto bwd_row = [&](int s) { return side ? (s < n_dn ? m + 1 + s : nx - 2) : (s < m ? m - 1 - s : 0); };
      d2 Rn[RP];
      {
        const int i = bwd_row(0);
#pragma unroll
        for (int p = 0; p < RP; ++p) Rn[p] = __builtin_nontemporal_load(&REC(i, p));
      }
      for (int s = 0; s < nb_; ++s) {
        const bool act = side ? s < n_dn : s < m;
        const int i = bwd_row(s);
        d2 R[RP];
#pragma unroll
        for (int p = 0; p < RP; ++p) R[p] = Rn[p];
        if (s + 1 < nb_) {
          const int in = bwd_row(s + 1);
#pragma unroll
          for (int p = 0; p < RP; ++p) Rn[p] = __builtin_nontemporal_load(&REC(in, p));
        }
        if (act) {
          // this lane's share of t - T x: its columns (column NB is t itself)
          double y[NB];
#pragma unroll
          for (int r = 0; r < NB; ++r) y[r] = 0.0;
#pragma unroll
          for (int jj = 0; jj < CL; ++jj) {
            const int j = col_j(jj);
            double xj = (2 * jj + 1 < NB) ? (h ? x[2 * jj + 1 < NB ? 2 * jj + 1 : 0] : x[2 * jj < NB ? 2 * jj : 0])
                                          : ((2 * jj < NB && !h) ? x[2 * jj < NB ? 2 * jj : 0] : 0.0);
            const double w = j == NB ? 1.0 : (j < NB ? -xj : 0.0);
#pragma unroll
            for (int r = 0; r < NB; ++r) y[r] = __builtin_fma(R[(jj * NB + r) >> 1][(jj * NB + r) & 1], w, y[r]);
          }
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            y[r] += dpp<DPP_SWAP_HALF>(y[r]);
            x[r] = y[r];
          }
          if (!h) {
#pragma unroll
            for (int p = 0; p < VP; ++p) {
              d2 v;
              v[0] = y[2 * p];
              v[1] = 2 * p + 1 < NB ? y[2 * p + 1 < NB ? 2 * p + 1 : 0] : 0.0;
              XS(i, p) = v;
            }
          }
          const double a = fabs(y[N]);
          mphi = fmax(mphi, a);
          if (!(a == a)) mphi = INFINITY;
        }
      }
    }
    mphi = fmax(mphi, dpp<DPP_SWAP_HALF>(mphi));
    mphi = fmax(mphi, dpp<DPP_SWAP_SIDE>(mphi));
    double lam = 1.0;
    if (A.dphi_max > 0.0 && mphi > A.dphi_max) lam = A.dphi_max / mphi;
    // =========================== update: the rows of a direction are split between its two halves ==================================================
    double upd = 0.0;
    {
      // upward pair: rows 0 .. m; downward pair: rows m+1 .. nx-1; half h takes every other row
      const int lo = side ? m + 1 : 0, cnt = side ? n_dn + 1 : m + 1;
      const int nu_ = ((n_dn + 1 > m + 1 ? n_dn + 1 : m + 1) + 1) / 2;
      auto upd_row = [&](int s) { const int q = 2 * s + h; return lo + (q < cnt ? q : cnt - 1); };
      d2 xn[VP], cn2[VP];
      {
        const int i = upd_row(0);
#pragma unroll
        for (int p = 0; p < VP; ++p) {
          xn[p] = XS(i, p);
          cn2[p] = TS(i, p);
        }
      }
      for (int s = 0; s < nu_; ++s) {
        const bool act = 2 * s + h < cnt;
        const int i = upd_row(s);
        d2 x2[VP], c2[VP];
#pragma unroll
        for (int p = 0; p < VP; ++p) {
          x2[p] = xn[p];
          c2[p] = cn2[p];
        }
        if (s + 1 < nu_) {
          const int in = upd_row(s + 1);
#pragma unroll
          for (int p = 0; p < VP; ++p) {
            xn[p] = XS(in, p);
            cn2[p] = TS(in, p);
          }
        }
        if (act) {
          double du[NB], cc_[N], cn[N];
#pragma unroll
          for (int r = 0; r < NB; ++r) du[r] = x2[r >> 1][r & 1];
          double f_old = 0.0, f_new = 0.0;
#pragma unroll
          for (int k = 0; k < N; ++k) {
            cc_[k] = c2[k >> 1][k & 1];
            const double rel = fabs(du[k]) / (fabs(cc_[k]) + fabs(s_cb[k][o]) + 1e-300);
            upd = fmax(upd, rel);
            if (!(du[k] == du[k])) upd = INFINITY;
            const double t_ = __builtin_fma(lam, du[k], cc_[k]);
            const double lo_ = 0.1 * cc_[k];
            cn[k] = t_ < lo_ ? lo_ : t_;
            if constexpr (MPB) {
              f_old = __builtin_fma(G.vol[k], cc_[k], f_old);
              f_new = __builtin_fma(G.vol[k], cn[k], f_new);
            }
          }
          if constexpr (MPB) {
            const double free_ = 1.0 - f_old;
            const double target = fmax(0.1 * free_, 1e-12);
            if ((1.0 - f_new) < target) {
              const double theta = (free_ - target) / (f_new - f_old);
#pragma unroll
              for (int k = 0; k < N; ++k) cn[k] = __builtin_fma(theta, cn[k] - cc_[k], cc_[k]);
            }
          }
          if (have) {
            double out[2 * VP];
#pragma unroll
            for (int k = 0; k < N; ++k) out[k] = cn[k];
            out[N] = __builtin_fma(lam, du[N], c2[N >> 1][N & 1]);
            if (NB < 2 * VP) out[2 * VP - 1] = 0.0;
#pragma unroll
            for (int p = 0; p < VP; ++p) {
              d2 v;
              v[0] = out[2 * p];
              v[1] = out[2 * p + 1];
              TS(i, p) = v;
            }
          }
        }
      }
    }
    upd = fmax(upd, dpp<DPP_SWAP_HALF>(upd));
    upd = fmax(upd, dpp<DPP_SWAP_SIDE>(upd));
    upd = fmax(upd, mphi * A.vt_inv);
    alarm = fmax(alarm, dpp<DPP_SWAP_SIDE>(alarm));          // (both halves of a direction saw the same pivots)
    // =========================== bookkeeping (identical in the four lanes of an operating point) =====================================================
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
        if (step >= A.nsteps) {
          have = false;
          if ((lane & 3) == 0) {
            G.status[b] = alarm > 0.0 ? (int)PNP_STATUS_MAXIT : st;
            G.iters[b] = total_it;
          }
        }
      }
    }
  }
}

#define LAUNCH_BDF(MODE_)                                                                                              \
  do {                                                                                                                 \
    if (a.bdf2) hipLaunchKernelGGL((newton_lane2_kernel<NB, MODE_, true>), dim3((unsigned)ng), dim3(64), 0, stream, a);  \
    else hipLaunchKernelGGL((newton_lane2_kernel<NB, MODE_, false>), dim3((unsigned)ng), dim3(64), 0, stream, a);       \
  } while (0)

template <int NB>
static hipError_t launch_lane2_nb(const NewtonArgs& a0, hipStream_t stream) {
  const int64_t groups = (a0.B + OG - 1) / OG;
  const int64_t cap = a0.lane_groups > 0 ? a0.lane_groups : 1;
  for (int64_t g0 = 0; g0 < groups; g0 += cap) {
    NewtonArgs a = a0;
    a.lane_group0 = g0;
    a.lane_lg = OG;
    a.lane_pivot_limit = lane_pivot_limit(a.opt);
    const int64_t ng = groups - g0 < cap ? groups - g0 : cap;
    hipError_t e = launch_lane_transpose(a, ng, true, stream);
    if (e != hipSuccess) return e;
    if (a.rt || a.convect) LAUNCH_BDF(2);
    else if (a.mpb) LAUNCH_BDF(1);
    else LAUNCH_BDF(0);
    e = launch_lane_transpose(a, ng, false, stream);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}

hipError_t launch_newton_lane2(const NewtonArgs& a, hipStream_t stream) {
  switch (a.N + 1) {
    case 6: return launch_lane2_nb<6>(a, stream);
    case 7: return launch_lane2_nb<7>(a, stream);
    case 8: return launch_lane2_nb<8>(a, stream);
    case 9: return launch_lane2_nb<9>(a, stream);
    default: return hipErrorInvalidValue;
  }
}

bool newton_lane2_supported(int nb, int nx, int mode) { return nb >= 6 && nb <= 9 && nx >= 5 && mode <= 2; }

}  // namespace pnp
