// Physical mode, batches of a few thousand operating points and long grids: the LANE-QUAD kernel -- the lane kernel of pnp_lane.hip
// with every block row shared by FOUR lanes (eight lanes per operating point: two sweep directions x four column lanes).
//
// Why (round 3 measurements): a lone wave runs at the SIMD's full instruction rate (one vector instruction per
// ~4 cycles), so at a given batch the only way to go faster is more waves.  The lane kernel holds 32 operating points per wave (256
// waves for 8192 points: three SIMDs of every CU idle), the lane-pair kernel 16 (512 waves) at 1.28 x the instructions per row and
// wave.  Here a wave holds 8 points: 8192 points give every SIMD of the chip a wave, and a wave's row costs about half the pair
// kernel's instructions because a lane carries a quarter of the columns:
//   * the augmented block [D' | Ah | r] is distributed by COLUMNS: unknown j belongs to column lane j & 3 (local index j >> 2); the
//     right-hand side is column N+1 of [Ah | r].  D' column j = D column j - Bk T[:, j] needs only T's column j -- which the same lane
//     produced in the previous row -- so the elimination is formed without any exchange;
//   * Gauss-Jordan over the distributed columns: at pivot k the owner's column k reaches the other three lanes of the quad by one DPP
//     quad_perm broadcast per 32-bit half (no LDS, no select), all four compute the same multipliers and update their own columns;
//   * species assembly is split: lane q evaluates the edge fluxes of the species k with k & 3 == q, the quad gathers them by DPP;
//   * the back-substitution sums each lane's partial products over its columns across the quad (two DPP adds); the update pass splits
//     the ROWS between the four column lanes.
// Same mathematics, damping, stopping rule, batch-innermost 16-byte layouts and software pipelining as pnp_lane.hip / pnp_lane2.hip
// (the solve the reference hands to COMSOL, catint/comsol_model.py:465-516); MODE 0 point ions, 1 steric ions, 2 + homogeneous
// reactions (comsol_model.py:781-867) and the constant convection term (:901-903).  Lane = 8 * point + 4 * direction + column lane.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "pnp_lane_common.h"

namespace pnp {

using namespace lane;

namespace {

constexpr int QG = 8;       // operating points per wave (eight lanes each)

template <int CTRL>
__device__ __forceinline__ double dpp4(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int Q>
__device__ __forceinline__ double quad_bcast(double v) {      // the value of column lane Q, in all four lanes of the quad
  return dpp4<Q | (Q << 2) | (Q << 4) | (Q << 6)>(v);
}
// (Q known after unrolling: the switch folds to one of the four broadcasts)
__device__ __forceinline__ double quad_bcast_sel(const int Q, double v) {
  return Q == 0 ? quad_bcast<0>(v) : (Q == 1 ? quad_bcast<1>(v) : (Q == 2 ? quad_bcast<2>(v) : quad_bcast<3>(v)));
}
constexpr int DPP_XOR1 = 1 | (0 << 2) | (3 << 4) | (2 << 6);
constexpr int DPP_XOR2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);
__device__ __forceinline__ double other_side(double v) { return __shfl_xor(v, 4, 64); }      // the same column lane of the other direction

// local columns: of the augmented block [Ah | r] (NB + 1 columns) and of D' (NB columns)
template <int NB> struct QuadShape {
  static constexpr int CL = (NB + 4) / 4;
  static constexpr int CLD = (NB + 3) / 4;
  static constexpr int NREC = CL * NB;          // record doubles per lane and row (slots of columns that do not exist stay untouched)
  static constexpr int RP = (NREC + 1) / 2;     // ... in 16-byte pairs
};

}  // namespace

// per group of 8 operating points
size_t newton_lane4_rec_doubles(int nb, int nx) { return (size_t)nx * 4 * (size_t)((((nb + 4) / 4) * nb + 1) / 2 * 2) * QG; }
size_t newton_lane4_state_doubles(int nb, int nx) { return (size_t)nx * (size_t)(2 * ((nb + 1) / 2 * 2) + 2 * (nb / 2 * 2)) * QG; }

// (BDF: the BDF2 history inside the launch -- a template flag here: as a run-time flag it cost the backward-Euler instances 3-8 % in
// registers moved around, tools/probe/ab_old_new.sh)
template <int NB, int MODE, bool BDF>
__global__ __launch_bounds__(64) void newton_lane4_kernel(const NewtonArgs G) {
  constexpr int N = NB - 1;
  constexpr int CL = QuadShape<NB>::CL, CLD = QuadShape<NB>::CLD, RP = QuadShape<NB>::RP;
  constexpr int VP = (NB + 1) / 2, CP = (N + 1) / 2;
  constexpr bool MPB = MODE >= 1;
  constexpr bool FULL = MODE == 2;
  __shared__ double s_cb[N][QG];
  __shared__ LaneParams sP;
  // MODE 2 only (no LDS in the other instances): the flattened mass-action table and the per-row values its slots point into
  __shared__ double s_react[FULL ? sizeof(ReactionSides) / sizeof(double) + RC_ROWS * 64 : 1];
  ReactionSides& sS = *(ReactionSides*)s_react;
  double (*s_rc)[64] = (double (*)[64])(s_react + sizeof(ReactionSides) / sizeof(double));
  const int lane = threadIdx.x, o = lane >> 3;
  const int q = lane & 3;                     // column lane: owns the columns j with j & 3 == q
  const bool side = (lane & 4) != 0;          // false: from the wall upwards; true: from the bulk downwards
  const double sgn = side ? -1.0 : 1.0;
  const int nx = G.nx;
  const int m = (nx - 1) >> 1;
  const int n_dn = nx - 2 - m;
  const int64_t g = blockIdx.x;
  const int64_t slot = (G.lane_group0 + g) * QG + o;
  const int64_t slot_c = slot < G.B ? slot : G.B - 1;
  const int64_t b = G.lane_perm ? (int64_t)G.lane_perm[slot_c] : slot_c;      // the operating point these eight lanes hold
  const bool valid = slot < G.B && !(G.lane_mask && !G.lane_mask[b]);
  d2* ts = (d2*)G.lane_ts + (size_t)g * (size_t)nx * VP * QG + o;
  d2* xs = (d2*)G.lane_xs + (size_t)g * (size_t)nx * VP * QG + o;
  d2* tco = (d2*)G.lane_tco + (size_t)g * (size_t)nx * CP * QG + o;
  d2* tcn = (d2*)G.lane_tcn + (size_t)g * (size_t)nx * CP * QG + o;      // BDF2: the time level before the previous one
  d2* rec = (d2*)G.lane_rec + (size_t)g * (size_t)nx * 4 * RP * QG + (size_t)q * RP * QG + o;      // this column lane's records
  auto TS = [&](int i, int p) -> d2& { return ts[((size_t)i * VP + p) * QG]; };
  auto XS = [&](int i, int p) -> d2& { return xs[((size_t)i * VP + p) * QG]; };
  auto CO = [&](int i, int p) -> d2& { return tco[((size_t)i * CP + p) * QG]; };
  auto CN = [&](int i, int p) -> d2& { return tcn[((size_t)i * CP + p) * QG]; };
  auto REC = [&](int i, int p) -> d2& { return rec[((size_t)i * 4 * RP + p) * QG]; };
  const double phiM = G.pb[b * 4 + 0], phiB = G.pb[b * 4 + 1];
  if ((lane & 7) == 0) {
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
  auto col_j = [&](const int jj) { return 4 * jj + q; };      // unknown of local column jj (run-time lane, compile-time jj)
  // element 4 jj + q of a table of LEN entries (0 beyond its end): compile-time indices, the column lane selects
  auto pickq = [&](auto&& tab, const int jj, const int LEN) {
    double v = 0.0;
#pragma unroll
    for (int Q = 0; Q < 4; ++Q)
      if (4 * jj + Q < LEN) v = (q == Q) ? tab(4 * jj + Q) : v;
    return v;
  };
  // pairs of this lane's record that hold columns which exist (column j = 4 jj + q <= NB): the others are never stored nor loaded
  const int ncol = (NB - q) / 4 + 1;                        // local columns with j <= NB
  const int rp_valid = (ncol * NB + 1) / 2;

  bool have = valid, fresh = true;
  int step = 0, it = 0, total_it = 0, st = PNP_STATUS_OK;
  double upd_prev = INFINITY;
  double alarm = 0.0;            // pivot monitor (sticky)
  // Phase stagger: every wave of a launch starts with a forward pass (writes, ~75 % of an iteration) followed by a back-substitution
  // that reads its records back at several TB/s -- with one wave per SIMD and equal row times the whole chip is in the same pass at
  // the same time, the back-substitutions of all 1024 waves hit HBM together (5 TB/s, 2.7 x the cycles per row of a wave alone on the
  // chip) and the forward passes leave it idle.  Four groups of waves start a quarter of an iteration apart.
  for (int z = (int)(g & 3) * G.lane_stagger; z > 0; --z) __builtin_amdgcn_s_sleep(127);
#ifdef PNP_LANE_STAMPS
  double stamp_f = 0.0, stamp_b = 0.0, stamp_u = 0.0, stamp_n = 0.0;
  double row_a = 0.0, row_d = 0.0, row_g = 0.0, row_s = 0.0, row_n = 0.0;      // forward row: assembly / D' and Ah columns / Gauss-Jordan / stores
#endif

  for (;;) {
    if (__ballot(have) == 0ull) break;
#ifdef PNP_LANE_STAMPS
    const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
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
    double eJ[N], eBd[N], eBn[N], eJu[N];                 // behind edge, every species, in all four column lanes
    // this lane's columns of the behind record [T | t], [row][local column]: the SAME registers hold the augmented block [Ah | r] of the
    // row being eliminated -- column jj of the new block depends on column jj of the record only, so it is built in place and what
    // the Gauss-Jordan leaves there is the next row's record (no copy at the end of a row, 54 registers less to keep alive)
    double Xl[NB][CL];
#pragma unroll
    for (int jj = 0; jj < CL; ++jj)
#pragma unroll
      for (int r = 0; r < NB; ++r) Xl[r][jj] = 0.0;
    d2 p_a[VP], p_co[CP];
    // (first iteration of a BDF2 step with a history: the previous-level slot of the prefetch carries the level BEFORE the previous one
    //  -- the previous level of such an iteration is formed from the state itself and not read; one pointer chosen per iteration)
    const d2* tprev = (BDF && first && hist) ? tcn : tco;
    double p_vi, p_wea, p_web;
    auto request = [&](int s) {
#ifdef L4_NO_REQUEST      // (diagnosis build: every row reads the first one again -- cache hits)
      const int i = fwd_row(0) + 0 * s;
#else
      const int i = fwd_row(s);
#endif
      const int ia = side ? i - 1 : i + 1;
#pragma unroll
      for (int p = 0; p < VP; ++p) p_a[p] = TS(ia, p);
#pragma unroll
      for (int p = 0; p < CP; ++p) p_co[p] = tprev[((size_t)i * CP + p) * QG];
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
      if (side) {        // the bulk row is the downward quad's initial state (see pnp_lane.hip); all four column lanes evaluate it alike
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
        // t is column NB of [T | t]: it lives in column lane NB & 3, local index NB >> 2
        if (q == (NB & 3)) {
#pragma unroll
          for (int r = 0; r < NB; ++r) Xl[r][NB >> 2] = tb[r];
        }
        if (q == 0) {
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
      if (last) {       // the middle row reads the downward quad's last record back from device memory (see pnp_lane.hip)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
      }
      if (act) {
#ifdef PNP_LANE_STAMPS
        const unsigned long long r0 = __builtin_readcyclecounter();
#endif
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
        // ---- ahead edge: column lane q evaluates the species k with k & 3 == q, the quad gathers them ------------------------------
        double aJ[N], aBd[N], aBn[N], aJu[N];
        {
          const double dphi = aphi - hphi, dw = aw - hw;
          const double rwea = FULL ? 1.0 / wea : 0.0;
#pragma unroll
          for (int kk = 0; kk < (N + 3) / 4; ++kk) {
            // (a group with fewer than four species: the spare lanes recompute its last one)
            auto clampk = [&](int k) { return k < N ? k : N - 1; };
            const int k0 = clampk(4 * kk), k1 = clampk(4 * kk + 1), k2 = clampk(4 * kk + 2), k3 = clampk(4 * kk + 3);
            const int kq = 4 * kk + q < N ? 4 * kk + q : N - 1;          // this lane's species of the group
            const double qb_ = P->qb[kq];
            const double hc_ = q == 0 ? hc[k0] : (q == 1 ? hc[k1] : (q == 2 ? hc[k2] : hc[k3]));
            const double ac_ = q == 0 ? ac[k0] : (q == 1 ? ac[k1] : (q == 2 ? ac[k2] : ac[k3]));
            double pe_ = 0.0;
            if constexpr (FULL) pe_ = P->pe[kq];
            const double u = sgn * __builtin_fma(qb_, dphi, dw) - (FULL ? pe_ * rwea : 0.0);
            const double cl = side ? ac_ : hc_, cr = side ? hc_ : ac_;
#ifdef L4_CHEAP_EDGE      // (diagnosis build: the edge without its exponential)
            LEdge e;
            e.Bp = wea * (1.0 - 0.5 * u);
            e.Bm = wea * (1.0 + 0.5 * u);
            e.J = -(e.Bm * cr - e.Bp * cl);
            e.Ju = -wea * 0.5 * (cr + cl);
#else
            const LEdge e = lane_edge_flux(u, cl, cr, wea);
#endif
            const double mJ = sgn * e.J, mBd = side ? e.Bm : e.Bp, mBn = side ? e.Bp : e.Bm, mJu = e.Ju;
            if (4 * kk + 0 < N) {
              aJ[k0] = quad_bcast<0>(mJ);
              aBd[k0] = quad_bcast<0>(mBd);
              aBn[k0] = quad_bcast<0>(mBn);
              aJu[k0] = quad_bcast<0>(mJu);
            }
            if (4 * kk + 1 < N) {
              aJ[k1] = quad_bcast<1>(mJ);
              aBd[k1] = quad_bcast<1>(mBd);
              aBn[k1] = quad_bcast<1>(mBn);
              aJu[k1] = quad_bcast<1>(mJu);
            }
            if (4 * kk + 2 < N) {
              aJ[k2] = quad_bcast<2>(mJ);
              aBd[k2] = quad_bcast<2>(mBd);
              aBn[k2] = quad_bcast<2>(mBn);
              aJu[k2] = quad_bcast<2>(mJu);
            }
            if (4 * kk + 3 < N) {
              aJ[k3] = quad_bcast<3>(mJ);
              aBd[k3] = quad_bcast<3>(mBd);
              aBn[k3] = quad_bcast<3>(mBn);
              aJu[k3] = quad_bcast<3>(mJu);
            }
          }
        }
        // ---- right-hand side and the diagonal block's ingredients (every species, all column lanes) -----------------------------------
        double rhs[NB], diag[N], Js[N];
        double rho = 0.0;
        double cs_[N];      // the previous-level value of this step: the state itself (backward Euler) or the BDF2 combination
#pragma unroll
        for (int k = 0; k < N; ++k) cs_[k] = hc[k];
        if (BDF && first && hist) {      // (under its own branch: eight divisions that only the first iteration of a BDF2 step needs)
#pragma unroll
          for (int k = 0; k < N; ++k) cs_[k] = (4.0 * hc[k] - co[k]) / 3.0;
        }
        if (first && q == 0) {
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
#ifdef PNP_LANE_STAMPS
        // (the stamps of a row must wait for the section's results: the counter read is scalar code the scheduler would float over the
        //  vector work -- a sum of everything the section produced is handed to an empty asm first)
        {
          double z = rhs[N] + dNN;
#pragma unroll
          for (int k = 0; k < N; ++k) z += (rhs[k] + diag[k]) + (Js[k] + aBn[k]);
          asm volatile("" ::"v"(z));
        }
        const unsigned long long r1 = __builtin_readcyclecounter();
#endif
        // ---- this lane's columns of D' = D - Bk T and of [Ah | r - Bk t] ---------------------------------------------------------------
        double Dl[NB][CLD];           // [row][local column]
        // (Bk col)[k] = -eBn_k col[k] + eJu_k (qb_k col[N] + binv sum_q vol_q col[q]),  (Bk col)[N] = web col[N]
        auto minus_bk = [&](const double (&col)[NB], double (&v)[NB], const double (&bn)[N], const double (&ju)[N], double inv_, double wN) {
          double sj = 0.0;
          if constexpr (MPB) {
#pragma unroll
            for (int qq = 0; qq < N; ++qq) sj = __builtin_fma(G.vol[qq], col[qq], sj);
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
          const int j = col_j(jj);                      // (run-time column lane, compile-time jj)
          const bool isrhs = j == NB, isphi = j == N;
          double volj = 0.0;
          if constexpr (MPB) volj = pickq([&](int k) { return G.vol[k]; }, jj, N);
          const double peqj = pickq([&](int k) { return P->peq[k]; }, jj, N);
          double v[NB];
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const bool dk = (k >> 2) == jj && (k & 3) == q;                           // k == j
            double d = MPB ? -Js[k] * (volj * hinv) : 0.0;                            // (volj = 0 outside the species columns)
            d = dk ? d + diag[k] : d;
            d = isphi ? -G.qb[k] * Js[k] : d;
            v[k] = isrhs ? rhs[k] : (j > NB ? 0.0 : d);
          }
          v[N] = isrhs ? rhs[N] : (isphi ? dNN : (j > NB ? 0.0 : pq * peqj));
          double col[NB];
#pragma unroll
          for (int r = 0; r < NB; ++r) col[r] = Xl[r][jj];
          minus_bk(col, v, eBn, eJu, binv, web);
          // D' has columns j < NB, the augmented block gets the right-hand side (j == NB) here and the Ah columns below
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            if (jj < CLD) Dl[r][jj < CLD ? jj : 0] = j >= NB ? 0.0 : v[r];
            Xl[r][jj] = v[r];            // (kept where j == NB; overwritten with the Ah column otherwise)
          }
        }
        // ---- implicit wall kinetics (see pnp_lane.hip / fill_row) -------------------------------------------------------------------------
        if (wall && A.n_wk > 0) {
          for (int w_ = 0; w_ < A.n_wk; ++w_) {
            const int sp = A.wk_species[w_];
            double cs = 1.0;
#pragma unroll
            for (int k = 0; k < N; ++k) cs = (k == sp) ? hc[k] : cs;
            const double kr = G.wk_k[(size_t)b * PNP_MAX_WALL_REACTIONS + w_];
            const double al = A.wk_alpha[w_], den = 1.0 / (1.0 + A.wk_sat[w_] * cs);
            const double E = al != 0.0 ? exp(al * (phiM - hphi)) : 1.0;
            const double gq = cs * den * E, dg = den * den * E;
#pragma unroll
            for (int k = 0; k < N; ++k) {
              const double a = A.wk_nu[w_][k] * kr * A.fl[k];
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
        // ---- homogeneous reactions (see pnp_lane.hip): every lane evaluates the rates, each column lane keeps its own columns -----------------
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
                if constexpr (MPB) volj = pickq([&](int k) { return G.vol[k]; }, jj, N);
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
          // ---- middle row: the downward quad's record of row m+1 (same column lane, same local columns) enters with the ahead block ---------
          const d2* other = (const d2*)G.lane_rec + (size_t)g * (size_t)nx * 4 * RP * QG + (size_t)q * RP * QG + o;
#pragma unroll
          for (int jj = 0; jj < CL; ++jj) {
            const int j = col_j(jj);
            const bool isrhs = j == NB;
            double col[NB], v[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
              const int e = jj * NB + r;
              double cv = 0.0;
              if (j <= NB)      // (columns that do not exist were never stored)
                cv = __hip_atomic_load((const double*)&other[((size_t)(m + 1) * 4 * RP + (e >> 1)) * QG] + (e & 1), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
              col[r] = cv;
              v[r] = isrhs ? Xl[r][jj] : (jj < CLD ? Dl[r][jj < CLD ? jj : 0] : 0.0);
            }
            minus_bk(col, v, aBn, aJu, ainv, ahNN);
#pragma unroll
            for (int r = 0; r < NB; ++r) {
              if (jj < CLD) Dl[r][jj < CLD ? jj : 0] = j >= NB ? 0.0 : v[r];
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
          if constexpr (MPB) volj = pickq([&](int k) { return G.vol[k]; }, jj, N);
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const bool dk = (k >> 2) == jj && (k & 3) == q;
            double y = MPB ? aJu[k] * (volj * ainv) : 0.0;
            y = dk ? y - aBn[k] : y;
            y = isphi ? aJu[k] * G.qb[k] : y;
            y = (j > NB || last) ? 0.0 : y;                 // (no such column: a spare slot; middle row: only the right-hand side)
            Xl[k][jj] = isrhs ? Xl[k][jj] : y;
          }
          Xl[N][jj] = isrhs ? Xl[N][jj] : ((isphi && !last) ? ahNN : 0.0);
        }
#ifdef PNP_LANE_STAMPS
        {
          double z = 0.0;
#pragma unroll
          for (int r = 0; r < NB; ++r) {
#pragma unroll
            for (int jj = 0; jj < CL; ++jj) z += Xl[r][jj] + (jj < CLD ? Dl[r][jj < CLD ? jj : 0] : 0.0);
          }
          asm volatile("" ::"v"(z));
        }
        const unsigned long long r2 = __builtin_readcyclecounter();
#endif
        // ---- Gauss-Jordan over the distributed columns ------------------------------------------------------------------------------------------
#ifdef L4_NO_GJ      // (diagnosis build, tools/probe/lane4_stamps.sh: what the row costs without its elimination; results are wrong)
#pragma unroll
        for (int k = 0; k < 0; ++k) {
#else
#pragma unroll
        for (int k = 0; k < NB; ++k) {
#endif
          double pc[NB];
#pragma unroll
          for (int r = 0; r < NB; ++r) pc[r] = quad_bcast_sel(k & 3, Dl[r][k >> 2]);
          {   // pivot monitor (pnp_lane_common.h): rows k+1 .. of the pivot column against the pivot
            double cmax = 0.0;
#pragma unroll
            for (int r = k + 1; r < NB; ++r) cmax = fmax(cmax, fabs(pc[r]));
            alarm = (fabs(pc[k]) * G.lane_pivot_limit < cmax) ? 1.0 : alarm;
          }
          const double inv = nrcp(pc[k]);
#pragma unroll
          for (int jj = 0; jj < CLD; ++jj) {
            if (4 * jj + 3 <= k) continue;          // columns j <= k are finished in all four lanes (the pivot column is not needed again)
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
#ifdef PNP_LANE_STAMPS
        {
          double z = 0.0;
#pragma unroll
          for (int r = 0; r < NB; ++r) {
#pragma unroll
            for (int jj = 0; jj < CL; ++jj) z += Xl[r][jj];
          }
          asm volatile("" ::"v"(z));
        }
        const unsigned long long r3 = __builtin_readcyclecounter();
#endif
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
        // ---- the record: this lane's columns, in 16-byte pairs (pairs of columns that do not exist are skipped) -----------------------------
        double held = 0.0;
#pragma unroll
        for (int jj = 0; jj < CL; ++jj)
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            const int e = jj * NB + r;
            if ((e & 1) == 0) {
              held = Xl[r][jj];
              if (e == CL * NB - 1) {
                d2 pr;
                pr[0] = held;
                pr[1] = 0.0;
#ifndef L4_NO_REC_STORE
                if ((e >> 1) < rp_valid) __builtin_nontemporal_store(pr, &REC(i, e >> 1));
#endif
              }
            } else {
              d2 pr;
              pr[0] = held;
              pr[1] = Xl[r][jj];
#ifndef L4_NO_REC_STORE      // (diagnosis builds, tools/probe/lane4_stamps.sh: results are wrong)
              if ((e >> 1) < rp_valid) __builtin_nontemporal_store(pr, &REC(i, e >> 1));
#else
              asm volatile("" :: "v"(pr));
#endif
            }
          }
        // (middle row: x_m = the solved right-hand side now sits in its owner's t slot Xl[.][NB >> 2], where the hand-over below takes it;
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
#ifdef PNP_LANE_STAMPS
        {
          const unsigned long long r4 = __builtin_readcyclecounter();
          row_a += (double)(r1 - r0);
          row_d += (double)(r2 - r1);
          row_g += (double)(r3 - r2);
          row_s += (double)(r4 - r3);
          row_n += 1.0;
        }
#endif
      }
    }
#ifdef PNP_LANE_STAMPS
    const unsigned long long ts1 = __builtin_readcyclecounter();
#endif
    // =========================== backward ========================================================================================================
    // x_m sits in the upward quad's column lane NB & 3 (local column NB >> 2); the upward quad broadcasts it, the downward quad takes it
    // from its partner lanes
    double x[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      const double up = quad_bcast<(NB & 3)>(Xl[r][NB >> 2]);
      const double from_up = other_side(up);            // (cross-lane: every lane takes part, the select comes afterwards)
      x[r] = side ? from_up : up;
    }
    if (!side) {
      const double a = fabs(x[N]);
      mphi = a;
      if (!(a == a)) mphi = INFINITY;
      if (q == 0) {
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
      // A row of this pass is a few hundred instructions against 14 record loads: with the next row's record requested one row ahead the
      // pass ran at one memory round trip per row (34 % of the wave's life in s_waitcnt at one wave per SIMD).  The records do not depend
      // on the solution, so they are requested BD rows ahead into a ring of BD register buffers (compile-time indices: the loop is
      // unrolled BD times); vector memory operations complete in order, so consuming buffer d waits for its own loads only.
      constexpr int BD = 4;
      d2 Rb[BD][RP];
#pragma unroll
      for (int d = 0; d < BD; ++d)
#pragma unroll
        for (int p = 0; p < RP; ++p) {       // (pairs of columns that do not exist are never loaded: they stay zero)
          Rb[d][p][0] = 0.0;
          Rb[d][p][1] = 0.0;
        }
#pragma unroll
      for (int d = 0; d < BD; ++d) {
        if (d < nb_) {
          const int i = bwd_row(d);
#pragma unroll
          for (int p = 0; p < RP; ++p)
            if (p < rp_valid) Rb[d][p] = __builtin_nontemporal_load(&REC(i, p));
        }
      }
      for (int s0 = 0; s0 < nb_; s0 += BD) {
#pragma unroll
       for (int d = 0; d < BD; ++d) {
        const int s = s0 + d;
        if (s >= nb_) break;
        const bool act = side ? s < n_dn : s < m;
        const int i = bwd_row(s);
        d2 R[RP];
#pragma unroll
        for (int p = 0; p < RP; ++p) R[p] = Rb[d][p];
        if (s + BD < nb_) {
          const int in = bwd_row(s + BD);
#pragma unroll
          for (int p = 0; p < RP; ++p)
            if (p < rp_valid) Rb[d][p] = __builtin_nontemporal_load(&REC(in, p));
        }
        if (act) {
          // this lane's share of t - T x: its columns (column NB is t itself)
          double y[NB];
#pragma unroll
          for (int r = 0; r < NB; ++r) y[r] = 0.0;
#pragma unroll
          for (int jj = 0; jj < CL; ++jj) {
            const int j = col_j(jj);
            const double xj = pickq([&](int k) { return x[k]; }, jj, NB);
            const double w = j == NB ? 1.0 : (j < NB ? -xj : 0.0);
#pragma unroll
            for (int r = 0; r < NB; ++r) {
              const double rv = j <= NB ? R[(jj * NB + r) >> 1][(jj * NB + r) & 1] : 0.0;      // (a pair may straddle into a column that does not exist)
              y[r] = __builtin_fma(rv, w, y[r]);
            }
          }
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            y[r] += dpp4<DPP_XOR1>(y[r]);
            y[r] += dpp4<DPP_XOR2>(y[r]);
            x[r] = y[r];
          }
          if (q == 0) {
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
    }
    mphi = fmax(mphi, other_side(mphi));
    double lam = 1.0;
    if (A.dphi_max > 0.0 && mphi > A.dphi_max) lam = A.dphi_max / mphi;
#ifdef PNP_LANE_STAMPS
    const unsigned long long ts2 = __builtin_readcyclecounter();
#endif
    // =========================== update: the rows of a direction are split between its four column lanes =========================================
    double upd = 0.0;
    {
      // upward quad: rows 0 .. m; downward quad: rows m+1 .. nx-1; column lane q takes every fourth row
      const int lo = side ? m + 1 : 0, cnt = side ? n_dn + 1 : m + 1;
      const int nu_ = ((n_dn + 1 > m + 1 ? n_dn + 1 : m + 1) + 3) / 4;
      auto upd_row = [&](int s) { const int z = 4 * s + q; return lo + (z < cnt ? z : cnt - 1); };
      // (requested UD rows ahead, like the records of the backward pass)
      constexpr int UD = 4;
      d2 xb[UD][VP], cb2[UD][VP];
#pragma unroll
      for (int d = 0; d < UD; ++d) {
        const int i = upd_row(d);          // (clamped to the lane's last row beyond its share: a valid address)
#pragma unroll
        for (int p = 0; p < VP; ++p) {
          xb[d][p] = XS(i, p);
          cb2[d][p] = TS(i, p);
        }
      }
      for (int s0 = 0; s0 < nu_; s0 += UD) {
#pragma unroll
       for (int d = 0; d < UD; ++d) {
        const int s = s0 + d;
        if (s >= nu_) break;
        const bool act = 4 * s + q < cnt;
        const int i = upd_row(s);
        d2 x2[VP], c2[VP];
#pragma unroll
        for (int p = 0; p < VP; ++p) {
          x2[p] = xb[d][p];
          c2[p] = cb2[d][p];
        }
        if (s + UD < nu_) {
          const int in = upd_row(s + UD);
#pragma unroll
          for (int p = 0; p < VP; ++p) {
            xb[d][p] = XS(in, p);
            cb2[d][p] = TS(in, p);
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
    }
    upd = fmax(upd, dpp4<DPP_XOR1>(upd));
    upd = fmax(upd, dpp4<DPP_XOR2>(upd));
    upd = fmax(upd, other_side(upd));
    upd = fmax(upd, mphi * A.vt_inv);
    alarm = fmax(alarm, other_side(alarm));          // (the four column lanes of a direction saw the same pivots)
#ifdef PNP_LANE_STAMPS
    {   // diagnosis build (tools/probe/lane4_stamps.sh): cycles of the three passes of this iteration, summed per wave
      const unsigned long long ts3 = __builtin_readcyclecounter();
      stamp_f += (double)(ts1 - ts0);
      stamp_b += (double)(ts2 - ts1);
      stamp_u += (double)(ts3 - ts2);
      stamp_n += 1.0;
    }
#endif
    // =========================== bookkeeping (identical in the eight lanes of an operating point) ====================================================
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
          if ((lane & 7) == 0) {
            G.status[b] = alarm > 0.0 ? (int)PNP_STATUS_MAXIT : st;
            G.iters[b] = total_it;
          }
        }
      }
    }
  }
#ifdef PNP_LANE_STAMPS
  if (lane == 0 && slot + 3 < G.B) {      // per wave: mean cycles per iteration of the three passes, in place of four points' iteration counts
    G.iters[slot + 0] = (int32_t)(stamp_f / stamp_n);
    G.iters[slot + 1] = (int32_t)(stamp_b / stamp_n);
    G.iters[slot + 2] = (int32_t)(stamp_u / stamp_n);
    G.iters[slot + 3] = (int32_t)stamp_n;
    if (slot + 7 < G.B) {      // ... and the sections of a forward row, mean cycles per row
      G.iters[slot + 4] = (int32_t)(row_a / row_n);
      G.iters[slot + 5] = (int32_t)(row_d / row_n);
      G.iters[slot + 6] = (int32_t)(row_g / row_n);
      G.iters[slot + 7] = (int32_t)(row_s / row_n);
    }
  }
#endif
}

#define LAUNCH_BDF(MODE_)                                                                                              \
  do {                                                                                                                 \
    if (a.bdf2) hipLaunchKernelGGL((newton_lane4_kernel<NB, MODE_, true>), dim3((unsigned)ng), dim3(64), 0, stream, a);  \
    else hipLaunchKernelGGL((newton_lane4_kernel<NB, MODE_, false>), dim3((unsigned)ng), dim3(64), 0, stream, a);       \
  } while (0)

template <int NB>
static hipError_t launch_lane4_nb(const NewtonArgs& a0, hipStream_t stream) {
  const int64_t groups = (a0.B + QG - 1) / QG;
  const int64_t cap = a0.lane_groups > 0 ? a0.lane_groups : 1;
  for (int64_t g0 = 0; g0 < groups; g0 += cap) {
    NewtonArgs a = a0;
    a.lane_group0 = g0;
    a.lane_lg = QG;
    a.lane_pivot_limit = lane_pivot_limit(a.opt);
    a.lane_stagger = (a.opt && a.opt->lane_stagger >= 0) ? a.opt->lane_stagger : 0;
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

hipError_t launch_newton_lane4(const NewtonArgs& a, hipStream_t stream) {
  switch (a.N + 1) {
    case 6: return launch_lane4_nb<6>(a, stream);
    case 7: return launch_lane4_nb<7>(a, stream);
    case 8: return launch_lane4_nb<8>(a, stream);
    case 9: return launch_lane4_nb<9>(a, stream);
    default: return hipErrorInvalidValue;
  }
}

bool newton_lane4_supported(int nb, int nx, int mode) { return nb >= 6 && nb <= 9 && nx >= 5 && mode <= 2; }

bool newton_lane4_preferred(int nb, int nx, int64_t B, int mode, const Options& opt) {
  if (!newton_lane4_supported(nb, nx, mode)) return false;
  if (opt.newton_kernel != NK_AUTO) return opt.newton_kernel == NK_LANE4;
  // Measured on one device in one process (tools/probe/lane4_probe.py -> profiles/r04_lane4_probe.jsonl; N = 8 steric, nx = 512,
  // timesteps/s, lane quad / lane pair / lane / lane teams): B = 1024 1.66e5 / 1.19e5 / 1.00e5 / 1.38e5, 2048 3.24e5 / 2.32e5 / 1.97e5 /
  // 1.41e5, 4096 5.61e5 / 4.46e5 / 3.86e5 / 1.45e5, 8192 9.09e5 / 7.43e5 / 7.18e5, 12 288 7.78e5 / 9.96e5 / 9.55e5, 16 384 0.87e6 /
  // 1.05e6 / 1.10e6; nx = 4096: B = 2048 3.97e4 / 2.86e4 / 2.49e4, 8192 1.10e5 / 0.97e5 / 0.92e5; N = 6, nx = 1024: 8192 6.50e5 / 5.53e5 /
  // 5.26e5, 16 384 6.98e5 / 8.28e5 / 8.15e5.  Eight points per wave: the chip is full at 8192 points (one wave per SIMD) and the
  // kernel saturates there; beyond, a second round of waves per SIMD costs more than the lane pair's and the lane kernel's larger
  // waves (a wave's row: ~2000 vector instructions for 8 points against ~4200 for the lane kernel's 32).
  return B >= 896 && B < 10240;
}

}  // namespace pnp
