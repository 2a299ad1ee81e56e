// Physical mode, large batches: the LANE kernel -- one operating point per LANE (and per sweep direction), every block
// operation of the block-tridiagonal Newton solve in that lane's own registers.
//
// Why (measured on the lane-team / sweep kernels of pnp_newton.hip): with N+1 lanes sharing a block row,
// every Gauss-Jordan pivot is a write -> wave barrier -> read round trip through LDS and most instructions of a row are
// pivot search, selects and broadcasts executed by 7 operating points per wave: ~1 100 wave instructions per grid row for 7
// points, 65 % of a wave's life parked.  Here a wave instruction advances 32 operating points by the same amount of block
// algebra, nothing is exchanged between lanes inside a row and the dependent chain of a row is plain fp64 arithmetic.
//
// What it solves: the same Newton system as the other kernels of the physical mode (comsol_model.py:465-516 asks COMSOL for a
// fully coupled Newton with a direct linear solve; physics as restated in pnp_newton.hip / oracle/pnp_physical.py), point ions
// or steric (MPB) ions, Dirichlet or Stern wall, prescribed wall fluxes, implicit wall kinetics, uniform or graded grid,
// stationary or backward-Euler steps; homogeneous reactions stay with the pivoting kernels.
//
// Algorithm: block Thomas from BOTH ends (twisted factorisation, as newton_sweep2_kernel).  Lanes 0..31 of a wave walk 32
// operating points from the wall upwards (rows 0 .. m-1), lanes 32..63 walk the SAME 32 points from the bulk downwards (rows
// nx-1 .. m+1); the middle row m sees both eliminations and both halves substitute outwards.  With "behind" = the neighbour
// already eliminated and "ahead" = the one not yet, both directions run one instruction stream:
//     D'_i = D_i - Bk_i T_b ,   T_i = D'_i^-1 Ah_i ,   t_i = D'_i^-1 (r_i - Bk_i t_b) ;    x_i = t_i - T_i x_ahead .
// The off-diagonal blocks are never formed: a species row couples to its own species, to the potential and (steric ions) to a
// rank-one term, so Bk T costs 3 fused multiply-adds per element instead of N+1.  D' is factorised in place (LU without row
// exchanges: the species block is a positive diagonal plus a positive rank-one matrix, the potential column only deepens the
// Poisson pivot -- see block_solve in pnp_newton.hip; a pivot monitor flags lanes where that assumption fails) and the N+2
// columns of [Ah | r'] are solved one by one, each leaving for device memory as soon as it is complete.
//
// Layout in HBM: batch-innermost, 16 bytes per lane.  The handle's state c[b][k][i] (x fastest, made for one-grid-per-wave kernels)
// would make every lane touch its own 64-byte line, so a launch first transposes the state of each group of 32 operating points
// into ts[group][i][variable pair][32][2] (pack kernel), works there, and transposes back (unpack kernel, which also raises the
// NaN status).  Records rec[group][i][pair][32][2] in the order they are produced (t, then the columns of T), the Newton update
// xs[group][i][pair][32][2].  Every wave instruction moves two contiguous 512-byte pieces (one per sweep direction), and a wave
// keeps twice the bytes in flight that 8-byte accesses would allow (at most 64 vector-memory instructions are outstanding).
// One wave per SIMD leaves nobody to hide memory latency behind, and loads complete in order BEHIND older stores (one vmcnt
// counter): every pass therefore requests the next row's inputs before it starts on the current row (software pipeline).
// Algorithmic traffic of one Newton iteration and operating point, in doubles per grid row: forward 2N+1 read (c, phi, c_old),
// (N+1)(N+2) written; backward (N+1)(N+2) read, N+1 written; update 2(N+1) read, N+1 written  =  2 (N+1)(N+2) + 6N + 5.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "pnp_lane_common.h"

namespace pnp {

using namespace lane;

namespace {

__device__ __forceinline__ double partner(double v) { return __shfl_xor(v, 32, 64); }   // the same operating point, other direction

constexpr int LG = 32;      // operating points per wave (two lanes each)


}  // namespace

size_t newton_lane_rec_doubles(int nb, int nx) { return (size_t)nx * (size_t)(nb * nb + nb) * LG; }
// ts + xs ((N+1) variables each, padded to pairs) + tco (N, padded)
// (... + tcn, the BDF2 history, like tco)
size_t newton_lane_state_doubles(int nb, int nx) { return (size_t)nx * (size_t)(2 * ((nb + 1) / 2 * 2) + 2 * (nb / 2 * 2)) * LG; }

// ---- state transposition: c[b][k][ldx], phi[b][ldx]  <->  ts[group][i][pair][32][2] (variable N: potential); tco = c on the way in ----
template <bool IN>
__global__ __launch_bounds__(256) void lane_transpose_kernel(const NewtonArgs G) {
  __shared__ double tile[LG][65];
  const int N = G.N, nx = G.nx, ldx = G.ldx;
  const int VP = (N + 2) / 2, CP = (N + 1) / 2;
  const int LGr = G.lane_lg;                  // operating points per group (32: lane kernel, 16: lane-pair kernel)
  const int64_t g = blockIdx.x;
  const int i0 = blockIdx.y * 64;
  const int t = threadIdx.x;
  double* ts = G.lane_ts + (size_t)g * (size_t)nx * VP * LGr * 2;
  double* tco = G.lane_tco + (size_t)g * (size_t)nx * CP * LGr * 2;
  double* tcn = G.lane_tcn + (size_t)g * (size_t)nx * CP * LGr * 2;
  const int64_t b0 = (G.lane_group0 + g) * LGr;
  bool bad = false;
  for (int v = 0; v <= N; ++v) {
    if constexpr (IN) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int op = rr * 4 + (t >> 6), ii = t & 63;
        const int64_t slot = b0 + op;
        double val = 0.0;
        if (op < LGr && slot < G.B && i0 + ii < nx) {
          const int64_t b = G.lane_perm ? G.lane_perm[slot] : slot;
          val = v < N ? G.c[((size_t)b * N + v) * ldx + i0 + ii] : G.phi[(size_t)b * ldx + i0 + ii];
        }
        tile[op][ii] = val;
      }
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int ii = rr * 8 + (t >> 5), op = t & 31;
        if (i0 + ii < nx && op < LGr) {
          const double val = tile[op][ii];
          ts[(((size_t)(i0 + ii) * VP + (v >> 1)) * LGr + op) * 2 + (v & 1)] = val;
          if (v < N && !G.ext_old) tco[(((size_t)(i0 + ii) * CP + (v >> 1)) * LGr + op) * 2 + (v & 1)] = val;
        }
      }
      __syncthreads();
      if (G.bdf2 && G.bdf_hist0 && v < N) {      // BDF2 inside the launch: the history c_n-1 comes in from its home between launches
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int op = rr * 4 + (t >> 6), ii = t & 63;
          const int64_t slot = b0 + op;
          double val = 0.0;
          if (op < LGr && slot < G.B && i0 + ii < nx) {
            const int64_t b = G.lane_perm ? G.lane_perm[slot] : slot;
            val = G.c_old2[((size_t)b * N + v) * ldx + i0 + ii];
          }
          tile[op][ii] = val;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int ii = rr * 8 + (t >> 5), op = t & 31;
          if (i0 + ii < nx && op < LGr) tcn[(((size_t)(i0 + ii) * CP + (v >> 1)) * LGr + op) * 2 + (v & 1)] = tile[op][ii];
        }
        __syncthreads();
      }
      if (G.ext_old && v < N) {      // the previous-level combination prepared by the caller (BDF2) instead of the state itself
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int op = rr * 4 + (t >> 6), ii = t & 63;
          const int64_t slot = b0 + op;
          double val = 0.0;
          if (op < LGr && slot < G.B && i0 + ii < nx) {
            const int64_t b = G.lane_perm ? G.lane_perm[slot] : slot;
            val = G.c_old[((size_t)b * N + v) * ldx + i0 + ii];
          }
          tile[op][ii] = val;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int ii = rr * 8 + (t >> 5), op = t & 31;
          if (i0 + ii < nx && op < LGr) tco[(((size_t)(i0 + ii) * CP + (v >> 1)) * LGr + op) * 2 + (v & 1)] = tile[op][ii];
        }
        __syncthreads();
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int ii = rr * 8 + (t >> 5), op = t & 31;
        double val = 0.0;
        if (i0 + ii < nx && op < LGr) val = ts[(((size_t)(i0 + ii) * VP + (v >> 1)) * LGr + op) * 2 + (v & 1)];
        if (!(fabs(val) < INFINITY)) bad = true;
        tile[op][ii] = val;
      }
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int op = rr * 4 + (t >> 6), ii = t & 63;
        const int64_t slot = b0 + op;
        if (op < LGr && slot < G.B && i0 + ii < nx) {
          const int64_t b = G.lane_perm ? G.lane_perm[slot] : slot;
          if (!(G.lane_mask && !G.lane_mask[b])) {
            if (v < N) G.c[((size_t)b * N + v) * ldx + i0 + ii] = tile[op][ii];
            else G.phi[(size_t)b * ldx + i0 + ii] = tile[op][ii];
          }
        }
      }
      __syncthreads();
      if (G.bdf2 && v < N) {      // ... and the history goes home
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int ii = rr * 8 + (t >> 5), op = t & 31;
          double val = 0.0;
          if (i0 + ii < nx && op < LGr) val = tcn[(((size_t)(i0 + ii) * CP + (v >> 1)) * LGr + op) * 2 + (v & 1)];
          tile[op][ii] = val;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int op = rr * 4 + (t >> 6), ii = t & 63;
          const int64_t slot = b0 + op;
          if (op < LGr && slot < G.B && i0 + ii < nx) {
            const int64_t b = G.lane_perm ? G.lane_perm[slot] : slot;
            if (!(G.lane_mask && !G.lane_mask[b])) G.c_old2[((size_t)b * N + v) * ldx + i0 + ii] = tile[op][ii];
          }
        }
        __syncthreads();
      }
    }
  }
  if constexpr (!IN) {
    const int64_t slot = b0 + (t & 31);
    if (bad && (t & 31) < LGr && slot < G.B) {
      const int64_t b = G.lane_perm ? G.lane_perm[slot] : slot;
      if (!(G.lane_mask && !G.lane_mask[b])) atomicMax(&G.status[b], (int32_t)PNP_STATUS_NAN);
    }
  }
}

// ---- the solver ---------------------------------------------------------------------------------------------------------------
// R32 (option LANE_RECORDS = f32, FUSED only): the columns T of the record travel in SINGLE precision -- t, the elimination, the
// residual and everything else stay double.  The back-substitution x_i = t_i - T_i x_i+1 then carries a relative error of ~1e-7 into the
// Newton UPDATE, not into the solution: the next residual is exact, so the iteration converges to the same state (oracle arithmetic,
// tools/probe/f32_records_oracle.py: the same iteration counts on the bench workload and states equal to 5e-16) while a row moves 46 x
// 8 + 84 x 4 bytes of record instead of 90 x 8 each way: 139 instead of 215 doubles per row and iteration.
template <int NB, int MODE, bool FUSED, bool R32 = false>
__global__ __launch_bounds__(64) void newton_lane_kernel(const NewtonArgs G) {
  static_assert(!R32 || FUSED, "single-precision records: the fused kernel only");
  constexpr int N = NB - 1, NREC = NB * NB + NB;
  constexpr int VP = (NB + 1) / 2, CP = (N + 1) / 2, RP = NREC / 2;     // 16-byte pairs per row: state / previous level / record
  constexpr bool MPB = MODE >= 1;
  constexpr bool FULL = MODE == 2;           // + homogeneous reactions (G.rt) and a constant convection velocity (G.pe); steric code path
  // R32 record of a row, in 16-byte units: TP pairs of doubles (t, padded), then the NB * NB entries of T as floats, four to a unit
  constexpr int TP = (NB + 1) / 2, TF = (NB * NB + 3) / 4, RQ = R32 ? TP + TF : RP;      // RQ: units a row's record takes
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ double s_cb[N][LG];             // bulk concentrations of the wave's operating points
  // The record a row hands to the next one (T, t) and the LU factors of D' do not both fit the register file next to everything
  // else once the blocks are 8 x 8 or 9 x 9: the first TL columns of T then live in LDS (written as they are solved, read back
  // column by column when the next row forms D'), the rest and t stay in registers.  4 waves per CU: at most 40 KiB each.
  // How many: measured at the end of round 4 (tools/probe/lane_tl_variants.sh + ab_libs.sh: the variants alternated process by process
  // on one device, timesteps/s).  9 x 9 blocks, 32 768 points, 8 / 7 / 6 / 5 / 4 columns in LDS: 1.75e6 / 1.77e6 / 1.66e6 / 1.70e6 /
  // 1.59e6 on one box, 1.56e6 / 1.70e6 (8 / 7) on another; 16 384 points 1.14e6 / 1.29e6 (8 / 7): seven -- the eighth column's LDS
  // traffic costs more than the 64 bytes of scratch it saves.  7 x 7 blocks (no scratch to speak of): 0 / 2 / 4 / 6 columns level at
  // 32 768 points (1.20e6), 1.03e6 / 1.02e6 / 0.94e6 / 0.92e6 at 16 384: none.
#ifndef PNP_LANE_TL9      // (A/B builds: tools/probe/lane_tl_variants.sh)
#define PNP_LANE_TL9 7
#endif
#ifndef PNP_LANE_TL8
#define PNP_LANE_TL8 5
#endif
#ifndef PNP_LANE_TL7
#define PNP_LANE_TL7 0
#endif
  constexpr int TL = NB >= 9 ? PNP_LANE_TL9 : (NB >= 8 ? PNP_LANE_TL8 : (NB >= 7 ? PNP_LANE_TL7 : 0));
  constexpr int TR = NB - TL;                // columns of T in registers
  __shared__ double s_T[TL > 0 ? TL * NB : 1][64];
  // per-species constants: with all five arrays read from the kernel arguments the scalar registers run out and are spilled to
  // vector-register lanes (600 v_readlane / v_writelane per row).  The two arrays of the inner loops (q_k beta, ion volumes) stay
  // scalar; those used once per row come from LDS (broadcast reads issued early in the row)
  __shared__ LaneParams sP;
  // MODE 2 only (no LDS in the other instances): the flattened mass-action table and the per-row values its slots point into
  __shared__ double s_react[FULL ? sizeof(ReactionSides) / sizeof(double) + RC_ROWS * 64 : 1];
  ReactionSides& sS = *(ReactionSides*)s_react;
  double (*s_rc)[64] = (double (*)[64])(s_react + sizeof(ReactionSides) / sizeof(double));
  const int lane = threadIdx.x, o = lane & 31;
  const bool side = lane >= 32;              // false: from the wall upwards; true: from the bulk downwards
  const double sgn = side ? -1.0 : 1.0;
  const int nx = G.nx;
  const int m = (nx - 1) >> 1;               // middle row
  const int n_dn = nx - 2 - m;               // rows of the downward half (nx-2 .. m+1; the bulk row nx-1 is its initial state); the
                                             // upward half has m (0 .. m-1) and the middle row
  const int64_t g = blockIdx.x;
  const int64_t slot = (G.lane_group0 + g) * LG + o;
  const int64_t slot_c = slot < G.B ? slot : G.B - 1;   // (parameter loads of the padding lanes stay in range)
  const int64_t b = G.lane_perm ? (int64_t)G.lane_perm[slot_c] : slot_c;      // the operating point this lane holds
  const bool valid = slot < G.B && !(G.lane_mask && !G.lane_mask[b]);
  d2* ts = (d2*)G.lane_ts + (size_t)g * (size_t)nx * VP * LG + o;
  d2* xs = (d2*)G.lane_xs + (size_t)g * (size_t)nx * VP * LG + o;
  d2* tco = (d2*)G.lane_tco + (size_t)g * (size_t)nx * CP * LG + o;
  d2* tcn = (d2*)G.lane_tcn + (size_t)g * (size_t)nx * CP * LG + o;      // BDF2: the time level before the previous one
  d2* rec = (d2*)G.lane_rec + (size_t)g * (size_t)nx * RP * LG + o;
  // Two copies of the state (G.lane_ts, G.lane_xs): the back-substitution of an iteration reads the current one and writes the updated
  // state into the other (see "backward" below); `cur` says which one is current for this lane's operating point.
  int cur = 0;
  d2 *tsc = ts, *tsn = xs;        // (set at the top of every iteration from `cur`)
  auto TS = [&](int i, int p) -> d2& { return tsc[((size_t)i * VP + p) * LG]; };      // current state
  auto TN = [&](int i, int p) -> d2& { return tsn[((size_t)i * VP + p) * LG]; };      // the state being written
  auto XS = [&](int i, int p) -> d2& { return xs[((size_t)i * VP + p) * LG]; };       // (!FUSED: the Newton update; cur stays 0)
  auto CO = [&](int i, int p) -> d2& { return tco[((size_t)i * CP + p) * LG]; };
  auto CN = [&](int i, int p) -> d2& { return tcn[((size_t)i * CP + p) * LG]; };
  auto REC = [&](int i, int p) -> d2& { return rec[((size_t)i * RQ + p) * LG]; };
  const double phiM = G.pb[b * 4 + 0], phiB = G.pb[b * 4 + 1];
  if (!side) {
#pragma unroll
    for (int k = 0; k < N; ++k) s_cb[k][o] = G.cbulk[(size_t)b * N + k];
  }
  if (lane < PNP_NEWTON_MAX_SPECIES) {
    sP.sig[lane] = G.sig[lane];
    sP.peq[lane] = G.peq[lane];
    sP.pe[lane] = G.pe[lane];
    sP.rs[lane] = G.rs[lane];
  }
  if constexpr (FULL) {
    lane_stage_reaction_sides(sS, G.sides, lane);
    lane_reaction_init(s_rc, lane);
  }
  __syncthreads();
  // row visited by the lane in forward step s (clamped to a valid row where the lane rests)
  auto fwd_row = [&](int s) { return side ? (s < n_dn ? nx - 2 - s : m + 1) : (s < m ? s : m); };

  bool have = valid, fresh = true;
  int step = 0, it = 0, total_it = 0, st = PNP_STATUS_OK;
  double upd_prev = INFINITY;
  double alarm = 0.0;            // pivot monitor (sticky): see PIVOT_GROWTH_LIMIT
#ifdef PNP_LANE_STAMPS
  double stamp_f = 0.0, stamp_b = 0.0, stamp_u = 0.0, stamp_n = 0.0;
#endif

  for (;;) {
    if (__ballot(have) == 0ull) break;
    // (Lanes whose operating point has finished its timesteps run along until the slowest point of the wave is done and keep streaming
    // their records: measured HBM bytes are 1.13 x the algorithmic figure on 2-step launches, ~1.05 x on long ones.  Two ways of silencing
    // them were measured on one device in one call, tools/probe/lane_ab.sh, and both LOST: the whole iteration under their exec mask
    // -9 %, every access through a range-checked buffer resource with out-of-range offsets for finished lanes -6 ... -13 %.)
    const NewtonArgs& A = G;
    tsc = cur ? xs : ts;
    tsn = cur ? ts : xs;
    // first iteration of a timestep: the previous time level is the state itself -- unless the caller prepared it (BDF2: G.ext_old)
    const bool first = fresh && !G.ext_old;
    // BDF2 inside a launch of several timesteps (G.bdf2, lane kernels): a step has a history -- the time level before the previous one,
    // kept in CN -- if the launch started with one or it is not the operating point's first step; then the previous-level value is the
    // combination (4 c_n - c_n-1) / 3 and 1/dt carries 3/2 (pnp_capi.hip: newton_timesteps; comsol_model.py:518-531, maxorder 2)
    const bool hist = G.bdf2 && have && (G.bdf_hist0 || step > 0);
    const double sgs = hist ? 1.5 : 1.0;
    if (fresh) {
      it = 0;
      upd_prev = INFINITY;
      fresh = false;
    }
    it += 1;
#ifdef PNP_LANE_STAMPS
    const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
    // =========================== forward: both halves eliminate towards the middle ======================================
    // (an opaque OFFSET, not an opaque pointer: the address stays a known LDS address -- ds_read, not flat_load)
    int poff = 0;
    asm volatile("" : "+v"(poff));
    const LaneParams* P = (const LaneParams*)((const char*)&sP + poff);
    double hc[N], hphi, hw = 0.0, hinv = 1.0;             // the point "here"
    double bphi = 0.0, binv = 1.0;                        // behind: potential, 1/(1 - phi0)
    double eJ[N], eBd[N], eBn[N], eJu[N];                 // behind edge as the point here sees it: outflow, own / neighbour weight, dJ/du
    double Tr[TR][NB], t[NB];                             // behind record: T[j][r] = element (r, j) of D'^-1 Ah, t = D'^-1 r'
    auto Tget = [&](const int j, const int r) { return j < TL ? s_T[(j < TL ? j : 0) * NB + r][lane] : Tr[j >= TL ? j - TL : 0][r]; };
    auto Tset = [&](const int j, const int r, double v) {
      if (j < TL) s_T[(j < TL ? j : 0) * NB + r][lane] = v;
      else Tr[j >= TL ? j - TL : 0][r] = v;
    };
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      t[j] = 0.0;
#pragma unroll
      for (int r = 0; r < NB; ++r) Tset(j, r, 0.0);
    }
    // inputs of the next row, requested one row ahead: the point ahead, the previous time level, the grid weights
    d2 p_a[VP], p_co[CP];
    // (first iteration of a BDF2 step with a history: the previous-level slot of the prefetch carries the level BEFORE the previous one
    //  -- the previous level of such an iteration is formed from the state itself and not read; one pointer chosen per iteration)
    const d2* tprev = (first && hist) ? tcn : tco;
    double p_vi, p_wea, p_web;
    auto request = [&](int s) {
      const int i = fwd_row(s);
      const int ia = side ? i - 1 : i + 1;
#pragma unroll
      for (int p = 0; p < VP; ++p) p_a[p] = TS(ia, p);
#pragma unroll
      for (int p = 0; p < CP; ++p) p_co[p] = tprev[((size_t)i * CP + p) * LG];
      p_vi = G.gv[i];
      p_wea = G.gw[side ? i - 1 : i];
      p_web = G.gw[side ? i : (i > 0 ? i - 1 : 0)];
    };
    double mphi = 0.0;
    {
      // upward half: starts on the wall row with nothing behind it.  Downward half: the bulk row nx-1 (Dirichlet: identity block,
      // x = -(state - bulk values), nothing carried ahead: T = 0) IS its initial state -- it starts on row nx-2 with the bulk point
      // and the last edge behind it, so no row of the sweep needs switched-off edges
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
      if (side) {
        double bc[N], bw = 0.0;
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
          // (FULL: a constant convection velocity v shifts the drift argument by -v h_e / D_k, comsol_model.py:901-903)
          const LEdge e = lane_edge_flux(__builtin_fma(G.qb[k], dphi, dw) - (FULL ? sP.pe[k] / we : 0.0), hc[k], bc[k], we);     // left: row nx-2, right: the bulk row
          eJ[k] = e.J;
          eBd[k] = e.Bp;
          eBn[k] = e.Bm;
          eJu[k] = e.Ju;
          t[k] = -(bc[k] - s_cb[k][o]);
        }
        t[N] = -(bphi - phiB);
        mphi = fabs(t[N]);
        if (!(mphi == mphi)) mphi = INFINITY;
        if constexpr (!FUSED) {
#pragma unroll
          for (int p = 0; p < VP; ++p) {
            d2 v;
            v[0] = t[2 * p];
            v[1] = 2 * p + 1 < NB ? t[2 * p + 1 < NB ? 2 * p + 1 : 0] : 0.0;
            XS(nx - 1, p) = v;
          }
        }
        if (first) {
#pragma unroll
          for (int p = 0; p < CP; ++p) {
            d2 v;
            v[0] = bc[2 * p];
            v[1] = 2 * p + 1 < N ? bc[2 * p + 1 < N ? 2 * p + 1 : 0] : 0.0;
            CO(nx - 1, p) = v;
            if (G.bdf2 && have) CN(nx - 1, p) = v;
          }
        }
      }
    }
    const int S = (n_dn > m ? n_dn : m) + 1;     // row steps of the longer half (the other one rests in the last of them), then the middle row
    for (int s = 0; s < S; ++s) {
      const bool last = s == S - 1;     // the middle row: upward half only
      const bool act = last ? !side : (side ? s < n_dn : s < m);
      // this row's inputs have arrived during the previous row; the next row's are requested before this row's stores are issued
      double ac[N], aphi, co[N];
#pragma unroll
      for (int k = 0; k < N; ++k) {
        ac[k] = p_a[k >> 1][k & 1];
        co[k] = p_co[k >> 1][k & 1];
      }
      aphi = p_a[N >> 1][N & 1];
      const double vi = p_vi, wea = p_wea, web_ = p_web;
      if (!last) request(s + 1);
      // middle row: the upward half needs the downward half's final record (row m+1).  The columns kept in LDS it reads from its
      // partner lane's slots; the rest it reads back from the record in device memory, which the same wave has just written: the
      // release makes those stores visible at the L2, the reads are agent-scope loads that do not look into this CU's L1 (which may
      // hold the previous iteration's lines of that row)
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
      }
      if (act) {
        asm volatile("" : "+v"(poff));    // (the constants are read again in this row instead of living in registers across rows)
        P = (const LaneParams*)((const char*)&sP + poff);
        const int i = fwd_row(s);
        const bool wall = s == 0 && !side;
        // ---- the point ahead and the edge towards it --------------------------------------------------------------------
        double aw = 0.0, ainv = 1.0;
        if constexpr (MPB) {
          double f = 0.0;
#pragma unroll
          for (int k = 0; k < N; ++k) f = __builtin_fma(G.vol[k], ac[k], f);
          aw = -log1p_sc(-f);
          ainv = 1.0 / (1.0 - f);
        }
        const double web = web_;
        double aJ[N], aBd[N], aBn[N], aJu[N];
        {
          const double dphi = aphi - hphi, dw = aw - hw;
          const double rwea = FULL ? 1.0 / wea : 0.0;
#pragma unroll
          for (int k = 0; k < N; ++k) {
            // the edge is evaluated in its left -> right orientation whichever way the lane walks
            const double u = sgn * __builtin_fma(G.qb[k], dphi, dw) - (FULL ? P->pe[k] * rwea : 0.0);
            double h_ = hc[k];              // (opaque: a select between elements of two arrays is otherwise turned into one
            asm volatile("" : "+v"(h_));    //  dynamically indexed stack array)
            const double cl = side ? ac[k] : h_, cr = side ? h_ : ac[k];
            const LEdge e = lane_edge_flux(u, cl, cr, wea);
            aJ[k] = sgn * e.J;                    // (unweighted: the same edge is the next row's behind edge)
            aBd[k] = side ? e.Bm : e.Bp;
            aBn[k] = side ? e.Bp : e.Bm;
            aJu[k] = e.Ju;
          }
        }
        // ---- right-hand side and the diagonal block's ingredients ---------------------------------------------------------
        double rhs[NB], diag[N], Js[N];
        double rho = 0.0;
        double cs_[N];      // the previous-level value of this step: the state itself (backward Euler) or the BDF2 combination
#pragma unroll
        for (int k = 0; k < N; ++k) cs_[k] = hc[k];
        if (first && hist) {      // (under its own branch: eight divisions that only the first iteration of a BDF2 step needs)
#pragma unroll
          for (int k = 0; k < N; ++k) cs_[k] = (4.0 * hc[k] - co[k]) / 3.0;
        }
        if (first) {
#pragma unroll
          for (int p = 0; p < CP; ++p) {
            d2 v;
            v[0] = cs_[2 * p];
            v[1] = 2 * p + 1 < N ? cs_[2 * p + 1 < N ? 2 * p + 1 : 0] : 0.0;
            CO(i, p) = v;
            if (G.bdf2 && have) {      // (a finished point runs along with its wave: its history stays)
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
        double dNN, ahNN;           // Poisson row: diagonal entry, entry of the ahead block
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
        const double pq = wall ? 0.0 : vi;                    // charge term of the Poisson row (not on the wall row)
        // ---- D' = D - Bk T, r' = r - Bk t: column by column, the behind record is consumed on the way -------------------------
        // Bk[k][j] = -eBn_k [j == k] + eJu_k (qb_k [j == N] + vol_j binv) (k < N);  Bk[N][N] = web
        // (Bk T)[k][j] = -eBn_k T[k][j] + eJu_k (qb_k T[N][j] + binv sum_q vol_q T[q][j])
        double D[NB][NB];           // D[r][c], row-major
#pragma unroll
        for (int j = 0; j <= NB; ++j) {         // j == NB: the right-hand side, with t for T[j]
          double col[NB];
#pragma unroll
          for (int r = 0; r < NB; ++r) col[r] = j < NB ? Tget(j < NB ? j : 0, r) : t[r];
          double sj = 0.0;
          if constexpr (MPB) {
#pragma unroll
            for (int q = 0; q < N; ++q) sj = __builtin_fma(G.vol[q], col[q], sj);
            sj *= binv;
          }
          const double tN = col[N];
#pragma unroll
          for (int k = 0; k < N; ++k) {
            double v;
            if (j == NB) v = rhs[k];
            else if (j == N) v = -G.qb[k] * Js[k];
            else v = (MPB ? -Js[k] * (G.vol[j < N ? j : 0] * hinv) : 0.0) + (j == k ? diag[k] : 0.0);
            v = __builtin_fma(eBn[k], col[k], v);
            v = __builtin_fma(-eJu[k], __builtin_fma(G.qb[k], tN, sj), v);
            if (j == NB) rhs[k] = v;
            else D[k][j < NB ? j : 0] = v;
          }
          if (j == NB) rhs[N] = __builtin_fma(-web, tN, rhs[N]);
          else D[N][j < NB ? j : 0] = __builtin_fma(-web, tN, j == N ? dNN : pq * P->peq[j < N ? j : 0]);
        }
        // ---- implicit wall kinetics (fill_row in pnp_newton.hip): rate K g(c_s) E joins the wall flux -------------------------
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
              rhs[k] += a * gq;
#pragma unroll
              for (int j = 0; j < N; ++j) D[k][j] -= (j == sp) ? a * dg : 0.0;
              if (al != 0.0) D[k][N] += a * al * gq;
            }
          }
        }
        // ---- homogeneous reactions (fill_row in pnp_newton.hip): source -(dx^2/D_k) v_i R_k and its Jacobian, a rank-one update of
        // the species block per reaction side; the stoichiometric weights are table data, so the row tests are scalar branches ------
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
              double da[N], db[N];
#pragma unroll
              for (int j = 0; j < N; ++j) {
                da[j] = lane_side_dprod(ra, j, MPB ? G.vol[j] : 0.0);
                db[j] = lane_side_dprod(rb, j, MPB ? G.vol[j] : 0.0);
              }
#pragma unroll
              for (int k = 0; k < N; ++k) {
                const double wa = Sa.w[k] * vrs[k], wb = Sb.w[k] * vrs[k];
                rhs[k] = __builtin_fma(wb, rb.prod, __builtin_fma(wa, ra.prod, rhs[k]));
#pragma unroll
                for (int j = 0; j < N; ++j) D[k][j] = __builtin_fma(-wb, db[j], __builtin_fma(-wa, da[j], D[k][j]));
              }
            }
          }
        }
        // ahead block Ah[k][j] = -aBn_k [j == k] + aJu_k (qb_k [j == N] + vol_j ainv) (k < N);  Ah[N][N] = ahNN
        if (last) {
          // ---- middle row: the downward half's record (row m+1) enters the same way -------------------------------------------
#pragma unroll
          for (int j = 0; j <= NB; ++j) {
            double col[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r)
              if constexpr (R32) {      // t: doubles in the first TP units; T[j][r]: float j * NB + r behind them
                col[r] = j < TL ? s_T[(j < TL ? j : 0) * NB + r][lane + 32]
                         : j == NB ? __hip_atomic_load((const double*)&REC(m + 1, r >> 1) + (r & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                   : (double)__hip_atomic_load((const float*)&REC(m + 1, TP + ((j < NB ? j : 0) * NB + r) / 4) + (((j < NB ? j : 0) * NB + r) & 3),
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              } else
              col[r] = j < TL ? s_T[(j < TL ? j : 0) * NB + r][lane + 32]
                              : __hip_atomic_load((const double*)&REC(m + 1, ((j < NB ? NB + j * NB : 0) + r) >> 1) + (((j < NB ? NB + j * NB : 0) + r) & 1),
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (j == NB: the partner's t)
            double sj = 0.0;
            if constexpr (MPB) {
#pragma unroll
              for (int q = 0; q < N; ++q) sj = __builtin_fma(G.vol[q], col[q], sj);
              sj *= ainv;
            }
#pragma unroll
            for (int k = 0; k < N; ++k) {
              double v = j == NB ? rhs[k] : D[k][j < NB ? j : 0];
              v = __builtin_fma(aBn[k], col[k], v);
              v = __builtin_fma(-aJu[k], __builtin_fma(G.qb[k], col[N], sj), v);
              if (j == NB) rhs[k] = v;
              else D[k][j < NB ? j : 0] = v;
            }
            if (j == NB) rhs[N] = __builtin_fma(-ahNN, col[N], rhs[N]);
            else D[N][j < NB ? j : 0] = __builtin_fma(-ahNN, col[N], D[N][j < NB ? j : 0]);
          }
        }
        // ---- LU in place, no row exchanges; reciprocal pivots on the diagonal ----------------------------------------------
        double gmax = 0.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          const double inv = nrcp(D[k][k]);
          D[k][k] = inv;
#pragma unroll
          for (int r = k + 1; r < NB; ++r) {
            const double l = D[r][k] * inv;
            gmax = fmax(gmax, fabs(l));          // pivot monitor: the multipliers are what row exchanges would have kept below one
            D[r][k] = l;
#pragma unroll
            for (int cc = k + 1; cc < NB; ++cc) D[r][cc] = __builtin_fma(-l, D[k][cc], D[r][cc]);
          }
        }
        alarm = (gmax > G.lane_pivot_limit || gmax != gmax) ? 1.0 : alarm;
        auto solve = [&](double (&y)[NB], const int start) {     // rows < start of y are zero
#pragma unroll
          for (int k = 0; k < NB; ++k) {
            if (k < start) continue;
#pragma unroll
            for (int r = k + 1; r < NB; ++r) y[r] = __builtin_fma(-D[r][k], y[k], y[r]);
          }
#pragma unroll
          for (int k = NB - 1; k >= 0; --k) {
            double acc = y[k];
#pragma unroll
            for (int cc = k + 1; cc < NB; ++cc) acc = __builtin_fma(-D[k][cc], y[cc], acc);
            y[k] = acc * D[k][k];
          }
        };
        // the record leaves in the order it is produced -- t, then the columns of T -- as 16-byte pairs: an element with an even
        // index waits for its odd neighbour
        double held = 0.0;
        f4 heldf = {0.0f, 0.0f, 0.0f, 0.0f};
        auto put = [&](const int e, double v) {
          if constexpr (R32) {
            if (e < NB) {                      // t: pairs of doubles, the odd one out with a zero
              if ((e & 1) == 0) held = v;
              if ((e & 1) == 1 || e == NB - 1) {
                d2 pr;
                pr[0] = held;
                pr[1] = (e & 1) ? v : 0.0;
                __builtin_nontemporal_store(pr, &REC(i, e >> 1));
              }
            } else {                           // T: four floats to a unit
              const int f = e - NB;
              heldf[f & 3] = (float)v;
              if ((f & 3) == 3 || f == NB * NB - 1) __builtin_nontemporal_store(heldf, (f4*)&REC(i, TP + f / 4));
            }
          } else {
            if ((e & 1) == 0) {
              held = v;
            } else {
              d2 pr;
              pr[0] = held;
              pr[1] = v;
              __builtin_nontemporal_store(pr, &REC(i, e >> 1));      // (read back a whole pass later: no reason to keep the line)
            }
          }
        };
        solve(rhs, 0);
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          t[r] = rhs[r];                                    // (middle row: the solution x_m itself)
          put(r, rhs[r]);
        }
        if (!last) {
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            double y[NB];
#pragma unroll
            for (int k = 0; k < N; ++k) {
              double v;
              if (j == N) v = aJu[k] * G.qb[k];
              else v = (MPB ? aJu[k] * (G.vol[j] * ainv) : 0.0) - (j == k ? aBn[k] : 0.0);
              y[k] = v;
            }
            y[N] = j == N ? ahNN : 0.0;
            solve(y, (MPB || j == N) ? 0 : j);
#pragma unroll
            for (int r = 0; r < NB; ++r) {
              Tset(j, r, y[r]);
              put(NB + j * NB + r, y[r]);
            }
          }
        } else {
          // the middle row's solution stays in t (separate update pass: it also goes where that pass looks for it); nothing else is
          // carried out of this row (fresh definitions end the live ranges of the old record here)
          if constexpr (!FUSED) {
#pragma unroll
            for (int p = 0; p < VP; ++p) {
              d2 v;
              v[0] = rhs[2 * p];
              v[1] = 2 * p + 1 < NB ? rhs[2 * p + 1 < NB ? 2 * p + 1 : 0] : 0.0;
              XS(i, p) = v;
            }
          }
#pragma unroll
          for (int j = TL; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < NB; ++r) Tset(j, r, 0.0);
        }
        // ---- the point ahead becomes the point here; its edge is seen from the other end -----------------------------------
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
#ifdef PNP_LANE_STAMPS
    const unsigned long long ts1 = __builtin_readcyclecounter();
#endif
    double lam = 1.0, upd = 0.0;
    if constexpr (FUSED) {
      // =========================== backward + update: x_i = t_i - T_i x_ahead-of-the-elimination, outwards from the middle ============
      // The update (damping, clips, oracle/pnp_physical.py: newton_step) is applied row by row as the solution appears: the new state
      // goes into the OTHER state buffer, the Newton update itself never visits device memory (round 3 wrote it, then read it back
      // together with the state in a third pass: 18 of 233 doubles per row and iteration).  The damping factor -- |d phi| <= dphi_max
      // over the whole grid -- is only known at the end, so the pass runs with a full step; an operating point that turns out to need a
      // shorter one walks its records a second time with the factor it now knows (same x, bit for bit; the buffer it reads was not
      // touched).  That happens in the first iterations of a solve far from its solution, not in the steps of a transient.
      // Rows of the pass: upward half m (x_m = t of the middle row, no record), m-1 ... 0; downward half m+1 ... nx-2, then the bulk row
      // nx-1 (Dirichlet: x = -(state - bulk values), no record).
      double xm[NB];
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        const double own = side ? 0.0 : t[r];
        const double other = partner(own);          // (cross-lane: every lane takes part, the select comes afterwards)
        xm[r] = side ? other : own;
      }
      const bool have0 = have;
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && __ballot(have0 && lam < 1.0) == 0ull) break;
        const bool wr = have0 && (pass == 0 || lam < 1.0);
        const double lm = pass == 0 ? 1.0 : lam;
        double x[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) x[r] = xm[r];
        double mphi_p = 0.0, upd_p = 0.0;
        const int nu_ = (n_dn > m ? n_dn : m) + 1;
        auto row_of = [&](int s) { return side ? (s < n_dn ? m + 1 + s : nx - 1) : (s <= m ? m - s : 0); };
        auto rec_of = [&](int s) { return side ? (s < n_dn ? m + 1 + s : nx - 2) : (s >= 1 && s <= m ? m - s : 0); };   // (a valid row where none is needed)
        d2 Rn[RQ], cn2[VP];
        {
          const int i = row_of(0), ir = rec_of(0);
#pragma unroll
          for (int p = 0; p < RQ; ++p) Rn[p] = __builtin_nontemporal_load(&REC(ir, p));
#pragma unroll
          for (int p = 0; p < VP; ++p) cn2[p] = TS(i, p);
        }
        for (int s = 0; s < nu_; ++s) {
          const bool act = side ? s <= n_dn : s <= m;
          const bool special = side ? s == n_dn : s == 0;      // the row without a record
          const int i = row_of(s);
          d2 R[RQ], c2[VP];
#pragma unroll
          for (int p = 0; p < RQ; ++p) R[p] = Rn[p];
#pragma unroll
          for (int p = 0; p < VP; ++p) c2[p] = cn2[p];
          if (s + 1 < nu_) {
            const int in = row_of(s + 1), irn = rec_of(s + 1);
#pragma unroll
            for (int p = 0; p < RQ; ++p) Rn[p] = __builtin_nontemporal_load(&REC(irn, p));
#pragma unroll
            for (int p = 0; p < VP; ++p) cn2[p] = TS(in, p);
          }
          double y[NB];
#pragma unroll
          for (int r = 0; r < NB; ++r) y[r] = R[r >> 1][r & 1];
          if constexpr (R32) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
              for (int r = 0; r < NB; ++r) {
                const f4 q = __builtin_bit_cast(f4, R[TP + (j * NB + r) / 4]);
                y[r] = __builtin_fma(-(double)q[(j * NB + r) & 3], x[j], y[r]);
              }
          } else {
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < NB; ++r) y[r] = __builtin_fma(-R[(NB + j * NB + r) >> 1][(NB + j * NB + r) & 1], x[j], y[r]);
          }
          if (act) {
            double cc_[N], cn[N];
#pragma unroll
            for (int k = 0; k < N; ++k) cc_[k] = c2[k >> 1][k & 1];
            const double cphi = c2[N >> 1][N & 1];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
              const double sp = side ? (r < N ? -(cc_[r < N ? r : 0] - s_cb[r < N ? r : 0][o]) : -(cphi - phiB)) : xm[r];
              y[r] = special ? sp : y[r];
              x[r] = y[r];
            }
            const double a = fabs(y[N]);
            mphi_p = fmax(mphi_p, a);
            if (!(a == a)) mphi_p = INFINITY;
            // ---- damping, clips, update of this row --------------------------------------------------------------------------
            double f_old = 0.0, f_new = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
              const double rel = fabs(y[k]) / (fabs(cc_[k]) + fabs(s_cb[k][o]) + 1e-300);
              upd_p = fmax(upd_p, rel);
              if (!(y[k] == y[k])) upd_p = INFINITY;
              const double t_ = __builtin_fma(lm, y[k], cc_[k]);
              const double lo = 0.1 * cc_[k];
              cn[k] = t_ < lo ? lo : t_;
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
            if (wr) {
              double out[2 * VP];
#pragma unroll
              for (int k = 0; k < N; ++k) out[k] = cn[k];
              out[N] = __builtin_fma(lm, y[N], cphi);
              if (NB < 2 * VP) out[2 * VP - 1] = 0.0;
#pragma unroll
              for (int p = 0; p < VP; ++p) {
                d2 v;
                v[0] = out[2 * p];
                v[1] = out[2 * p + 1];
                TN(i, p) = v;
              }
            }
          }
        }
        if (pass == 0) {
          mphi = fmax(mphi_p, partner(mphi_p));
          if (A.dphi_max > 0.0 && mphi > A.dphi_max) lam = A.dphi_max / mphi;
          upd = fmax(upd_p, partner(upd_p));
          upd = fmax(upd, mphi * A.vt_inv);
        }
      }
      if (have0) cur ^= 1;            // the state just written is the current one
    } else {
      // =========================== backward: x_i = t_i - T_i x_ahead-of-the-elimination, outwards from the middle ============
      // (t of the middle row IS x_m; the upward half holds it, the downward half gets it from its partner lane)
      double x[NB];
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        const double own = side ? 0.0 : t[r];
        const double other = partner(own);          // (cross-lane: every lane takes part, the select comes afterwards)
        x[r] = side ? other : own;
      }
      if (!side) {
        mphi = fabs(x[N]);
        if (!(mphi == mphi)) mphi = INFINITY;
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
            double y[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) y[r] = R[r >> 1][r & 1];
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
              for (int r = 0; r < NB; ++r) y[r] = __builtin_fma(-R[(NB + j * NB + r) >> 1][(NB + j * NB + r) & 1], x[j], y[r]);
#pragma unroll
            for (int r = 0; r < NB; ++r) x[r] = y[r];
#pragma unroll
            for (int p = 0; p < VP; ++p) {
              d2 v;
              v[0] = y[2 * p];
              v[1] = 2 * p + 1 < NB ? y[2 * p + 1 < NB ? 2 * p + 1 : 0] : 0.0;
              XS(i, p) = v;
            }
            const double a = fabs(y[N]);
            mphi = fmax(mphi, a);
            if (!(a == a)) mphi = INFINITY;
          }
        }
      }
      mphi = fmax(mphi, partner(mphi));
      lam = 1.0;
      if (A.dphi_max > 0.0 && mphi > A.dphi_max) lam = A.dphi_max / mphi;
#ifdef PNP_LANE_STAMPS
      const unsigned long long ts2 = __builtin_readcyclecounter();
#endif
      // =========================== damping, clips, update (oracle/pnp_physical.py: newton_step) ================================
      upd = 0.0;
      {
        const int nu_ = n_dn + 1 > m + 1 ? n_dn + 1 : m + 1;
        auto upd_row = [&](int s) { return side ? (s <= n_dn ? m + 1 + s : nx - 1) : (s <= m ? s : m); };
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
          const bool act = side ? s <= n_dn : s <= m;
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
              const double lo = 0.1 * cc_[k];
              cn[k] = t_ < lo ? lo : t_;
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
      upd = fmax(upd, partner(upd));
      upd = fmax(upd, mphi * A.vt_inv);
    }
#ifdef PNP_LANE_STAMPS
    const unsigned long long ts2 = __builtin_readcyclecounter();
    {   // diagnosis build (tools/probe/lane_stamps.sh): cycles of the passes of this iteration, summed per lane
      const unsigned long long ts3 = __builtin_readcyclecounter();
      stamp_f += (double)(ts1 - ts0);
      stamp_b += (double)(ts2 - ts1);
      stamp_u += (double)(ts3 - ts2);
      stamp_n += 1.0;
    }
#endif
    alarm = fmax(alarm, partner(alarm));
    // =========================== bookkeeping of the lane's operating point (identical in both halves) ========================
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
          if (!side) {
            G.status[b] = alarm > 0.0 ? (int)PNP_STATUS_MAXIT : st;        // (a lane that tripped the pivot monitor is never reported converged)
            G.iters[b] = total_it;
          }
        }
      }
    }
  }
  // An operating point whose current state lives in the second buffer brings it home for the transpose-out kernel (each half its rows;
  // once per launch: 2 (N+1) doubles per row against the ~215 of every Newton iteration)
  if (__ballot(cur != 0) != 0ull) {
    const int lo_ = side ? m + 1 : 0, cnt_ = side ? nx - 1 - m : m + 1;
    const int nmax_ = (nx - 1 - m > m + 1) ? nx - 1 - m : m + 1;
    for (int s = 0; s < nmax_; ++s) {
      if (cur != 0 && s < cnt_) {
        const int i = lo_ + s;
#pragma unroll
        for (int p = 0; p < VP; ++p) ts[((size_t)i * VP + p) * LG] = xs[((size_t)i * VP + p) * LG];
      }
    }
  }
#ifdef PNP_LANE_STAMPS
  if (lane == 0 && slot + 3 < G.B) {      // per wave: mean cycles per iteration of the three passes, in place of four lanes' iteration counts
    G.iters[slot + 0] = (int32_t)(stamp_f / stamp_n);
    G.iters[slot + 1] = (int32_t)(stamp_b / stamp_n);
    G.iters[slot + 2] = (int32_t)(stamp_u / stamp_n);
    G.iters[slot + 3] = (int32_t)stamp_n;
  }
#endif
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
// mode 2: homogeneous reactions and / or a constant convection velocity (MODE 2 instances: the steric code path with both terms)
bool newton_lane_supported(int nb, int nx, int mode) { return nb >= 2 && nb <= 9 && nx >= 5 && mode <= 2; }

bool newton_lane_preferred(int nb, int nx, int64_t B, int mode, const Options& opt) {
  if (!newton_lane_supported(nb, nx, mode)) return false;
  if (opt.newton_kernel != NK_AUTO) return opt.newton_kernel == NK_LANE;      // a kernel family is forced (tests, probes)
  // Measured (tools/probe/lane_sweep.py -> profiles/r03_lane_sweep.jsonl: transient steps, steric ions, Stern wall; timesteps/s of
  // this kernel over the best of the others).  A lane advances its operating point at a fixed pace whatever the batch (N = 8,
  // nx = 512: 3.4 ms per Newton iteration), so its rate grows with the batch until every SIMD holds a wave (B = 32 768) and is
  // HBM-bound from there on; the workgroup-per-point kernels saturate at a few hundred points.  Crossovers:
  //   N >= 5:  0.4-0.6 x at B = 512, 1.3-1.7 x at 2048, 4-5 x at 8192, 8-10 x at 32 768
  //   N = 4:   against the pair kernel in its 512-register build (tools/probe/pair_crossover.py, profiles/r03_pair_crossover.jsonl):
  //            0.36-0.38 x at 2048, 0.64-0.69 x at 4096, 1.07-1.15 x at 8192, 1.5-1.6 x at 16 384, 2.1 x at 32 768; grids too long
  //            for the pair kernel (nx > 512): 0.9-1.0 x at 2048, 1.65 x at 4096, 2.9 x at 8192 against the lane teams
  //   N = 3:   0.6-0.75 x at 8192, 1.1-1.5 x at 32 768 against the pair kernel; grids too long for it (nx > 1024): 2.7 x at 8192
  //   N = 2:   only on grids too long for the pair kernel (1.95 x at B = 8192, nx = 4096)
  // Round 4, small blocks again with the update fused into the back-substitution (+15-20 % for this kernel; tools/probe/family_rates.py
  // --families lane+fused,workgroup -> profiles/r04_family_rates_small_blocks.jsonl: N = 2, 3, 4 x nx = 512, 1024, 2048 x seven batches,
  // 10-step launches, timesteps/s fused lane kernel / the best workgroup-per-point kernel):
  //   N = 4, pair kernel (nx = 512): 4096 1.15e6 / 1.49e6, 8192 2.01e6 / 1.52e6;  lane teams (nx >= 1024): 2048 2.99e5 / 2.64e5
  //   N = 3, pair kernel: nx = 512: 12 288 3.10e6 / 3.50e6, 16 384 3.91e6 / 3.57e6; nx = 1024: 8192 1.24e6 / 1.45e6, 12 288 1.63e6 / 1.48e6;
  //          lane teams (nx = 2048): 2048 1.78e5 / 1.86e5, 4096 3.40e5 / 1.97e5
  //   N = 2, pair kernel: nx = 512: 24 576 6.35e6 / 6.76e6, 32 768 7.57e6 / 6.93e6; nx = 1024: 16 384 2.67e6 / 2.84e6, 24 576 3.23e6 / 2.89e6;
  //          lane teams (nx = 2048): 2048 2.37e5 / 3.42e5, 4096 4.44e5 / 3.38e5
  if (nb >= 6) return B >= 1280;
  const bool pair = newton_pair_threads(nb, nx) > 0;
  if (nb == 5) return pair ? B >= 6144 : B >= 2048;
  if (nb == 4) return pair ? B >= (nx > 768 ? 11264 : 15360) : B >= 3072;
  if (nb == 3) return pair ? B >= (nx > 768 ? 20480 : 28672) : B >= 4096;
  return false;
}

hipError_t launch_lane_transpose(const NewtonArgs& a, int64_t ngroups, bool in, hipStream_t stream) {
  const dim3 tg((unsigned)ngroups, (unsigned)((a.nx + 63) / 64));
  if (in) hipLaunchKernelGGL((lane_transpose_kernel<true>), tg, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((lane_transpose_kernel<false>), tg, dim3(256), 0, stream, a);
  return hipGetLastError();
}

bool newton_lane2_preferred(int nb, int nx, int64_t B, int mode, const Options& opt) {
  if (!newton_lane2_supported(nb, nx, mode)) return false;
  if (opt.newton_kernel != NK_AUTO) return opt.newton_kernel == NK_LANE2;
  // Measured on one device in one call (tools/probe/lane2_probe.sh; N = 8, nx = 512, timesteps/s, lane pair / lane / lane teams):
  // B = 1024 1.16e5 / 1.00e5 / 1.40e5, 2048 2.40e5 / 1.96e5 / 1.43e5, 4096 4.51e5 / 3.85e5 / 1.44e5, 8192 7.31e5 / 7.07e5 / 1.51e5,
  // 16 384 0.99e6 / 1.03e6.  Twice the waves for the same batch, but the distribution overhead (selects, DPP moves, duplicated
  // assembly) leaves a wave's pace only 1.28 x the lane kernel's and two waves on a CU cost each other ~20 %.
  // Round 4: below 10 240 points the lane-quad kernel (pnp_lane4.hip) is ahead of both; the pair keeps the window up to the lane
  // kernel's crossover (profiles/r04_lane4_probe.jsonl: B = 12 288 lane pair 9.96e5 / lane 9.55e5 / lane quad 7.78e5; 16 384 1.05e6 /
  // 1.10e6 / 0.87e6).
  // End of the window, measured again with the fused lane kernel as the alternative (profiles/r04_family_rates.jsonl; lane pair /
  // lane fused): N = 8, nx = 512: B = 14 336 1.17e6 / 1.05e6, 16 384 (1024 waves of 16 points: the last batch in one round) 1.25e6 /
  // 1.19e6, 18 432 0.97e6 / 1.20e6; N = 8, nx = 1024: 16 384 5.6e5 / 5.4e5; N = 6, nx = 1024: 12 288 7.6e5 / 7.4e5, 14 336 7.8e5 / 8.1e5;
  // N = 6, nx = 512: 12 288 1.62e6 / 1.58e6, 16 384 1.90e6 / 2.04e6.
  // N = 7, nx = 512 (profiles/r04_family_rates_n5_n7.jsonl): 12 288 1.39e6 / 1.28e6, 14 336 1.44e6 / 1.47e6;  N = 5: 10 240 1.79e6 / 1.86e6,
  // 12 288 2.12e6 / 2.26e6 -- no window above the lane quad's.
  return B >= 1280 && B <= (nb >= 9 ? 16384 : (nb >= 7 ? 13311 : 10239));
}

template <int NB>
static hipError_t launch_lane_nb(const NewtonArgs& a0, hipStream_t stream) {
  const int64_t groups = (a0.B + LG - 1) / LG;
  const int64_t cap = a0.lane_groups > 0 ? a0.lane_groups : 1;
  for (int64_t g0 = 0; g0 < groups; g0 += cap) {
    NewtonArgs a = a0;
    a.lane_group0 = g0;
    a.lane_lg = LG;
    a.lane_pivot_limit = lane_pivot_limit(a.opt);
    const int64_t ng = groups - g0 < cap ? groups - g0 : cap;
    const dim3 tg((unsigned)ng, (unsigned)((a.nx + 63) / 64));
    hipLaunchKernelGGL((lane_transpose_kernel<true>), tg, dim3(256), 0, stream, a);
    // FUSED: update inside the back-substitution, two state copies, no round trip of the Newton update (-8 % of the bytes; a damped
    // iteration walks the back-substitution twice).  Timesteps: at every batch since the records move with non-temporal accesses
    // (tools/probe/family_rates.py -> profiles/r04_family_rates.jsonl, one device, one call, 20-step launches, timesteps/s separate /
    // fused: N = 8, nx = 512: B = 8192 6.24e5 / 6.95e5, 16 384 1.05e6 / 1.19e6, 24 576 1.32e6 / 1.53e6; N = 6, nx = 1024: 16 384
    // 9.14e5 / 9.15e5, 24 576 1.06e6 / 1.16e6; N = 4, nx = 1024, B = 16 384 1.44e6 / 1.72e6; N = 2, nx = 4096, B = 8192 3.4e5 / 4.2e5;
    // the round-3 measurement with cached record accesses had the separate passes ahead below 24 576 points).  Stationary solves from
    // the bulk state (damped iterations at their start): level, tools/probe/stationary_fused_ab.py -> profiles/r04_stationary_fused_ab.jsonl
    // (separate / fused, ms per solve: 8 x 512 x 16 384 26.6 / 26.3, 32 768 40.0 / 39.6, 6 x 1024 x 16 384 35.7 / 32.9, 3 x 512 x 16 384
    // 11.1 / 10.6) -- so the fused form is the one that runs; option LANE_FUSED = 0 keeps the separate passes selectable.
    const bool fused = (a.opt && a.opt->lane_fused >= 0) ? a.opt->lane_fused != 0 : true;
    const int mode = (a.rt || a.convect) ? 2 : (a.mpb ? 1 : 0);
    const dim3 gk((unsigned)ng), bk(64);
    if (fused) {
      bool launched = false;
      if constexpr (NB >= 3) {
        if (a.opt && a.opt->lane_records_f32 > 0) {
          if (mode == 2) hipLaunchKernelGGL((newton_lane_kernel<NB, 2, true, true>), gk, bk, 0, stream, a);
          else if (mode == 1) hipLaunchKernelGGL((newton_lane_kernel<NB, 1, true, true>), gk, bk, 0, stream, a);
          else hipLaunchKernelGGL((newton_lane_kernel<NB, 0, true, true>), gk, bk, 0, stream, a);
          launched = true;
        }
      }
      if (launched) {
      } else if (mode == 2) hipLaunchKernelGGL((newton_lane_kernel<NB, 2, true>), gk, bk, 0, stream, a);
      else if (mode == 1) hipLaunchKernelGGL((newton_lane_kernel<NB, 1, true>), gk, bk, 0, stream, a);
      else hipLaunchKernelGGL((newton_lane_kernel<NB, 0, true>), gk, bk, 0, stream, a);
    } else {
      if (mode == 2) hipLaunchKernelGGL((newton_lane_kernel<NB, 2, false>), gk, bk, 0, stream, a);
      else if (mode == 1) hipLaunchKernelGGL((newton_lane_kernel<NB, 1, false>), gk, bk, 0, stream, a);
      else hipLaunchKernelGGL((newton_lane_kernel<NB, 0, false>), gk, bk, 0, stream, a);
    }
    hipLaunchKernelGGL((lane_transpose_kernel<false>), tg, dim3(256), 0, stream, a);
  }
  return hipGetLastError();
}

hipError_t launch_newton_lane(const NewtonArgs& a, hipStream_t stream) {
  switch (a.N + 1) {
    case 2: return launch_lane_nb<2>(a, stream);
    case 3: return launch_lane_nb<3>(a, stream);
    case 4: return launch_lane_nb<4>(a, stream);
    case 5: return launch_lane_nb<5>(a, stream);
    case 6: return launch_lane_nb<6>(a, stream);
    case 7: return launch_lane_nb<7>(a, stream);
    case 8: return launch_lane_nb<8>(a, stream);
    case 9: return launch_lane_nb<9>(a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace pnp
