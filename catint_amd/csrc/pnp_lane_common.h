// Helpers shared by the lane kernels of the physical mode (pnp_lane.hip: one lane per operating point and sweep direction;
// pnp_lane2.hip: a lane pair).  gfx950 / MI355X only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "pnp_internal.h"
#include "pnp_math.h"

namespace pnp {
namespace lane {

struct LEdge {
  double Bp, Bm, J, Ju;
};

// Scharfetter-Gummel edge flux, the formulas of edge_flux in pnp_newton.hip (oracle/pnp_physical.py: bernoulli)
__device__ __forceinline__ LEdge lane_edge_flux(double u, double cl, double cr, double w) {
  double B, dB;
  if (fabs(u) < 0.05) {
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
  LEdge e;
  e.Bp = w * B;
  e.Bm = w * (B + u);
  e.J = -(e.Bm * cr - e.Bp * cl);
  e.Ju = -w * ((dB + 1.0) * cr - dB * cl);
  return e;
}


typedef double d2 __attribute__((ext_vector_type(2)));

// The lane kernels eliminate WITHOUT row exchanges (the species block of D' is a positive diagonal plus a positive rank-one matrix,
// the potential column only deepens the Poisson pivot: pnp_lane.hip).  The monitor checks what partial pivoting would have checked:
// a multiplier beyond PIVOT_GROWTH_LIMIT (a pivot 1e8 times smaller than an entry below it: eight digits lost) marks the
// lane, and a marked lane is reported as NOT converged (status 1) whatever its update norm says -- the caller's rerun ladder
// (Calculator.solve_physical) then solves it as a small batch, i.e. with the pivoting lane-team kernels.
constexpr double PIVOT_GROWTH_LIMIT = 1e8;
// (option LANE_PIVOT_LIMIT overrides it: tests set it below one so that every lane trips the monitor)
inline double lane_pivot_limit(const Options* opt) { return (opt && opt->lane_pivot_limit > 0.0) ? opt->lane_pivot_limit : PIVOT_GROWTH_LIMIT; }

struct LaneParams {
  double sig[PNP_NEWTON_MAX_SPECIES], peq[PNP_NEWTON_MAX_SPECIES];
  double pe[PNP_NEWTON_MAX_SPECIES], rs[PNP_NEWTON_MAX_SPECIES];      // MODE 2: convection v dx / D_k, reaction scale dx^2 / D_k
  double qb[PNP_NEWTON_MAX_SPECIES];      // q_k beta for the lane-DEPENDENT accesses of the lane-pair / lane-quad kernels: a select between
                                          // kernel-argument entries is turned into a select of ADDRESSES and a vector load from the argument
                                          // segment -- with an s_waitcnt vmcnt(0) in the middle of every row, behind the prefetched inputs
                                          // of the next row and the record stores of the previous one; an LDS read by index is neither
};

// Copy of the flattened reaction table in LDS (every lane of the wave copies its share of dwords; the caller synchronises).
__device__ __forceinline__ void lane_stage_reaction_sides(ReactionSides& dst, const ReactionSides* src, int lane) {
  if (!src) {
    if (lane == 0) dst.n = 0;
    return;
  }
  const int32_t* s_ = (const int32_t*)src;
  int32_t* d_ = (int32_t*)&dst;
  constexpr int WORDS = (int)(sizeof(ReactionSides) / sizeof(int32_t));
  for (int w = lane; w < WORDS; w += 64) d_[w] = s_[w];
}

// Per-row value table of the reaction terms in LDS, one column per lane: rows 0 .. 7 the concentrations at the point, row 8 the
// constant one (unused reactant slots), rows 9 .. 13 the powers gam^0 .. gam^4 of the activity coefficient.
constexpr int RC_ONE = 8, RC_GAM = 9, RC_ROWS = 14;
static_assert(sizeof(ReactionSides) % sizeof(double) == 0, "the value table follows the reaction table in one LDS array of doubles");
static_assert(PNP_NEWTON_MAX_SPECIES <= RC_ONE && PNP_MAX_REACTANTS == 4, "slot encoding of ReactionSides");

template <int N>
__device__ __forceinline__ void lane_reaction_fill(double (*rc)[64], int lane, const double (&c)[N], double gam) {
#pragma unroll
  for (int k = 0; k < N; ++k) rc[k][lane] = c[k];
  const double g2 = gam * gam;
  rc[RC_GAM + 1][lane] = gam;
  rc[RC_GAM + 2][lane] = g2;
  rc[RC_GAM + 3][lane] = g2 * gam;
  rc[RC_GAM + 4][lane] = g2 * g2;
}
__device__ __forceinline__ void lane_reaction_init(double (*rc)[64], int lane) {      // the rows that never change
  rc[RC_ONE][lane] = 1.0;
  rc[RC_GAM][lane] = 1.0;
}

// One reaction side at the point whose values the table holds (comsol_model.py:781-867, :1064-1084; oracle/pnp_physical.py:
// reaction_rates): prod = k gam^order prod_a c_(i_a); rest[a] = the product with reactant a left out = its derivative with respect
// to c_(i_a), to be added to column col[a] (15: none); steric = prod order gam, the factor of vol_j in d prod / d c_j through gam.
// Branch-free: every side has four reactant slots, unused ones read the constant one.  The side's fields are wave-uniform values in
// LDS that every lane reads for itself; the species index of a slot selects a table ROW, i.e. an address, not a register.
struct LaneSide {
  double prod, rest[PNP_MAX_REACTANTS], steric;
  int col[PNP_MAX_REACTANTS];
};
template <bool MPB>
__device__ __forceinline__ LaneSide lane_reaction_side(const ReactionSides::Side& S, const double (*rc)[64], int lane) {
  const uint32_t slots = S.slots;
  const int order = S.order;
  const double pre = S.k * rc[RC_GAM + order][lane];
  double f[PNP_MAX_REACTANTS];
  LaneSide r;
#pragma unroll
  for (int a = 0; a < PNP_MAX_REACTANTS; ++a) {
    f[a] = rc[(slots >> (4 * a)) & 15u][lane];
    r.col[a] = (int)((slots >> (16 + 4 * a)) & 15u);
  }
  const double p01 = f[0] * f[1], p23 = f[2] * f[3];
  const double q23 = pre * p23, q01 = pre * p01;
  r.rest[0] = f[1] * q23;
  r.rest[1] = f[0] * q23;
  r.rest[2] = q01 * f[3];
  r.rest[3] = q01 * f[2];
  r.prod = q01 * p23;
  r.steric = MPB ? r.prod * (double)order * rc[RC_GAM + 1][lane] : 0.0;      // through gam: d gam / d c_j = vol_j gam^2
  return r;
}
// d prod / d c_j of a side for species column j
__device__ __forceinline__ double lane_side_dprod(const LaneSide& r, int j, double vol_j) {
  double d = r.steric * vol_j;
#pragma unroll
  for (int a = 0; a < PNP_MAX_REACTANTS; ++a) d += (r.col[a] == j) ? r.rest[a] : 0.0;
  return d;
}

}  // namespace lane
}  // namespace pnp
