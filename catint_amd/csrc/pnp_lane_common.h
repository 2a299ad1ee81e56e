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
};

// One side (0: educts, forward rate constant; 1: products, backward) of mass-action reaction r at a grid point with concentrations
// c and activity coefficient gam = 1/(1 - phi0) (comsol_model.py:781-867, :1064-1084; oracle/pnp_physical.py: reaction_rates; the
// formulas of fill_row in pnp_newton.hip):  prod = k gam^n prod_a c_a,  dprod[j] = d prod / d c_j,  sw[k] = the net stoichiometric
// weight with which (forward - backward) enters R_k -- table data, i.e. wave-uniform scalars.  Returns false for a side without a rate.
template <int N, bool MPB>
__device__ __forceinline__ bool lane_reaction_side(const ReactionTable* rt, const int r, const int side, const double (&c)[N],
                                                   const double gam, const double* vol, double& prod, double (&dprod)[N],
                                                   double (&sw)[N]) {
  const int nl = rt->n_lhs[r], nrh = rt->n_rhs[r];
  const int n = side == 0 ? nl : nrh;
  const int32_t* idx = side == 0 ? rt->lhs[r] : rt->rhs[r];
  const double kk = side == 0 ? rt->kf[r] : rt->kr[r];
  if (kk == 0.0) return false;             // n = 0 with a rate: constant source (the side consists of excluded species, e.g. H2O)
  auto pick = [&](int i_) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) v = (k == i_) ? c[k] : v;
    return v;
  };
  double pre = kk;
  for (int a = 0; a < n; ++a) pre *= gam;
  prod = pre;
  for (int a = 0; a < n; ++a) prod *= pick(idx[a]);
#pragma unroll
  for (int j = 0; j < N; ++j) dprod[j] = MPB ? prod * n * (vol[j] * gam) : 0.0;     // through gam (zero volumes: point ions)
  for (int a = 0; a < n; ++a) {
    double rest = pre;
    for (int b2 = 0; b2 < n; ++b2)
      if (b2 != a) rest *= pick(idx[b2]);
    const int ia = idx[a];
#pragma unroll
    for (int j = 0; j < N; ++j) dprod[j] += (j == ia) ? rest : 0.0;
  }
  const double sg = side == 0 ? 1.0 : -1.0;                    // forward minus backward
#pragma unroll
  for (int k = 0; k < N; ++k) sw[k] = 0.0;
  for (int a = 0; a < nl; ++a) {
    const int jsp = rt->lhs[r][a];
#pragma unroll
    for (int k = 0; k < N; ++k) sw[k] -= (k == jsp) ? sg : 0.0;   // educts lose
  }
  for (int a = 0; a < nrh; ++a) {
    const int jsp = rt->rhs[r][a];
#pragma unroll
    for (int k = 0; k < N; ++k) sw[k] += (k == jsp) ? sg : 0.0;   // products gain
  }
  return true;
}

}  // namespace lane
}  // namespace pnp
