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
// (CATINT_LANE_PIVOT_LIMIT overrides it: tests set it below one so that every lane trips the monitor)
inline double lane_pivot_limit_from_env() {
  const char* e = getenv("CATINT_LANE_PIVOT_LIMIT");
  const double v = e ? atof(e) : 0.0;
  return v > 0.0 ? v : PIVOT_GROWTH_LIMIT;
}

struct LaneParams {
  double sig[PNP_NEWTON_MAX_SPECIES], peq[PNP_NEWTON_MAX_SPECIES];
};

}  // namespace lane
}  // namespace pnp
