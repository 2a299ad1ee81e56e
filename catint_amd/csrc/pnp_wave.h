// Wave-level device helpers shared by the timestep kernels (pnp_kernels.hip, pnp_stream.hip): 16-byte buffer-resource row
// access, blocked prefix scans, the per-wave tridiagonal solver (per-lane Thomas + 64-row parallel cyclic reduction), DPP
// cross-lane moves.  gfx950 / MI355X only.
#pragma once
#include "pnp_internal.h"

namespace pnp {

// native 16-byte vector (the HIP vector class wrapper keeps register arrays from being promoted)
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double fast_rcp(double x) {
  // v_rcp_f64 seed (measured on gfx950: 4.6e-8 relative) + one Newton step -> 2.2e-15 relative
  // (tools/probe/rcp_probe.hip).  The pivots of the diagonally dominant systems solved here are
  // O(1) and well away from 0, so none of hipcc's div_scale/div_fixup range handling is needed; the
  // parity tolerance against the reference's LU (1e-9) leaves six orders of magnitude of margin.
  double r = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}

// full-accuracy variant (two Newton steps, 1.1e-16) for the few scalar, wave-uniform quotients
__device__ __forceinline__ double fast_rcp2(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}

// wave-private LDS hand-off: LDS operations of one wave execute in order, so only the compiler
// must be kept from moving accesses across the hand-off point.
__device__ __forceinline__ void lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// workgroup LDS hand-off that does NOT drain outstanding global stores (LDS-only fences).
template <int W>
__device__ __forceinline__ void wg_sync() {
  if constexpr (W == 1) {
    lds_sync();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  }
}

// padded LDS index of grid point i: one pad double per P points makes the blocked access
// (lane stride P doubles) hit 64 distinct banks for ds_read_b64 / ds_write_b64.
template <int P>
__device__ __forceinline__ int pidx(int i) {
  return i + (int)((unsigned)i / (unsigned)P);   // i >= 0 always; unsigned keeps it a single shift
}

// LDS doubles per staged row: covers the largest pitch of this P (64P+16), one dummy slot for
// masked stores, and the 384-double exchange area of the cyclic reduction.
template <int P>
__host__ __device__ constexpr int rowbuf_doubles() {
  constexpr int CAP = 128 * (P / 2 + 1);          // whole 1-KiB chunks covering the largest pitch 64P+16
  constexpr int need = CAP + CAP / P + 4;
  constexpr int n = need < 388 ? 388 : need;      // >= the 384-double exchange area of the cyclic reduction
  return (n + 1) & ~1;
}

// coalesced global row (16 B per lane, 1 KiB per wave instruction) -> registers -> padded LDS row.
// Split in two so that the HBM latency of a row overlaps the work issued between the halves.
template <int P>
struct RowRegs {
  static constexpr int IT = P / 2 + 1;  // ldx <= 64*P + 16
  d2 t[IT];
};

// Pair addresses are affine in (lane, it): e = 2*lane + 128*it and, because P divides 128,
// pidx(e) = pidx(2*lane) + it*(128 + 128/P); pidx(e+1) = pidx(e) + 1 for even P (+2 for P = 1).
template <int P>
__device__ __forceinline__ int pair_slot(int lane_slot, int it) {
  return lane_slot + it * (128 + 128 / P);
}
template <int P>
constexpr int PAIR_STEP = (P == 1) ? 2 : 1;

// Global rows are accessed through buffer resources (one 128-bit descriptor per row, built from
// wave-uniform values): the hardware range check returns 0 for loads and drops stores beyond the row
// pitch, so the partial last chunk of an odd pitch needs neither a lane predicate nor a branch, and
// the per-lane address is one 32-bit offset VGPR plus an immediate.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const double* row, int ldx) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(row), 0, ldx * 8, 0x00020000);
}

template <int P>
__device__ __forceinline__ void load_row_issue(__amdgpu_buffer_rsrc_t r, RowRegs<P>& rr, int tid_) {
#pragma unroll
  for (int it = 0; it < RowRegs<P>::IT; ++it)
    rr.t[it] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, tid_ * 16 + it * 1024, 0, 0));
}

template <int P>
__device__ __forceinline__ void load_row_commit(const RowRegs<P>& rr, double* buf, int tid_) {
  const int ls = pidx<P>(2 * tid_);
#pragma unroll
  for (int it = 0; it < RowRegs<P>::IT; ++it) {
    buf[pair_slot<P>(ls, it)] = rr.t[it].x;
    buf[pair_slot<P>(ls, it) + PAIR_STEP<P>] = rr.t[it].y;
  }
}

template <int P>
__device__ __forceinline__ void load_row(const double* __restrict__ g, double* buf, int ldx, int lane) {
  RowRegs<P> rr;
  load_row_issue<P>(row_rsrc(g, ldx), rr, lane);
  load_row_commit<P>(rr, buf, lane);
}

// AUX: cache-policy bits of the buffer instruction (0 default; 2 = nt: streamed, do not keep)
template <int P, int AUX = 0>
__device__ __forceinline__ void store_row(double* __restrict__ g, const double* buf, int ldx, int tid_) {
  const int ls = pidx<P>(2 * tid_);
  const __amdgpu_buffer_rsrc_t r = row_rsrc(g, ldx);
#pragma unroll
  for (int it = 0; it < RowRegs<P>::IT; ++it) {
    d2 t;
    t.x = buf[pair_slot<P>(ls, it)];
    t.y = buf[pair_slot<P>(ls, it) + PAIR_STEP<P>];
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, t), r, tid_ * 16 + it * 1024, 0, AUX);
  }
}

// Inclusive scan of the wave's 64*P blocked values (lane-major order), in place. REV = suffix scan.
// base = what the lane's first (REV: last) element inherited from the other lanes, so the exclusive
// scan is x[j-1] (REV: x[j+1]) inside the lane and `base` at its edge.
// The cross-lane part is a Hillis-Steele scan through a wave-private LDS strip X (128 doubles: the
// 64 lane slots sit between two 32-slot zero guards, so no lane needs an edge predicate).
template <int P, bool REV>
__device__ __forceinline__ void blocked_scan(double (&x)[P], double* X, int lane, double& total, double& base) {
  double* XS = X + 32 + lane;
  XS[(lane < 32) ? -32 : 32] = 0.0;   // guards
  double inc;
  if (!REV) {
#pragma unroll
    for (int j = 1; j < P; ++j) x[j] += x[j - 1];
    inc = x[P - 1];
  } else {
#pragma unroll
    for (int j = P - 2; j >= 0; --j) x[j] += x[j + 1];
    inc = x[0];
  }
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    XS[0] = inc;
    lds_sync();
    const double u = REV ? XS[s] : XS[-s];
    lds_sync();
    inc += u;
  }
  XS[0] = inc;
  lds_sync();
  total = REV ? X[32] : X[32 + 63];
  base = REV ? XS[1] : XS[-1];
  lds_sync();
#pragma unroll
  for (int j = 0; j < P; ++j) x[j] += base;
}

// Forward inclusive blocked scan of x[] with a second, scalar-per-lane quantity w summed over the wave
// in the same LDS round trips (wtotal = sum over lanes of w).  Used by the Dirichlet-Dirichlet Poisson
// fast path.  X needs 256 doubles.
template <int P>
__device__ __forceinline__ void blocked_scan_sum(double (&x)[P], double w, double* X, int lane, double& total,
                                                 double& base, double& wtotal) {
  double* XS = X + 32 + lane;
  double* XW = XS + 128;
  if (lane < 32) {   // only the left guards are read by a forward scan
    XS[-32] = 0.0;
    XW[-32] = 0.0;
  }
#pragma unroll
  for (int j = 1; j < P; ++j) x[j] += x[j - 1];
  double inc = x[P - 1];
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    XS[0] = inc;
    XW[0] = w;
    lds_sync();
    const double u = XS[-s];
    const double uw = XW[-s];
    lds_sync();
    inc += u;
    w += uw;
  }
  XS[0] = inc;
  XW[0] = w;
  lds_sync();
  total = X[32 + 63];
  wtotal = X[128 + 32 + 63];
  base = XS[-1];
  lds_sync();
#pragma unroll
  for (int j = 0; j < P; ++j) x[j] += base;
}

// ------------------------------------------------------------------------------------------------
// G independent tridiagonal systems of 64*P unknowns each, held P per lane, rows pre-scaled to unit
// diagonal:     a[g][j]*x[r-1] + x[r] + c[g][j]*x[r+1] = d[g][j],   r = lane*P + j
// (a of the wave's first row must be 0; beyond lane 63 the exchange reads zero guards).
// Everything is done in place in the three [G][P] arrays; the solutions overwrite d.
// The G systems (the species a wave advances together -- they are independent because the reference
// lags the potential) are interleaved statement by statement, so every dependent chain below and
// every LDS round trip is shared by G systems: instruction-level parallelism instead of occupancy.
// Stage 1 (per lane, registers): Thomas elimination of the P-1 interior rows against the two
//   interface unknowns yL = x[last row of lane-1] and y = x[last row of this lane].
// Stage 2 (across the wave): the 64 interface rows form a tridiagonal system solved by parallel
//   cyclic reduction in log2(64) = 6 steps.  Neighbour rows at distance s are exchanged through
//   a wave-private LDS strip per system (3 arrays of 128 doubles: 32 zero guard slots on either
//   side of the 64 lanes), so one step costs 3 LDS writes + 6 reads and no select for the edges.
// Stage 3: back-substitution of the interior rows.
// X + g*XSTRIDE is the strip of system g.
// ------------------------------------------------------------------------------------------------
template <int P, int G>
__device__ __forceinline__ void tridiag_wave(double (&a)[G][P], double (&c)[G][P], double (&d)[G][P], double* X,
                                             int XSTRIDE, int lane) {
  // in place: a[i] -> Vs_i, c[i] -> Ws_i, d[i] -> ds_i with x_i = ds_i - Vs_i*yL - Ws_i*y  (i < P-1)
  double ra[G], rc[G], rd[G];
  double* XA = X + 32 + lane;
  double* XC = XA + 128;
  double* XD = XA + 256;
  {  // zero guards: slots [0,32) and [96,128) of each array
    const int gofs = (lane < 32) ? -32 : 32;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      XA[g * XSTRIDE + gofs] = 0.0;
      XC[g * XSTRIDE + gofs] = 0.0;
      XD[g * XSTRIDE + gofs] = 0.0;
    }
  }
  if constexpr (P == 1) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      ra[g] = a[g][0];
      rc[g] = c[g][0];
      rd[g] = d[g][0];
    }
  } else {
#pragma unroll
    for (int i = 1; i < P - 1; ++i) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const double ai = a[g][i];
        const double bb = __builtin_fma(-ai, c[g][i - 1], 1.0);
        const double dd = __builtin_fma(-ai, d[g][i - 1], d[g][i]);
        const double vv = -ai * a[g][i - 1];
        const double r = fast_rcp(bb);
        a[g][i] = vv * r;
        d[g][i] = dd * r;
        c[g][i] = c[g][i] * r;
      }
    }
#pragma unroll
    for (int i = P - 3; i >= 0; --i) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const double cs = c[g][i];
        d[g][i] = __builtin_fma(-cs, d[g][i + 1], d[g][i]);
        a[g][i] = __builtin_fma(-cs, a[g][i + 1], a[g][i]);
        c[g][i] = -cs * c[g][i + 1];
      }
    }
    // first interior row of the next lane closes this lane's interface row
#pragma unroll
    for (int g = 0; g < G; ++g) {
      XA[g * XSTRIDE] = a[g][0];
      XC[g * XSTRIDE] = c[g][0];
      XD[g * XSTRIDE] = d[g][0];
    }
    lds_sync();
    double Vn0[G], Wn0[G], dn0[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {   // lane 63 reads the zero guard
      Vn0[g] = XA[g * XSTRIDE + 1];
      Wn0[g] = XC[g * XSTRIDE + 1];
      dn0[g] = XD[g * XSTRIDE + 1];
    }
    lds_sync();
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const double aL = a[g][P - 1], cL = c[g][P - 1];
      double rb = __builtin_fma(-aL, c[g][P - 2], 1.0);
      rb = __builtin_fma(-cL, Vn0[g], rb);
      const double rr = fast_rcp(rb);
      double t = __builtin_fma(-aL, d[g][P - 2], d[g][P - 1]);
      t = __builtin_fma(-cL, dn0[g], t);
      ra[g] = (-aL * a[g][P - 2]) * rr;
      rc[g] = (-cL * Wn0[g]) * rr;
      rd[g] = t * rr;
    }
  }
  // parallel cyclic reduction over the 64 interface rows (unit diagonal kept by renormalising)
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      XA[g * XSTRIDE] = ra[g];
      XC[g * XSTRIDE] = rc[g];
      XD[g * XSTRIDE] = rd[g];
    }
    lds_sync();
    double aL[G], aR[G], cL[G], cR[G], dL[G], dR[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      aL[g] = XA[g * XSTRIDE - s];
      aR[g] = XA[g * XSTRIDE + s];
      cL[g] = XC[g * XSTRIDE - s];
      cR[g] = XC[g * XSTRIDE + s];
      dL[g] = XD[g * XSTRIDE - s];
      dR[g] = XD[g * XSTRIDE + s];
    }
    lds_sync();
#pragma unroll
    for (int g = 0; g < G; ++g) {
      double nb = __builtin_fma(-ra[g], cL[g], 1.0);
      nb = __builtin_fma(-rc[g], aR[g], nb);
      double nd = __builtin_fma(-ra[g], dL[g], rd[g]);
      nd = __builtin_fma(-rc[g], dR[g], nd);
      const double na = -ra[g] * aL[g];
      const double nc = -rc[g] * cR[g];
      const double rr = fast_rcp(nb);
      ra[g] = na * rr;
      rc[g] = nc * rr;
      rd[g] = nd * rr;
    }
  }
  if constexpr (P > 1) {
#pragma unroll
    for (int g = 0; g < G; ++g) XA[g * XSTRIDE] = rd[g];
    lds_sync();
    double yL[G];
#pragma unroll
    for (int g = 0; g < G; ++g) yL[g] = XA[g * XSTRIDE - 1];   // lane 0 reads the zero guard
    lds_sync();
#pragma unroll
    for (int i = 0; i < P - 1; ++i) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const double t = __builtin_fma(-a[g][i], yL[g], d[g][i]);
        d[g][i] = __builtin_fma(-c[g][i], rd[g], t);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) d[g][P - 1] = rd[g];
}

template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ double dpp_f64(double old, double x) {
  // lanes whose DPP source is invalid, or whose row is masked off, keep `old`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, ROWMASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, ROWMASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_prev_lane(double old_lane0, double x) { return dpp_f64<0x138>(old_lane0, x); }   // wave_shr:1
__device__ __forceinline__ double from_next_lane(double old_lane63, double x) { return dpp_f64<0x130>(old_lane63, x); }  // wave_shl:1

// inclusive prefix sum over the 64 lanes (LLVM's GFX9 wave-scan sequence)
__device__ __forceinline__ double wave_scan_incl(double v) {
  v += dpp_f64<0x111>(0.0, v);        // row_shr:1
  v += dpp_f64<0x112>(0.0, v);        // row_shr:2
  v += dpp_f64<0x114>(0.0, v);        // row_shr:4
  v += dpp_f64<0x118>(0.0, v);        // row_shr:8
  v += dpp_f64<0x142, 0xa>(0.0, v);   // row_bcast:15 into rows 1,3
  v += dpp_f64<0x143, 0xc>(0.0, v);   // row_bcast:31 into rows 2,3
  return v;
}

__device__ __forceinline__ double read_lane(double v, int l) {   // l wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// value of a blocked array at wave-uniform interior index q (lane q/P, row q%P).  Written with per-lane
// compares on purpose: a wave-uniform register index makes LLVM spill the array to scratch.
template <int P>
__device__ __forceinline__ double pick_blocked(const double (&a)[P], int r0, int q) {
  double v = 0.0;
#pragma unroll
  for (int j = 0; j < P; ++j) v = (r0 + j == q) ? a[j] : v;
  return read_lane(v, q / P);
}

// window of P+2 doubles starting at element lane*P of a row (P even: 16-B aligned pieces)
template <int P, int AUX = 0>
__device__ __forceinline__ void load_window(__amdgpu_buffer_rsrc_t r, double (&w)[P + 2], int lane) {
#pragma unroll
  for (int q = 0; q < (P + 2) / 2; ++q) {
    const d2 t = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, lane * (P * 8) + q * 16, 0, AUX));
    w[2 * q] = t.x;
    w[2 * q + 1] = t.y;
  }
}

// store the P own rows (grid lane*P+1 ..) of a blocked array; the resource ends before the bulk point
template <int P>
__device__ __forceinline__ void store_rows(__amdgpu_buffer_rsrc_t r, const double (&x)[P], int lane) {
#pragma unroll
  for (int q = 0; q < P / 2; ++q) {
    d2 t;
    t.x = x[2 * q];
    t.y = x[2 * q + 1];
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, t), r, lane * (P * 8) + 8 + q * 16, 0, 0);
  }
}

}  // namespace pnp
