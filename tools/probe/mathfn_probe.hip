// dev probe: accuracy of expm1_sc / log1p_sc (catint_amd/csrc/pnp_math.h) against the host libm, in ulp.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/mathfn_probe.hip -o tools/probe/mathfn_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../catint_amd/csrc/pnp_math.h"

__global__ void k_expm1(const double* x, double* y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = pnp::expm1_sc(x[i]);
}
__global__ void k_log1p(const double* x, double* y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = pnp::log1p_sc(x[i]);
}

static double ulps(double got, double ref) {
  if (got == ref) return 0.0;
  const double u = std::nextafter(std::fabs(ref), INFINITY) - std::fabs(ref);
  return std::fabs(got - ref) / u;
}

int main() {
  const int n = 1 << 20;
  std::vector<double> xe(n), xl(n), ye(n), yl(n);
  for (int i = 0; i < n; ++i) {
    const double t = (i + 0.5) / n;
    // expm1: |u| in [0.05, 700], both signs, log-spaced
    const double mag = 0.05 * std::pow(700.0 / 0.05, t);
    xe[i] = (i & 1) ? mag : -mag;
    // log1p: x = -f, f in [1e-12, 1 - 1e-12]
    const double f = (i & 1) ? std::pow(10.0, -12.0 * t) : 1.0 - std::pow(10.0, -12.0 * t);
    xl[i] = -f;
  }
  double *dx, *dy;
  hipMalloc(&dx, n * sizeof(double));
  hipMalloc(&dy, n * sizeof(double));
  hipMemcpy(dx, xe.data(), n * sizeof(double), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_expm1, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
  hipMemcpy(ye.data(), dy, n * sizeof(double), hipMemcpyDeviceToHost);
  hipMemcpy(dx, xl.data(), n * sizeof(double), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_log1p, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
  hipMemcpy(yl.data(), dy, n * sizeof(double), hipMemcpyDeviceToHost);
  double me = 0, ml = 0;
  int ie = 0, il = 0;
  for (int i = 0; i < n; ++i) {
    const double e = ulps(ye[i], std::expm1(xe[i])), l = ulps(yl[i], std::log1p(xl[i]));
    if (e > me) { me = e; ie = i; }
    if (l > ml) { ml = l; il = i; }
  }
  printf("expm1_sc: max %.2f ulp at u = %.17g (got %.17g, ref %.17g)\n", me, xe[ie], ye[ie], std::expm1(xe[ie]));
  printf("log1p_sc: max %.2f ulp at x = %.17g (got %.17g, ref %.17g)\n", ml, xl[il], yl[il], std::log1p(xl[il]));
  return (me < 4.0 && ml < 4.0) ? 0 : 1;
}
