// dev probe: accuracy of v_rcp_f64 and of 1/2 Newton refinements on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double b = x[i];
  double r = __builtin_amdgcn_rcp(b);
  r0[i] = r;
  double e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
  r1[i] = r;
  e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
  r2[i] = r;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); h[i] = std::ldexp(0.5 + 0.5 * u, (int)(s % 40) - 20) * ((s >> 60) & 1 ? -1 : 1); }
  double *x, *r0, *r1, *r2;
  hipMalloc(&x, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8);
  hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(x, r0, r1, r2, n);
  std::vector<double> a(n), b(n), c(n);
  hipMemcpy(a.data(), r0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), r1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), r2, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; ++i) { long double t = 1.0L / (long double)h[i]; m0 = fmax(m0, fabs((double)((a[i] - t) / t))); m1 = fmax(m1, fabs((double)((b[i] - t) / t))); m2 = fmax(m2, fabs((double)((c[i] - t) / t))); }
  printf("max rel err: v_rcp_f64 %.3e  +1NR %.3e  +2NR %.3e\n", m0, m1, m2);
  return 0;
}
