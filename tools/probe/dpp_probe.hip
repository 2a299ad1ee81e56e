#include <hip/hip_runtime.h>
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ double dpp_f64(double old, double x) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, ROWMASK, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, ROWMASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__global__ void k(double* g) {
  const int lane = threadIdx.x;
  double v = g[lane];
  double s = v;
  s += dpp_f64<0x111>(0.0, s);
  s += dpp_f64<0x112>(0.0, s);
  s += dpp_f64<0x114>(0.0, s);
  s += dpp_f64<0x118>(0.0, s);
  s += dpp_f64<0x142, 0xa>(0.0, s);
  s += dpp_f64<0x143, 0xc>(0.0, s);
  g[64 + lane] = s;                                  // inclusive scan
  g[128 + lane] = dpp_f64<0x138>(-1.0, v);           // wave_shr:1  (lane i <- i-1, lane 0 keeps -1)
  g[192 + lane] = dpp_f64<0x130>(-2.0, v);           // wave_shl:1  (lane i <- i+1, lane 63 keeps -2)
}
int main() {
  double h[256], *d;
  for (int i = 0; i < 64; ++i) h[i] = i + 1;
  hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    if (h[64 + i] != (i + 1) * (i + 2) / 2) bad++;
    if (h[128 + i] != (i == 0 ? -1.0 : (double)i)) bad++;
    if (h[192 + i] != (i == 63 ? -2.0 : (double)(i + 2))) bad++;
  }
  printf("dpp probe: %d mismatches; scan[63]=%g shr[1]=%g shl[0]=%g\n", bad, h[127], h[129], h[192]);
  return bad != 0;
}
