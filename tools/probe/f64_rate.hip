// dev probe: issue cost of fp64 VALU instructions on gfx950 (cycles per wave-instruction, by waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, int iters) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = 1.0 + threadIdx.x * 1e-3 + i;
  const double m = 1.0000001, c = 1e-9;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = __builtin_fma(a[i], m, c);            // independent FMAs (8 chains)
      if (MODE == 1) a[i] = __builtin_amdgcn_rcp(a[i]) + 1.5;      // rcp + add
      if (MODE == 2) a[0] = __builtin_fma(a[0], m, c);             // one dependent chain
      if (MODE == 3) a[i] = a[i] * m;                              // mul
      if (MODE == 4) a[i] = (double)__builtin_amdgcn_rcpf((float)a[i]) + 1.5;   // f32 seed path + add
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int per_iter) {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 4096 * 256 * 8); hipMalloc(&cyc, 4096 * 8);
  const int iters = 2000;
  for (int wps : {1, 2, 4}) {                 // waves per SIMD: blocks of 64*4*wps threads, one block per CU
    const int threads = 64 * 4 * wps;
    k<MODE><<<256, threads>>>(out, cyc, iters);
    k<MODE><<<256, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= 256;
    printf("%-28s waves/SIMD=%d: %.2f cycles per wave-instr-group elem (per SIMD: %.2f cycles/instr)\n", name, wps,
           avg / (iters * 8.0), avg / (iters * 8.0 * per_iter) / wps);
  }
}
int main() {
  run<0>("fma f64 x8 independent", 1);
  run<2>("fma f64 dependent chain", 1);
  run<3>("mul f64 x8 independent", 1);
  run<1>("rcp f64 + add", 2);
  run<4>("cvt+rcp f32+cvt + add", 4);
  return 0;
}
