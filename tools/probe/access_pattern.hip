// Micro-benchmark: does the streaming kernel's blocked window load (lane l reads 16-byte pieces at byte offset 128 l + 16 q of an
// 8-KiB row: 64 different cache lines per wave instruction, each line consumed over 8 instructions) cost HBM bandwidth against the
// coalesced form (lane l reads 16 bytes at 16 l + 1024 q: 8 whole lines per instruction)?  Rows of 1024 doubles, one wave per row at
// a time, persistent grid, stores coalesced in both variants, 2 GiB in, 2 GiB out.
// build: hipcc --offload-arch=gfx950 -O3 -o access_pattern access_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(64) void copy_rows(const double* __restrict__ in, double* __restrict__ out, long rows) {
  const int lane = threadIdx.x;
  for (long r = blockIdx.x; r < rows; r += gridDim.x) {
    const char* src = (const char*)(in + r * 1024);
    char* dst = (char*)(out + r * 1024);
    d2 t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const long off = MODE == 0 ? (long)lane * 128 + q * 16 : (long)lane * 16 + q * 1024;
      t[q] = *(const d2*)(src + off);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) *(d2*)(dst + (long)lane * 16 + q * 1024) = t[q];
  }
}

int main(int argc, char** argv) {
  const long rows = 262144;                 // 2 GiB
  double *in, *out;
  hipMalloc(&in, rows * 8192);
  hipMalloc(&out, rows * 8192);
  hipMemset(in, 1, rows * 8192);
  hipMemset(out, 0, rows * 8192);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wpc : {8, 16, 32}) {
    for (int mode = 0; mode < 2; ++mode) {
      const int grid = 256 * wpc;
      float best = 1e30f;
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL((copy_rows<0, 1>), dim3(grid), dim3(64), 0, 0, in, out, rows);
        else hipLaunchKernelGGL((copy_rows<1, 1>), dim3(grid), dim3(64), 0, 0, in, out, rows);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 1 && ms < best) best = ms;
      }
      printf("waves/CU %2d  %s loads: %.3f ms  %.0f GB/s (read + write)  %.3f of 8 TB/s\n", wpc, mode == 0 ? "blocked  " : "coalesced", best,
             2.0 * rows * 8192 / (best * 1e-3) / 1e9, 2.0 * rows * 8192 / (best * 1e-3) / 8e12);
    }
  }
  return 0;
}
