// Read + write streaming ceiling on MI355X for the streaming kernel's row traffic: what do non-temporal hints, more bytes in flight
// per wave and larger workgroups buy?  (coalesced 16-B-per-lane accesses, 2 GiB in, 2 GiB out)
// build: hipcc --offload-arch=gfx950 -O3 -o access_pattern2 access_pattern2.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// AUXL / AUXS: cache-policy bits of the buffer instructions (bit 0 sc0, bit 1 nt, bit 4 sc1 on gfx94x/95x)
template <int ROWS, int AUXL, int AUXS>
__global__ __launch_bounds__(64) void copy_rows(const double* __restrict__ in, double* __restrict__ out, long rows) {
  const int lane = threadIdx.x;
  for (long r = (long)blockIdx.x * ROWS; r < rows; r += (long)gridDim.x * ROWS) {
    u4 t[ROWS][8];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const __amdgpu_buffer_rsrc_t s = rsrc(in + (r + j) * 1024, 8192);
#pragma unroll
      for (int q = 0; q < 8; ++q) t[j][q] = __builtin_amdgcn_raw_buffer_load_b128(s, lane * 16 + q * 1024, 0, AUXL);
    }
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const __amdgpu_buffer_rsrc_t d = rsrc(out + (r + j) * 1024, 8192);
#pragma unroll
      for (int q = 0; q < 8; ++q) __builtin_amdgcn_raw_buffer_store_b128(t[j][q], d, lane * 16 + q * 1024, 0, AUXS);
    }
  }
}

template <int ROWS, int AUXL, int AUXS>
static void run(const char* tag, const double* in, double* out, long rows, hipEvent_t e0, hipEvent_t e1) {
  for (int wpc : {8, 16}) {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((copy_rows<ROWS, AUXL, AUXS>), dim3(256 * wpc), dim3(64), 0, 0, in, out, rows);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep > 1 && ms < best) best = ms;
    }
    printf("%-44s waves/CU %2d: %.3f ms  %.0f GB/s  %.3f of 8 TB/s\n", tag, wpc, best, 2.0 * rows * 8192 / (best * 1e-3) / 1e9,
           2.0 * rows * 8192 / (best * 1e-3) / 8e12);
  }
}

int main() {
  const long rows = 262144;
  double *in, *out;
  (void)hipMalloc(&in, rows * 8192);
  (void)hipMalloc(&out, rows * 8192);
  (void)hipMemset(in, 1, rows * 8192);
  (void)hipMemset(out, 0, rows * 8192);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  run<1, 0, 0>("1 row in flight, default policy", in, out, rows, e0, e1);
  run<2, 0, 0>("2 rows in flight, default policy", in, out, rows, e0, e1);
  run<1, 2, 2>("1 row, nt loads + nt stores", in, out, rows, e0, e1);
  run<2, 2, 2>("2 rows, nt loads + nt stores", in, out, rows, e0, e1);
  run<1, 0, 2>("1 row, nt stores only", in, out, rows, e0, e1);
  run<1, 2, 0>("1 row, nt loads only", in, out, rows, e0, e1);
  run<1, 17, 17>("1 row, sc0 sc1 loads + stores", in, out, rows, e0, e1);
  run<1, 3, 3>("1 row, sc0 nt loads + stores", in, out, rows, e0, e1);
  // in place (the streaming kernel overwrites the state it read)
  run<1, 0, 0>("in place, default policy", in, in, rows, e0, e1);
  run<1, 2, 2>("in place, nt loads + nt stores", in, in, rows, e0, e1);
  return 0;
}
