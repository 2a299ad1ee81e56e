// Persistent waves copying 8-KiB rows: static round-robin row assignment against a work queue (atomic counter, rows handed out in
// address order as waves become free -- the order the hardware dispatcher gives a non-persistent grid).  2 GiB in, 2 GiB out.
// build: hipcc --offload-arch=gfx950 -O3 -o access_pattern4 access_pattern4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void rows_static(const u4* __restrict__ in, u4* __restrict__ out, long rows) {
  for (long r = blockIdx.x; r < rows; r += gridDim.x) {
    u4 t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = in[r * 512 + q * 64 + threadIdx.x];
#pragma unroll
    for (int q = 0; q < 8; ++q) out[r * 512 + q * 64 + threadIdx.x] = t[q];
  }
}
__global__ __launch_bounds__(64) void rows_queue(const u4* __restrict__ in, u4* __restrict__ out, long rows, unsigned* counter) {
  long r = blockIdx.x;
  while (r < rows) {
    unsigned nx = 0;
    if (threadIdx.x == 0) nx = atomicAdd(counter, 1u);
    const long rn = (long)__builtin_amdgcn_readfirstlane((int)nx) + gridDim.x;
    u4 t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = in[r * 512 + q * 64 + threadIdx.x];
#pragma unroll
    for (int q = 0; q < 8; ++q) out[r * 512 + q * 64 + threadIdx.x] = t[q];
    r = rn;
  }
}
__global__ __launch_bounds__(64) void rows_nonpersistent(const u4* __restrict__ in, u4* __restrict__ out) {
  const long r = blockIdx.x;
  u4 t[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) t[q] = in[r * 512 + q * 64 + threadIdx.x];
#pragma unroll
  for (int q = 0; q < 8; ++q) out[r * 512 + q * 64 + threadIdx.x] = t[q];
}

template <class F>
static void timeit(const char* tag, F launch, long bytes) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 7; ++rep) {
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 1 && ms < best) best = ms;
  }
  printf("%-52s %.3f ms  %.0f GB/s  %.3f of 8 TB/s\n", tag, best, 2.0 * bytes / (best * 1e-3) / 1e9, 2.0 * bytes / (best * 1e-3) / 8e12);
}

int main() {
  const long bytes = 2L << 30, rows = bytes / 8192;
  u4 *in, *out;
  unsigned* counter;
  (void)hipMalloc(&in, bytes);
  (void)hipMalloc(&out, bytes);
  (void)hipMalloc(&counter, 4);
  (void)hipMemset(in, 1, bytes);
  (void)hipMemset(out, 0, bytes);
  for (int wpc : {6, 8, 12, 16}) {
    char tag[96];
    snprintf(tag, sizeof tag, "static round-robin, %d waves/CU", wpc);
    timeit(tag, [&] { hipLaunchKernelGGL(rows_static, dim3(256 * wpc), dim3(64), 0, 0, in, out, rows); }, bytes);
    snprintf(tag, sizeof tag, "work queue, %d waves/CU", wpc);
    timeit(tag, [&] { (void)hipMemsetAsync(counter, 0, 4, 0); hipLaunchKernelGGL(rows_queue, dim3(256 * wpc), dim3(64), 0, 0, in, out, rows, counter); }, bytes);
  }
  timeit("non-persistent, one wave per row", [&] { hipLaunchKernelGGL(rows_nonpersistent, dim3(rows), dim3(64), 0, 0, in, out); }, bytes);
  return 0;
}
