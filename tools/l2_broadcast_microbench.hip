// How fast can every CU read the same 256 KB that (a) nobody wrote, (b) the previous kernel wrote?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// each block reads `n16` uint4 starting at buf + (unique ? block*n16 : 0); 1024 threads
__global__ __launch_bounds__(1024) void reader(const uint4* __restrict__ buf, int n16, int unique, unsigned* sink) {
  const uint4* p = buf + (unique ? (size_t)blockIdx.x * n16 : 0);
  unsigned acc = 0;
  uint4 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { int idx = threadIdx.x + 1024 * i; v[i] = idx < n16 ? p[idx] : make_uint4(0,0,0,0); }
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  if (acc == 0x12345678u) sink[0] = acc;
}
// two-phase: block first touches its own 1/32 slice (by position inside its XCD group), spins on a per-group
// counter until all 32 arrived, then reads everything.
__global__ __launch_bounds__(1024) void reader_2phase(const uint4* __restrict__ buf, int n16, unsigned* counters, unsigned epoch, unsigned* sink) {
  const int grp = blockIdx.x & 7, idx_in_grp = blockIdx.x >> 3;  // observed placement: b and b+8 share an XCD (speed only)
  const int per = n16 / 32;
  unsigned acc = 0;
  if (threadIdx.x < per) { uint4 w = buf[idx_in_grp * per + threadIdx.x]; acc += w.x; }
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&counters[grp * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&counters[grp * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * 32u && ++spins < 200000u) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  uint4 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { int idx = threadIdx.x + 1024 * i; v[i] = idx < n16 ? buf[idx] : make_uint4(0,0,0,0); }
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void writer(uint4* buf, int n16, unsigned val) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) buf[i] = make_uint4(val, val + 1, val + 2, val + 3);
}

int main() {
  const int n16 = 256 * 1024 / 16;  // 256 KB
  uint4* buf; unsigned* sink; unsigned* counters;
  CK(hipMalloc(&buf, (size_t)256 * n16 * 16)); CK(hipMemset(buf, 1, (size_t)256 * n16 * 16));
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&counters, 8 * 32 * 4)); CK(hipMemset(counters, 0, 8 * 32 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto fn, int reps, int kernels_per_rep) {
    for (int i = 0; i < 20; ++i) fn(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) fn(20 + i);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s: %.2f us per iteration\n", name, ms * 1e3f / reps);
  };
  timeit("256 blocks read the SAME 256 KB, never written", [&](int) { hipLaunchKernelGGL(reader, dim3(256), dim3(1024), 0, 0, buf, n16, 0, sink); }, 300, 1);
  timeit("256 blocks read UNIQUE 256 KB each (64 MB total)", [&](int) { hipLaunchKernelGGL(reader, dim3(256), dim3(1024), 0, 0, buf, n16, 1, sink); }, 300, 1);
  timeit("256 blocks read UNIQUE 64 KB each (16 MB total, fits the 8 L2s)", [&](int) { hipLaunchKernelGGL(reader, dim3(256), dim3(1024), 0, 0, buf, n16 / 4, 1, sink); }, 300, 1);
  timeit("256 blocks read UNIQUE 32 KB each (8 MB total)", [&](int) { hipLaunchKernelGGL(reader, dim3(256), dim3(1024), 0, 0, buf, n16 / 8, 1, sink); }, 300, 1);
  timeit("256 blocks read UNIQUE 128 KB each (32 MB total)", [&](int) { hipLaunchKernelGGL(reader, dim3(256), dim3(1024), 0, 0, buf, n16 / 2, 1, sink); }, 300, 1);
  timeit("64 blocks read the SAME 256 KB", [&](int) { hipLaunchKernelGGL(reader, dim3(64), dim3(1024), 0, 0, buf, n16, 0, sink); }, 300, 1);
  timeit("8 blocks read the SAME 256 KB", [&](int) { hipLaunchKernelGGL(reader, dim3(8), dim3(1024), 0, 0, buf, n16, 0, sink); }, 300, 1);
  timeit("256 blocks read the SAME 64 KB", [&](int) { hipLaunchKernelGGL(reader, dim3(256), dim3(1024), 0, 0, buf, n16 / 4, 0, sink); }, 300, 1);
  timeit("writer(256 KB) alone", [&](int i) { hipLaunchKernelGGL(writer, dim3(n16 / 256), dim3(256), 0, 0, buf, n16, (unsigned)i); }, 300, 1);
  timeit("writer(256 KB) + 256 blocks read the SAME 256 KB", [&](int i) {
    hipLaunchKernelGGL(writer, dim3(n16 / 256), dim3(256), 0, 0, buf, n16, (unsigned)i);
    hipLaunchKernelGGL(reader, dim3(256), dim3(1024), 0, 0, buf, n16, 0, sink); }, 300, 2);
  unsigned epoch = 0;
  timeit("writer + 2-phase cooperative L2 fill reader (256 blocks)", [&](int i) {
    hipLaunchKernelGGL(writer, dim3(n16 / 256), dim3(256), 0, 0, buf, n16, (unsigned)i);
    ++epoch;
    hipLaunchKernelGGL(reader_2phase, dim3(256), dim3(1024), 0, 0, buf, n16, counters, epoch, sink); }, 300, 2);
  return 0;
}
