// How much does a grid-wide barrier cost on MI355X?  N workgroups (one wave each, or 256 threads) meet at an atomic
// counter in device memory `iters` times.  Every spin loop has an iteration cap, so the kernel always terminates.
// hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_microbench.hip -o tools/grid_barrier.bin && tools/grid_barrier.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void barrier_loop(unsigned* counter, unsigned* fail, int iters, float* sink, const float* data, int work) {
  const unsigned n = gridDim.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    // a little per-step "work": read `work` floats of shared data (stands in for the h broadcast)
    for (int i = threadIdx.x; i < work; i += blockDim.x) acc += data[i];
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(counter, 1u);
      const unsigned target = n * (unsigned)(it + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 4000000u) { atomicAdd(fail, 1u); break; }
      }
    }
    __syncthreads();
  }
  if (acc == 12345.f) sink[0] = acc;
}

int main() {
  unsigned *counter, *fail;
  float *sink, *data;
  hipMalloc(&counter, 4); hipMalloc(&fail, 4); hipMalloc(&sink, 4); hipMalloc(&data, 1 << 20);
  hipMemset(data, 0, 1 << 20);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000;
  for (int threads : {64, 256}) {
    for (int nblk : {32, 64, 128, 256, 512}) {
      for (int work : {0, 16384}) {
        hipMemset(counter, 0, 4); hipMemset(fail, 0, 4);
        hipLaunchKernelGGL(barrier_loop, dim3(nblk), dim3(threads), 0, 0, counter, fail, 10, sink, data, work);  // warm
        hipDeviceSynchronize();
        hipMemset(counter, 0, 4);
        hipEventRecord(a);
        hipLaunchKernelGGL(barrier_loop, dim3(nblk), dim3(threads), 0, 0, counter, fail, iters, sink, data, work);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        unsigned f;
        hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
        printf("threads %3d blocks %3d work %5d floats: %.3f us per barrier step%s\n", threads, nblk, work, ms * 1e3 / iters,
               f ? "  (SPIN CAP HIT: not all blocks were resident)" : "");
      }
    }
  }
  return 0;
}
