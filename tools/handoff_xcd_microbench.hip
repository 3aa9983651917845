// What does it cost to hand a 64 KB row from 32 producers to 32 consumers once per "timestep" (the LSTM recurrence's
// hand-off, csrc/lstm.hip) when the 32 workgroups of a group sit on ONE XCD and store plain (the line stays in that
// XCD's L2) compared with the placement-independent protocol (sc1 write-through stores, consumers on any XCD)?
//   hipcc --offload-arch=gfx950 -O3 tools/handoff_xcd_microbench.hip -o /tmp/handoff.bin && /tmp/handoff.bin
// Every spin is bounded, every mode checks every word it reads (stale reads are counted, not assumed away).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

constexpr int kRowWords = 16384;   // 64 KB row = 32 members x 512 words
constexpr int kGroups = 8;

// mode bit 0: groups by XCD (else: blocks b .. b+31, i.e. 4 members per XCD); bit 1: plain stores (else sc1)
template <int MODE>
__global__ __launch_bounds__(256, 1) void handoff(unsigned* rows /* [groups][2][kRowWords] */, unsigned* counters /* [groups][32] */,
                                                  unsigned* census /* [8] */, unsigned* stale, unsigned* fail, int iters, int n_groups,
                                                  unsigned* sink) {
  extern __shared__ char lds[];   // ~100 KB: one workgroup per CU
  __shared__ int s_group, s_member, s_ok;
  const int tid = threadIdx.x;
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;   // HW_REG_XCC_ID[3:0]
    if (MODE & 1) {
      s_group = (int)xcc;
      s_member = (int)atomicAdd(&census[xcc], 1u);
    } else {
      s_group = blockIdx.x / 32;
      s_member = blockIdx.x % 32;
    }
    s_ok = 1;
  }
  __syncthreads();
  const int group = s_group, member = s_member;
  if (group >= n_groups || member >= 32) return;
  unsigned* row0 = rows + (size_t)group * 2 * kRowWords;
  unsigned* cnt = counters + group * 32;
  unsigned acc = 0, bad = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned* row = row0 + (it & 1) * kRowWords;
    // produce: 2 KB slice, 8 bytes per thread
    const __amdgpu_buffer_rsrc_t rs = rsrc(row);
    const u32x2 v = {(unsigned)(it + 1), (unsigned)(it + 1)};
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, (member * 512 + tid * 2) * 4, 0, (MODE & 2) ? 0 : 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = 32u * (unsigned)(it + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 2000000u) { atomicAdd(fail, 1u); s_ok = 0; break; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __syncthreads();
    if (!s_ok) break;
    // consume: the whole row, 16 x 16 bytes per thread, sc1 loads (bypass this CU's L1)
    u32x4 r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + 256 * i) * 16, 0, 16);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc += r[i][0] ^ r[i][3];
      bad += (r[i][0] != (unsigned)(it + 1)) + (r[i][1] != (unsigned)(it + 1)) + (r[i][2] != (unsigned)(it + 1)) +
             (r[i][3] != (unsigned)(it + 1));
    }
    __syncthreads();
  }
  if (bad) atomicAdd(stale, bad);
  if (acc == 0x12345u) sink[0] = acc;
  (void)lds;
}

template <int MODE>
void run(const char* name, int n_groups) {
  unsigned *rows, *counters, *census, *stale, *fail, *sink;
  hipMalloc(&rows, (size_t)kGroups * 2 * kRowWords * 4);
  hipMalloc(&counters, kGroups * 32 * 4);
  hipMalloc(&census, 32);
  hipMalloc(&stale, 4); hipMalloc(&fail, 4); hipMalloc(&sink, 4);
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(handoff<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  float best = 1e9f;
  unsigned st = 0, fl = 0, cen[8];
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(rows, 0, (size_t)kGroups * 2 * kRowWords * 4);
    hipMemset(counters, 0, kGroups * 32 * 4);
    hipMemset(census, 0, 32); hipMemset(stale, 0, 4); hipMemset(fail, 0, 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(handoff<MODE>, dim3(256), dim3(256), 100 * 1024, 0, rows, counters, census, stale, fail, iters, n_groups, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
    unsigned s1, f1;
    hipMemcpy(&s1, stale, 4, hipMemcpyDeviceToHost); hipMemcpy(&f1, fail, 4, hipMemcpyDeviceToHost);
    hipMemcpy(cen, census, 32, hipMemcpyDeviceToHost);
    st += s1; fl += f1;
  }
  printf("%-44s groups %d: %6.3f us per hand-off   stale words %u   spin caps %u", name, n_groups, best * 1e3 / iters, st, fl);
  if (MODE & 1) printf("   workgroups per XCD: %u %u %u %u %u %u %u %u", cen[0], cen[1], cen[2], cen[3], cen[4], cen[5], cen[6], cen[7]);
  printf("\n");
  hipFree(rows); hipFree(counters); hipFree(census); hipFree(stale); hipFree(fail); hipFree(sink);
}

int main() {
  for (int g : {8, 2}) {
    run<0>("spread over the XCDs, sc1 stores (today)", g);
    run<1>("one XCD per group, sc1 stores", g);
    run<3>("one XCD per group, plain stores", g);
    run<2>("spread over the XCDs, plain stores (invalid)", g);
  }
  return 0;
}
