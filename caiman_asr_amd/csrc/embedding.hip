// Gradient of the prediction network's embedding table, accumulated into the fp32 `.grad` in one launch — replaces autograd's
// embedding_dense_backward + AccumulateGrad for `torch.nn.Embedding` (training/caiman_asr_train/rnnt/model.py: the
// prediction network's `embed`): grad[v][:] += sum over the positions n with tokens[n] == v of dy[n][:], positions in
// ascending order (a fixed summation order: bit-identical from run to run, like the sort-based library kernel it replaces,
// which takes ~90 us for 2 600 tokens x 512 columns).
// One workgroup per vocabulary row: the token list goes through LDS in blocks of 1024, every thread scans it (a broadcast
// read per token) and adds the matching rows of dy for its own columns.  A row nobody emitted costs the scan only.
#include "common.h"

namespace caiman {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void embedding_grad_kernel(const int64_t* __restrict__ tokens, int64_t n,
                                                             const T* __restrict__ dy, int64_t E, float* __restrict__ grad) {
  __shared__ int tok[1024];
  __shared__ int hits;
  const int v = blockIdx.x, tid = threadIdx.x;
  // columns tid, tid + 256, ...: up to 8 per thread in registers (E <= 2048), else a loop over column groups
  for (int64_t c0 = 0; c0 < E; c0 += 2048) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool any = false;
    for (int64_t base = 0; base < n; base += 1024) {
      const int m = (int)(n - base < 1024 ? n - base : 1024);
      __syncthreads();
      if (tid == 0) hits = 0;
      __syncthreads();
      int mine = 0;
      for (int i = tid; i < m; i += 256) {
        const int t = (int)tokens[base + i];
        tok[i] = t;
        mine |= (t == v);
      }
      if (mine) hits = 1;
      __syncthreads();
      if (!hits) continue;       // uniform: read after the barrier
      any = true;
      for (int i = 0; i < m; ++i) {
        if (tok[i] != v) continue;
        const T* row = dy + (base + i) * E + c0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int64_t c = tid + 256 * j;
          if (c0 + c < E) acc[j] += static_cast<float>(row[c]);
        }
      }
    }
    if (any) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t c = c0 + tid + 256 * j;
        if (c < E) grad[(int64_t)v * E + c] += acc[j];
      }
    }
  }
}

}  // namespace
}  // namespace caiman

extern "C" int caiman_embedding_grad(const int64_t* tokens, int64_t n, const void* dy, int dtype, int64_t V, int64_t E,
                                     float* grad, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(n >= 0 && V >= 1 && V <= 0x7fffffff && E >= 1, "embedding_grad: bad extents");
  if (n == 0) return CAIMAN_OK;
  CAIMAN_CHECK(tokens && dy && grad, "embedding_grad: null pointer");
  CAIMAN_CHECK(dtype == CAIMAN_F32 || dtype == CAIMAN_BF16 || dtype == CAIMAN_F16, "embedding_grad: dy must be f32 / bf16 / f16");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)V);
  if (dtype == CAIMAN_F32) hipLaunchKernelGGL((embedding_grad_kernel<float>), grid, dim3(256), 0, s, tokens, n, (const float*)dy, E, grad);
  else if (dtype == CAIMAN_BF16) hipLaunchKernelGGL((embedding_grad_kernel<bf16_t>), grid, dim3(256), 0, s, tokens, n, (const bf16_t*)dy, E, grad);
  else hipLaunchKernelGGL((embedding_grad_kernel<f16_t>), grid, dim3(256), 0, s, tokens, n, (const f16_t*)dy, E, grad);
  return check_launch("embedding gradient");
}
