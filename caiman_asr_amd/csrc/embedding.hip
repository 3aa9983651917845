// Gradient of the prediction network's embedding table, accumulated into the fp32 `.grad` in one launch — replaces autograd's
// embedding_dense_backward + AccumulateGrad for `torch.nn.Embedding` (training/caiman_asr_train/rnnt/model.py: the
// prediction network's `embed`): grad[v][:] += sum over the positions n with tokens[n] == v of dy[n][:], positions in
// ascending order (a fixed summation order: bit-identical from run to run, like the sort-based library kernel it replaces,
// which takes ~90 us for 2 600 tokens x 512 columns).
// One workgroup per token POSITION n: it first looks for an earlier position with the same token (a parallel scan of the
// token list in front of it) and leaves if there is one -- so exactly one workgroup per distinct token goes on, the one at
// the token's first position.  That one walks the positions behind it in blocks of 256, compacts the matches of a block in
// ascending order (wave ballots + the waves' counts) and adds their rows of dy, every thread its own columns.
#include "common.h"

namespace caiman {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void embedding_grad_kernel(const int64_t* __restrict__ tokens, int64_t n,
                                                             const T* __restrict__ dy, int64_t V, int64_t E,
                                                             float* __restrict__ grad) {
  __shared__ int list[256];
  __shared__ int wave_count[4];
  const int64_t pos = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t v = tokens[pos];
  if (v < 0 || v >= V) return;                       // uniform
  int earlier = 0;
  for (int64_t base = 0; base < pos; base += 256) {
    const int64_t i = base + tid;
    earlier |= (i < pos && tokens[i] == v);
  }
  if (__syncthreads_or(earlier)) return;             // not the first position of this token
  for (int64_t c0 = 0; c0 < E; c0 += 2048) {         // columns tid, tid + 256, ...: eight per thread and pass
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t c = c0 + tid + 256 * j;
      acc[j] = c < E ? static_cast<float>(dy[pos * E + c]) : 0.f;
    }
    for (int64_t base = pos + 1; base < n; base += 256) {
      const int64_t i = base + tid;
      const bool hit = i < n && tokens[i] == v;
      const unsigned long long mask = __ballot(hit);
      __syncthreads();                               // the list of the block before has been consumed
      if (lane == 0) wave_count[wave] = __popcll(mask);
      __syncthreads();
      int off = 0, total = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        off += w < wave ? wave_count[w] : 0;
        total += wave_count[w];
      }
      if (hit) list[off + __popcll(mask & ((1ull << lane) - 1ull))] = tid;
      __syncthreads();
      for (int k = 0; k < total; ++k) {
        const T* row = dy + (base + list[k]) * E + c0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int64_t c = tid + 256 * j;
          if (c0 + c < E) acc[j] += static_cast<float>(row[c]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t c = c0 + tid + 256 * j;
      if (c < E) grad[v * E + c] += acc[j];
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
  CAIMAN_CHECK(n <= 0x7fffffff, "embedding_grad: too many tokens for one launch");
  const dim3 grid((unsigned)n);
  if (dtype == CAIMAN_F32) hipLaunchKernelGGL((embedding_grad_kernel<float>), grid, dim3(256), 0, s, tokens, n, (const float*)dy, V, E, grad);
  else if (dtype == CAIMAN_BF16) hipLaunchKernelGGL((embedding_grad_kernel<bf16_t>), grid, dim3(256), 0, s, tokens, n, (const bf16_t*)dy, V, E, grad);
  else hipLaunchKernelGGL((embedding_grad_kernel<f16_t>), grid, dim3(256), 0, s, tokens, n, (const f16_t*)dy, V, E, grad);
  return check_launch("embedding gradient");
}
