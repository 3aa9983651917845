// Column sums of a tall row-major matrix: out[b][c] = sum_r x[b][r][c] -- the bias gradients of the LSTM layers
// (training/lib/src/rnnt_ext/custom_lstm/lstm.py:57, `dB = dG.sum([0, 1])`) and of the joint's linear layers.
// The matrices are [T*B, 4H] with tens of thousands of rows and a few thousand columns: pure streaming, HBM-bound.
// Each thread owns 8 adjacent columns (one 16-byte load per row), a 256-thread workgroup a 2048-column stripe of
// a row range; partial sums go to a [splits, cols] fp32 scratch and a second kernel adds the splits in a fixed
// order, so the result does not depend on scheduling (no atomics).
#include "common.h"

namespace caiman {
namespace {

constexpr int kColsPerThread = 8;
constexpr int kStripe = 256 * kColsPerThread;

template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int64_t rows, int64_t cols,
                                                            int64_t batch_stride, int splits, float* __restrict__ partial) {
  using vec = __attribute__((ext_vector_type(kColsPerThread))) T;
  const int64_t c0 = (int64_t)blockIdx.x * kStripe + (int64_t)threadIdx.x * kColsPerThread;
  const int split = blockIdx.y, b = blockIdx.z;
  if (c0 >= cols) return;
  const int64_t per = (rows + splits - 1) / splits;
  const int64_t r0 = split * per, r1 = r0 + per < rows ? r0 + per : rows;
  const T* p = x + b * batch_stride + c0;
  float acc[kColsPerThread];
#pragma unroll
  for (int q = 0; q < kColsPerThread; ++q) acc[q] = 0.f;
  int64_t r = r0;
  for (; r + 4 <= r1; r += 4) {   // four rows in flight
    vec v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const vec*>(p + (r + i) * cols);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < kColsPerThread; ++q) acc[q] += static_cast<float>(v[i][q]);
  }
  for (; r < r1; ++r) {
    const vec v = *reinterpret_cast<const vec*>(p + r * cols);
#pragma unroll
    for (int q = 0; q < kColsPerThread; ++q) acc[q] += static_cast<float>(v[q]);
  }
  float* o = partial + ((int64_t)b * splits + split) * cols + c0;
#pragma unroll
  for (int q = 0; q < kColsPerThread; ++q) o[q] = acc[q];
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int64_t cols, int splits,
                                                          T* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (c >= cols) return;
  const float* p = partial + (int64_t)b * splits * cols + c;
  float s = 0.f;
  for (int i = 0; i < splits; ++i) s += p[(int64_t)i * cols];
  out[(int64_t)b * cols + c] = static_cast<T>(s);
}

}  // namespace
}  // namespace caiman

extern "C" int64_t caiman_colsum_splits(int64_t batch, int64_t rows, int64_t cols) {
  if (batch < 1 || rows < 1 || cols < 1) return 1;
  const int64_t stripes = (cols + caiman::kStripe - 1) / caiman::kStripe;
  int64_t splits = (4096 + batch * stripes - 1) / (batch * stripes);   // ~16 workgroups per CU over the whole grid
  const int64_t max_by_rows = (rows + 31) / 32;                         // at least 32 rows per workgroup
  if (splits > max_by_rows) splits = max_by_rows;
  if (splits > 1024) splits = 1024;
  return splits < 1 ? 1 : splits;
}

extern "C" int caiman_colsum(const void* x, int64_t batch, int64_t rows, int64_t cols, int64_t batch_stride, void* out,
                             float* partial, int64_t splits, int dtype, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(batch >= 1 && rows >= 0 && cols >= 1 && batch <= 65535, "colsum: bad extents");
  CAIMAN_CHECK(dtype == CAIMAN_F16 || dtype == CAIMAN_BF16, "colsum: f16 / bf16 only");
  CAIMAN_CHECK(cols % kColsPerThread == 0, "colsum: the column count must be a multiple of %d (got %lld)", kColsPerThread,
               (long long)cols);
  CAIMAN_CHECK(x && out && partial, "colsum: null pointer");
  CAIMAN_CHECK((reinterpret_cast<uintptr_t>(x) & 15u) == 0 && batch_stride % kColsPerThread == 0, "colsum: x must be 16-byte aligned");
  CAIMAN_CHECK(splits >= 1 && splits <= 65535, "colsum: 1..65535 splits");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 g1((unsigned)((cols + kStripe - 1) / kStripe), (unsigned)splits, (unsigned)batch);
  const dim3 g2((unsigned)((cols + 255) / 256), (unsigned)batch);
  if (dtype == CAIMAN_BF16) {
    hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), g1, dim3(256), 0, s, (const bf16_t*)x, rows, cols, batch_stride, (int)splits, partial);
    hipLaunchKernelGGL((colsum_final_kernel<bf16_t>), g2, dim3(256), 0, s, partial, cols, (int)splits, (bf16_t*)out);
  } else {
    hipLaunchKernelGGL((colsum_partial_kernel<f16_t>), g1, dim3(256), 0, s, (const f16_t*)x, rows, cols, batch_stride, (int)splits, partial);
    hipLaunchKernelGGL((colsum_final_kernel<f16_t>), g2, dim3(256), 0, s, partial, cols, (int)splits, (f16_t*)out);
  }
  return check_launch("caiman_colsum");
}
