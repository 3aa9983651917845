// Row-wise logsumexp with no [rows, n] temporary — gfx950 / wave64.
//
// Behavioural contract follows the reference op
//   training/lib/csrc/logsumexp.cu:65-105 (kernel), :189-243 (host wrapper):
//   max pass (NaN-propagating) -> if the max is not finite emit it -> sum of
//   exp(x - max) in the accumulate type -> max + log(sum).
// The implementation is new: one 256-thread workgroup (4 waves) per row, 16-byte
// coalesced loads, the row is kept in registers between the two passes when it
// fits so HBM is read once, wave64 shuffles + one LDS hop for the block folds
// (the reference's block_map_fold hard-codes 32-lane warps,
// training/lib/csrc/myrtle/block_map_fold.cuh:65,125).
#include "common.h"

namespace caiman {
namespace {

template <typename A>
__device__ __forceinline__ A nan_max(A x, A y) {
  // training/lib/csrc/logsumexp.cu:30-34 — NaN wins from either side.
  return (x != x) ? x : (x > y ? x : y);
}

template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) vec_of {
  T v[VEC];
};

template <typename A>
__device__ __forceinline__ A dev_exp(A x);
template <>
__device__ __forceinline__ float dev_exp<float>(float x) {
  return __expf(x);
}
template <>
__device__ __forceinline__ double dev_exp<double>(double x) {
  return exp(x);
}

// NW waves per row, VEC elements per 16-byte load, CACHE chunks per thread held
// in registers across the two passes (0 = always re-read, used for odd layouts).
template <typename T, typename O, int NW, int VEC, int CACHE>
__global__ __launch_bounds__(NW* kWave) void lse_rows_kernel(const T* __restrict__ in,
                                                             int64_t n, int64_t stride,
                                                             O* __restrict__ out) {
  using A = acc_t<T>;
  using V = vec_of<T, VEC>;
  constexpr int NT = NW * kWave;
  __shared__ A smem[NW];

  const T* row = in + (int64_t)blockIdx.x * stride;
  const int tid = threadIdx.x;
  const int64_t nfull = n / VEC;  // whole chunks
  const A lowest = -INFINITY;

  A vals[CACHE > 0 ? CACHE * VEC : 1];
  A m = lowest;

  // ---- pass 1: max -----------------------------------------------------------
  if constexpr (CACHE > 0) {
    // all CACHE loads first, unconditionally per lane (a lane past the row's end loads the row's last chunk again and
    // discards it): under the per-lane condition `c < nfull` every load was a branch with a `vmcnt(0)` of its own behind
    // it -- one 16-byte load in flight per wave (seen in the ISA)
    V raw[CACHE] = {};
    if (nfull > 0) {
#pragma unroll
      for (int k = 0; k < CACHE; ++k) {
        const int64_t c = tid + (int64_t)k * NT;
        raw[k] = *reinterpret_cast<const V*>(row + (c < nfull ? c : nfull - 1) * VEC);
      }
    }
#pragma unroll
    for (int k = 0; k < CACHE; ++k) {
      const bool own = tid + (int64_t)k * NT < nfull;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        vals[k * VEC + j] = own ? static_cast<A>(raw[k].v[j]) : lowest;
        m = nan_max(m, vals[k * VEC + j]);
      }
    }
  }
  for (int64_t c = tid + (int64_t)CACHE * NT; c < nfull; c += NT) {
    V v = *reinterpret_cast<const V*>(row + c * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) m = nan_max(m, static_cast<A>(v.v[j]));
  }
  for (int64_t i = nfull * VEC + tid; i < n; i += NT) m = nan_max(m, static_cast<A>(row[i]));

  m = block_reduce<NW>(m, [](A a, A b) { return nan_max(a, b); }, smem);

  if (!isfinite(m)) {  // logsumexp.cu:91-96
    if (tid == 0) out[blockIdx.x] = static_cast<O>(m);
    return;
  }

  // ---- pass 2: sum exp(x - max) ------------------------------------------------
  A s = 0;
  if constexpr (CACHE > 0) {
#pragma unroll
    for (int k = 0; k < CACHE * VEC; ++k) s += dev_exp<A>(vals[k] - m);  // exp(-inf)=0 pads
  }
  for (int64_t c = tid + (int64_t)CACHE * NT; c < nfull; c += NT) {
    V v = *reinterpret_cast<const V*>(row + c * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) s += dev_exp<A>(static_cast<A>(v.v[j]) - m);
  }
  for (int64_t i = nfull * VEC + tid; i < n; i += NT) s += dev_exp<A>(static_cast<A>(row[i]) - m);

  s = block_reduce<NW>(s, [](A a, A b) { return a + b; }, smem);
  if (tid == 0) out[blockIdx.x] = static_cast<O>(m + log(s));
}

template <typename T, typename O>
int launch_lse(const T* in, int64_t rows, int64_t n, int64_t stride, O* out,
               hipStream_t stream) {
  constexpr int VEC = 16 / sizeof(T);
  const bool aligned = (reinterpret_cast<uintptr_t>(in) % 16 == 0) &&
                       ((stride * (int64_t)sizeof(T)) % 16 == 0);
  const dim3 grid((unsigned)rows);
  if (!aligned) {
    hipLaunchKernelGGL((lse_rows_kernel<T, O, 4, 1, 0>), grid, dim3(256), 0, stream, in, n,
                       stride, out);
  } else if (n <= 64 * VEC * 4) {
    // short rows: a single wave, no LDS hop
    hipLaunchKernelGGL((lse_rows_kernel<T, O, 1, VEC, 4>), grid, dim3(64), 0, stream, in, n,
                       stride, out);
  } else if constexpr (sizeof(T) == 8) {
    hipLaunchKernelGGL((lse_rows_kernel<T, O, 4, VEC, 0>), grid, dim3(256), 0, stream, in, n,
                       stride, out);
  } else if (n <= 256 * VEC * 5) {
    // V = 8704 (base) lands here for bf16/f16: 1088 chunks -> 5 per thread.
    hipLaunchKernelGGL((lse_rows_kernel<T, O, 4, VEC, 5>), grid, dim3(256), 0, stream, in, n,
                       stride, out);
  } else {
    // V = 17408 (large) bf16: 2176 chunks -> 9 per thread; f32 base: 2176 chunks.
    hipLaunchKernelGGL((lse_rows_kernel<T, O, 4, VEC, 9>), grid, dim3(256), 0, stream, in, n,
                       stride, out);
  }
  return check_launch("caiman_logsumexp");
}

}  // namespace
}  // namespace caiman

extern "C" int caiman_logsumexp(const void* in, int64_t rows, int64_t n, int64_t row_stride,
                                int in_dtype, void* out, int out_dtype, uint32_t /*max_threads*/,
                                caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(rows >= 0 && n >= 0, "logsumexp: negative extent");
  CAIMAN_CHECK(row_stride >= n, "logsumexp: input tensor must not alias itself (stride %lld < n %lld)",
               (long long)row_stride, (long long)n);  // logsumexp.cu:195
  CAIMAN_CHECK(rows < (int64_t)1 << 31, "logsumexp: too many rows for one launch");
  if (rows == 0) return CAIMAN_OK;
  CAIMAN_CHECK(in != nullptr && out != nullptr, "logsumexp: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return CAIMAN_DISPATCH(in_dtype, "logsumexp", [&]() -> int {
    using A = acc_t<scalar_t>;
    const int acc_tag = std::is_same<A, double>::value ? CAIMAN_F64 : CAIMAN_F32;
    if (out_dtype == acc_tag) {
      return launch_lse<scalar_t, A>(static_cast<const scalar_t*>(in), rows, n, row_stride,
                                     static_cast<A*>(out), s);
    }
    CAIMAN_CHECK(out_dtype == in_dtype, "logsumexp: out dtype must be the input or accumulate type");
    return launch_lse<scalar_t, scalar_t>(static_cast<const scalar_t*>(in), rows, n, row_stride,
                                          static_cast<scalar_t*>(out), s);
  });
}
