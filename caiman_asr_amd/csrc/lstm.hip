// LSTM recurrent passes for gfx950: gates[t] += y[t]·Rᵀ fused with the cell update
// (forward) and dh[t] = delta[t] + dG[t+1]·R fused with the gate gradients (backward).
//
// Arithmetic contract: training/lib/csrc/lstm.cu
//   :22-76   soft / hard activations and their derivatives on ACTIVATED values
//   :85-135  forward pointwise (gate order i,f,g,o; gates overwritten with activations)
//   :137-212 backward pointwise
//   :214-272 forward time loop (cuBLAS GEMM + pointwise launch per step)
//   :274-346 backward time loop
// The reference issues 2 launches per step and round-trips the pre-activations through
// HBM in the gate dtype.  Here ONE launch per step does the 4-gate recurrent GEMM on
// MFMA (bf16/f16 -> fp32 accumulate), reduces the split-K partials through LDS and
// applies the cell update in fp32 before a single rounding to the storage dtype.
//   fwd tile : 32 batch rows x (4 gates x 8 hidden units), v_mfma_f32_32x32x16, K = H
//              split over the 4 waves of the workgroup; operands are read straight
//              from L2 into registers (every element is used once per workgroup, so
//              LDS staging would be pure overhead — guide §5 "GEMV / M<=16" row).
//   bwd tile : 32 batch rows x 16 hidden units, v_mfma_f32_16x16x32, K = 4H split over
//              8 waves, B operand from a once-per-call transposed copy of R.
// A generic scalar kernel covers f32 / f64 and sizes the MFMA tiles do not divide.
#include "common.h"

namespace caiman {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <typename T>
struct frag8 {
  using type = __attribute__((ext_vector_type(8))) T;
};

__device__ __forceinline__ f32x16 mfma32(frag8<bf16_t>::type a, frag8<bf16_t>::type b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(frag8<f16_t>::type a, frag8<f16_t>::type b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(frag8<bf16_t>::type a, frag8<bf16_t>::type b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(frag8<f16_t>::type a, frag8<f16_t>::type b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// ---- activations (lstm.cu:22-76) -------------------------------------------
template <typename A, bool HARD>
struct Act {
  __device__ __forceinline__ static A clampv(A z, A lo, A hi) { return fmax(lo, fmin(z, hi)); }
  __device__ __forceinline__ static A sigm(A z) {
    if constexpr (HARD) return clampv(A(0.5) + z / A(8), A(0), A(1));
    else return A(1) / (A(1) + exp(-z));
  }
  __device__ __forceinline__ static A tanhv(A z) {
    if constexpr (HARD) return clampv(z, A(-1), A(1));
    else return tanh(z);
  }
  __device__ __forceinline__ static A sigm_prime(A a) {
    if constexpr (HARD) return (a == A(0) || a == A(1)) ? A(0) : A(0.125);
    else return (A(1) - a) * a;
  }
  __device__ __forceinline__ static A tanh_prime(A a) {
    if constexpr (HARD) return (a == A(-1) || a == A(1)) ? A(0) : A(1);
    else return A(1) - a * a;
  }
};

// ===========================================================================
// Generic scalar kernels (any dtype / size): one thread per (b, n).
// ===========================================================================
template <typename T, bool HARD>
__global__ void lstm_fwd_step_generic(const T* __restrict__ R, T* __restrict__ g,
                                      const T* __restrict__ c0, T* __restrict__ c1,
                                      const T* __restrict__ y0, T* __restrict__ y1, int B, int H) {
  using A = acc_t<T>;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (n >= H) return;
  A pre[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const T* rr = R + (int64_t)(q * H + n) * H;
    const T* yy = y0 + (int64_t)b * H;
    A s = 0;
    for (int k = 0; k < H; ++k) s += static_cast<A>(yy[k]) * static_cast<A>(rr[k]);
    pre[q] = static_cast<A>(g[(int64_t)b * 4 * H + q * H + n]) + s;
  }
  const A i = Act<A, HARD>::sigm(pre[0]), f = Act<A, HARD>::sigm(pre[1]);
  const A gg = Act<A, HARD>::tanhv(pre[2]), o = Act<A, HARD>::sigm(pre[3]);
  const A c = i * gg + f * static_cast<A>(c0[(int64_t)b * H + n]);
  g[(int64_t)b * 4 * H + 0 * H + n] = static_cast<T>(i);
  g[(int64_t)b * 4 * H + 1 * H + n] = static_cast<T>(f);
  g[(int64_t)b * 4 * H + 2 * H + n] = static_cast<T>(gg);
  g[(int64_t)b * 4 * H + 3 * H + n] = static_cast<T>(o);
  c1[(int64_t)b * H + n] = static_cast<T>(c);
  y1[(int64_t)b * H + n] = static_cast<T>(o * Act<A, HARD>::tanhv(c));
}

template <typename T, bool HARD>
__global__ void lstm_bwd_step_generic(const T* __restrict__ R, const T* __restrict__ g,
                                      const T* __restrict__ c_prev, const T* __restrict__ c_cur,
                                      const T* __restrict__ delta, int64_t d_sb,
                                      const T* __restrict__ dG_next, T* __restrict__ dG,
                                      acc_t<T>* __restrict__ dC, int B, int H) {
  using A = acc_t<T>;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (n >= H) return;
  A dy = static_cast<A>(delta[(int64_t)b * d_sb + n]);
  if (dG_next) {
    const T* dg = dG_next + (int64_t)b * 4 * H;
    A s = 0;
    for (int r = 0; r < 4 * H; ++r) s += static_cast<A>(dg[r]) * static_cast<A>(R[(int64_t)r * H + n]);
    dy += s;
  }
  const int64_t gb = (int64_t)b * 4 * H;
  const A i = static_cast<A>(g[gb + n]), f = static_cast<A>(g[gb + H + n]);
  const A gg = static_cast<A>(g[gb + 2 * H + n]), o = static_cast<A>(g[gb + 3 * H + n]);
  const A ct = Act<A, HARD>::tanhv(static_cast<A>(c_cur[(int64_t)b * H + n]));
  const A dO = dy * ct * Act<A, HARD>::sigm_prime(o);
  const A dc = dy * o * Act<A, HARD>::tanh_prime(ct) + dC[(int64_t)b * H + n];
  dG[gb + n] = static_cast<T>(dc * gg * Act<A, HARD>::sigm_prime(i));
  dG[gb + H + n] = static_cast<T>(dc * static_cast<A>(c_prev[(int64_t)b * H + n]) * Act<A, HARD>::sigm_prime(f));
  dG[gb + 2 * H + n] = static_cast<T>(dc * i * Act<A, HARD>::tanh_prime(gg));
  dG[gb + 3 * H + n] = static_cast<T>(dO);
  dC[(int64_t)b * H + n] = dc * f;
}

// ===========================================================================
// MFMA forward step.  grid = (H/8, ceil(B/32)), 256 threads.
// ===========================================================================
template <typename T, bool HARD>
__global__ __launch_bounds__(256) void lstm_fwd_step_mfma(const T* __restrict__ R, T* __restrict__ g,
                                                          const T* __restrict__ c0, T* __restrict__ c1,
                                                          const T* __restrict__ y0, T* __restrict__ y1,
                                                          int B, int H) {
  using frag = typename frag8<T>::type;
  __shared__ float tile[4][32][33];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int j0 = blockIdx.x * 8, m0 = blockIdx.y * 32;
  const int Kw = H >> 2;  // K range of this wave
  const int kbase = wave * Kw + 8 * hh;
  // B operand: column n = r of the tile <-> R row (gate = r>>3, unit = j0 + (r&7)), K-contiguous.
  const T* Rrow = R + (int64_t)((r >> 3) * H + j0 + (r & 7)) * H + kbase;
  const int brow = m0 + r;
  const bool bvalid = brow < B;
  const T* Arow = y0 + (int64_t)(bvalid ? brow : 0) * H + kbase;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  const int nk = Kw >> 4;
#pragma unroll 8
  for (int s = 0; s < nk; ++s) {
    frag a = *reinterpret_cast<const frag*>(Arow + 16 * s);
    frag b = *reinterpret_cast<const frag*>(Rrow + 16 * s);
    if (!bvalid) {
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] = static_cast<T>(0.f);
    }
    acc = mfma32(a, b, acc);
  }
  // C layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int q = 0; q < 16; ++q) tile[wave][(q & 3) + 8 * (q >> 2) + 4 * hh][r] = acc[q];
  __syncthreads();
  const int eb = threadIdx.x >> 3, eu = threadIdx.x & 7;
  const int b = m0 + eb, n = j0 + eu;
  if (b >= B) return;
  float pre[4];
  const int64_t gb = (int64_t)b * 4 * H + n;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = q * 8 + eu;
    pre[q] = static_cast<float>(g[gb + (int64_t)q * H]) + tile[0][eb][col] + tile[1][eb][col] +
             tile[2][eb][col] + tile[3][eb][col];
  }
  const float i = Act<float, HARD>::sigm(pre[0]), f = Act<float, HARD>::sigm(pre[1]);
  const float gg = Act<float, HARD>::tanhv(pre[2]), o = Act<float, HARD>::sigm(pre[3]);
  const float c = i * gg + f * static_cast<float>(c0[(int64_t)b * H + n]);
  g[gb] = static_cast<T>(i);
  g[gb + H] = static_cast<T>(f);
  g[gb + 2 * (int64_t)H] = static_cast<T>(gg);
  g[gb + 3 * (int64_t)H] = static_cast<T>(o);
  c1[(int64_t)b * H + n] = static_cast<T>(c);
  y1[(int64_t)b * H + n] = static_cast<T>(o * Act<float, HARD>::tanhv(c));
}

// ===========================================================================
// MFMA backward step.  grid = (H/16, ceil(B/32)), 512 threads (8 waves, split-K over 4H).
// Rt = Rᵀ, [H, 4H] row-major.
// ===========================================================================
template <typename T, bool HARD>
__global__ __launch_bounds__(512) void lstm_bwd_step_mfma(const T* __restrict__ Rt, const T* __restrict__ g,
                                                          const T* __restrict__ c_prev,
                                                          const T* __restrict__ c_cur,
                                                          const T* __restrict__ delta, int64_t d_sb,
                                                          const T* __restrict__ dG_next, T* __restrict__ dG,
                                                          float* __restrict__ dC, int B, int H) {
  using frag = typename frag8<T>::type;
  __shared__ float tile[8][2][16][17];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 32;
  if (dG_next) {
    const int r = lane & 15, kq = lane >> 4;
    const int K = 4 * H, Kw = K >> 3;
    const int kbase = wave * Kw + 8 * kq;
    const T* Brow = Rt + (int64_t)(n0 + r) * K + kbase;
    const int b0 = m0 + r, b1 = m0 + 16 + r;
    const bool v0 = b0 < B, v1 = b1 < B;
    const T* A0 = dG_next + (int64_t)(v0 ? b0 : 0) * K + kbase;
    const T* A1 = dG_next + (int64_t)(v1 ? b1 : 0) * K + kbase;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int nk = Kw >> 5;
#pragma unroll 4
    for (int s = 0; s < nk; ++s) {
      frag bb = *reinterpret_cast<const frag*>(Brow + 32 * s);
      frag a0 = *reinterpret_cast<const frag*>(A0 + 32 * s);
      frag a1 = *reinterpret_cast<const frag*>(A1 + 32 * s);
      if (!v0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a0[q] = static_cast<T>(0.f);
      }
      if (!v1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a1[q] = static_cast<T>(0.f);
      }
      acc0 = mfma16(a0, bb, acc0);
      acc1 = mfma16(a1, bb, acc1);
    }
    // C layout 16x16: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      tile[wave][0][kq * 4 + q][r] = acc0[q];
      tile[wave][1][kq * 4 + q][r] = acc1[q];
    }
  }
  __syncthreads();
  const int eb = threadIdx.x >> 4, eu = threadIdx.x & 15;  // 32 rows x 16 units
  const int b = m0 + eb, n = n0 + eu;
  if (b >= B) return;
  float dy = static_cast<float>(delta[(int64_t)b * d_sb + n]);
  if (dG_next) {
    const int mt = eb >> 4, rr = eb & 15;
#pragma unroll
    for (int w = 0; w < 8; ++w) dy += tile[w][mt][rr][eu];
  }
  const int64_t gb = (int64_t)b * 4 * H + n;
  const float i = static_cast<float>(g[gb]), f = static_cast<float>(g[gb + H]);
  const float gg = static_cast<float>(g[gb + 2 * (int64_t)H]), o = static_cast<float>(g[gb + 3 * (int64_t)H]);
  const float ct = Act<float, HARD>::tanhv(static_cast<float>(c_cur[(int64_t)b * H + n]));
  const float dO = dy * ct * Act<float, HARD>::sigm_prime(o);
  const float dc = dy * o * Act<float, HARD>::tanh_prime(ct) + dC[(int64_t)b * H + n];
  dG[gb] = static_cast<T>(dc * gg * Act<float, HARD>::sigm_prime(i));
  dG[gb + H] = static_cast<T>(dc * static_cast<float>(c_prev[(int64_t)b * H + n]) * Act<float, HARD>::sigm_prime(f));
  dG[gb + 2 * (int64_t)H] = static_cast<T>(dc * i * Act<float, HARD>::tanh_prime(gg));
  dG[gb + 3 * (int64_t)H] = static_cast<T>(dO);
  dC[(int64_t)b * H + n] = dc * f;
}

// [rows, cols] -> [cols, rows], 32x32 LDS tiles.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                        int rows, int cols) {
  __shared__ T t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int k = ty; k < 32; k += 8)
    if (r0 + k < rows && c0 + tx < cols) t[k][tx] = in[(int64_t)(r0 + k) * cols + c0 + tx];
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (c0 + k < cols && r0 + tx < rows) out[(int64_t)(c0 + k) * rows + r0 + tx] = t[tx][k];
}

template <typename T>
constexpr bool kHasMfma = std::is_same<T, bf16_t>::value || std::is_same<T, f16_t>::value;

template <typename T, bool HARD>
int run_fwd(const T* R, T* gates, T* c, T* y, int64_t Tn, int64_t B, int64_t H, hipStream_t s) {
  const int64_t go = B * 4 * H, so = B * H;
  bool mfma = false;
  if constexpr (kHasMfma<T>) mfma = (H % 64 == 0);
  for (int64_t t = 0; t < Tn; ++t) {
    T* g = gates + go * t;
    if constexpr (kHasMfma<T>) {
      if (mfma) {
        hipLaunchKernelGGL((lstm_fwd_step_mfma<T, HARD>), dim3((unsigned)(H / 8), (unsigned)((B + 31) / 32)),
                           dim3(256), 0, s, R, g, c + so * t, c + so * (t + 1), y + so * t,
                           y + so * (t + 1), (int)B, (int)H);
        continue;
      }
    }
    hipLaunchKernelGGL((lstm_fwd_step_generic<T, HARD>), dim3((unsigned)((H + 63) / 64), (unsigned)B), dim3(64),
                       0, s, R, g, c + so * t, c + so * (t + 1), y + so * t, y + so * (t + 1), (int)B, (int)H);
  }
  return check_launch("caiman_lstm_fused_fwd");
}

template <typename T, bool HARD>
int run_bwd(const T* R, const T* gates, const T* c, const T* delta, int64_t d_st, int64_t d_sb, T* dG,
            acc_t<T>* dC, T* Rt, int64_t Tn, int64_t B, int64_t H, hipStream_t s) {
  const int64_t go = B * 4 * H, so = B * H;
  if (hipMemsetAsync(dC, 0, sizeof(acc_t<T>) * (size_t)so, s) != hipSuccess) return check_launch("lstm_bwd memset");
  bool mfma = false;
  if constexpr (kHasMfma<T>) mfma = (H % 64 == 0) && Rt != nullptr;
  if (mfma && Tn > 1) {
    hipLaunchKernelGGL((transpose_kernel<T>), dim3((unsigned)((H + 31) / 32), (unsigned)((4 * H + 31) / 32)),
                       dim3(256), 0, s, R, Rt, (int)(4 * H), (int)H);
  }
  for (int64_t t = Tn - 1; t >= 0; --t) {
    const T* dgn = (t < Tn - 1) ? dG + go * (t + 1) : nullptr;
    if constexpr (kHasMfma<T>) {
      if (mfma) {
        hipLaunchKernelGGL((lstm_bwd_step_mfma<T, HARD>), dim3((unsigned)(H / 16), (unsigned)((B + 31) / 32)),
                           dim3(512), 0, s, Rt, gates + go * t, c + so * t, c + so * (t + 1),
                           delta + d_st * t, d_sb, dgn, dG + go * t, dC, (int)B, (int)H);
        continue;
      }
    }
    hipLaunchKernelGGL((lstm_bwd_step_generic<T, HARD>), dim3((unsigned)((H + 63) / 64), (unsigned)B), dim3(64),
                       0, s, R, gates + go * t, c + so * t, c + so * (t + 1), delta + d_st * t, d_sb, dgn,
                       dG + go * t, dC, (int)B, (int)H);
  }
  return check_launch("caiman_lstm_fused_bwd");
}

}  // namespace
}  // namespace caiman

extern "C" int caiman_lstm_fused_fwd(const void* R, void* gates, void* c, void* y, int64_t T, int64_t B,
                                     int64_t H, int dtype, int hard, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(T >= 0 && B >= 1 && H >= 1, "lstm_fused_fwd: bad extents T=%lld B=%lld H=%lld", (long long)T,
               (long long)B, (long long)H);
  CAIMAN_CHECK(B <= 65535, "lstm_fused_fwd: batch too large for one launch");
  if (T == 0) return CAIMAN_OK;
  CAIMAN_CHECK(R && gates && c && y, "lstm_fused_fwd: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return CAIMAN_DISPATCH(dtype, "lstm_fused_fwd", [&]() -> int {
    auto Rp = static_cast<const scalar_t*>(R);
    auto gp = static_cast<scalar_t*>(gates);
    auto cp = static_cast<scalar_t*>(c);
    auto yp = static_cast<scalar_t*>(y);
    return hard ? run_fwd<scalar_t, true>(Rp, gp, cp, yp, T, B, H, s)
                : run_fwd<scalar_t, false>(Rp, gp, cp, yp, T, B, H, s);
  });
}

extern "C" int caiman_lstm_fused_bwd(const void* R, const void* gates, const void* c, const void* delta,
                                     int64_t delta_stride_t, int64_t delta_stride_b, void* dG, void* dC,
                                     void* Rt, int64_t T, int64_t B, int64_t H, int dtype, int hard,
                                     caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(T >= 0 && B >= 1 && H >= 1, "lstm_fused_bwd: bad extents T=%lld B=%lld H=%lld", (long long)T,
               (long long)B, (long long)H);
  CAIMAN_CHECK(B <= 65535, "lstm_fused_bwd: batch too large for one launch");
  if (T == 0) return CAIMAN_OK;
  CAIMAN_CHECK(R && gates && c && delta && dG && dC, "lstm_fused_bwd: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return CAIMAN_DISPATCH(dtype, "lstm_fused_bwd", [&]() -> int {
    using A = acc_t<scalar_t>;
    auto Rp = static_cast<const scalar_t*>(R);
    auto gp = static_cast<const scalar_t*>(gates);
    auto cp = static_cast<const scalar_t*>(c);
    auto dp = static_cast<const scalar_t*>(delta);
    auto dGp = static_cast<scalar_t*>(dG);
    auto dCp = static_cast<A*>(dC);
    auto Rtp = static_cast<scalar_t*>(Rt);
    return hard ? run_bwd<scalar_t, true>(Rp, gp, cp, dp, delta_stride_t, delta_stride_b, dGp, dCp, Rtp, T, B, H, s)
                : run_bwd<scalar_t, false>(Rp, gp, cp, dp, delta_stride_t, delta_stride_b, dGp, dCp, Rtp, T, B, H, s);
  });
}
