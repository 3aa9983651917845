// Transducer joint: h[b,t,u,:] = dropout(relu(f[b,t,:] + g[b,u,:])), written packed or padded.
//
// Replaces the third-party op the reference calls on this path,
// apex.contrib.transducer.TransducerJoint (not vendored under /root/reference; call sites
// training/caiman_asr_train/rnnt/model.py:228-238,425-434).  Semantics pinned by the
// reference's tests: equal to the broadcast add on the valid region with -1 in the padded
// region (training/tests/rnnt/test_model.py:34-64), dropout inactive in eval (:67-104),
// packed row order = batch_offset[b-1] + t*(U_b+1) + u, as addressed by the loss kernel
// (training/lib/csrc/transducer_loss.cu:107-116).  The dropout mask stream is this library's
// own counter-based generator (parity with apex's Philox stream is not defined).
//
// HBM-bound elementwise / reduction kernels: one wave64 per output row, 16-byte accesses.
#include "common.h"

namespace caiman {
namespace {

template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) vecj {
  T v[VEC];
};


struct JointShape {
  const int32_t* f_len;
  const int32_t* g_len;
  const int64_t* batch_offset;
  int64_t B, T, U, H, total_rows;
  int packed;
};

// One workgroup per (t, b): its four waves take the rows u = wave, wave + 4, ... of that frame.  (The first version gave
// every wave one packed row and found (b, t, u) by a binary search over batch_offset: five dependent loads in front of
// 1.5 KB of work per wave.)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void joint_fwd_kernel(const T* __restrict__ f, const T* __restrict__ g,
                                                        JointShape s, int relu, float drop_p, uint64_t seed,
                                                        T* __restrict__ out) {
  using V = vecj<T, VEC>;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwave = blockDim.x / kWave;
  const int b = blockIdx.y;
  const int64_t t = blockIdx.x;
  const int64_t fl = s.f_len[b], gl = s.g_len[b];
  if (s.packed && t >= fl) return;                         // the packed layout has no rows for this frame
  const int64_t nu = s.packed ? gl : s.U;
  const int64_t base = s.packed ? (b == 0 ? 0 : s.batch_offset[b - 1]) + t * gl : ((int64_t)b * s.T + t) * s.U;
  const T* fr = f + ((int64_t)b * s.T + t) * s.H;
  const int64_t nfull = s.H / VEC;
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  // `ks`: 1 without dropout, else 0 (dropped) or 1 / (1 - p) (common.h drop_scale: one hash per four consecutive elements)
  auto one = [&](float a, float c, float ks) -> float {
    float v = a + c;
    // NaN-propagating like torch.relu (the reference's CPU joint, model.py:441-447): a non-finite encoder / prediction
    // output must reach the loss as NaN so that the batch is dropped (core.py:20-42), not be clamped to 0 here
    if (relu) v = (v > 0.f || v != v) ? v : 0.f;
    return drop_p > 0.f ? (ks == 0.f ? 0.f : v * ks) : v;
  };
  for (int64_t u = wave; u < nu; u += nwave) {
    const int64_t row = base + u;
    T* o = out + row * s.H;
    if (!(t < fl && u < gl)) {  // padded layout, don't-care cell: -1 (training/tests/rnnt/test_model.py:57-60)
      V m;
#pragma unroll
      for (int j = 0; j < VEC; ++j) m.v[j] = static_cast<T>(-1.f);
      for (int64_t c = lane; c < nfull; c += kWave) *reinterpret_cast<V*>(o + c * VEC) = m;
      for (int64_t h = nfull * VEC + lane; h < s.H; h += kWave) o[h] = static_cast<T>(-1.f);
      continue;
    }
    const T* gr = g + ((int64_t)b * s.U + u) * s.H;
    for (int64_t c = lane; c < nfull; c += kWave) {
      const V a = *reinterpret_cast<const V*>(fr + c * VEC);
      const V d = *reinterpret_cast<const V*>(gr + c * VEC);
      V r;
      if constexpr (VEC % 4 == 0) {
#pragma unroll
        for (int j4 = 0; j4 < VEC; j4 += 4) {
          float ks[4] = {1.f, 1.f, 1.f, 1.f};
          if (drop_p > 0.f) drop_scale4(seed, (uint64_t)(row * s.H + c * VEC + j4), drop_p, keep_scale, ks);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            r.v[j4 + j] = static_cast<T>(one(static_cast<float>(a.v[j4 + j]), static_cast<float>(d.v[j4 + j]), ks[j]));
        }
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float ks = drop_p > 0.f ? drop_scale(seed, (uint64_t)(row * s.H + c * VEC + j), drop_p, keep_scale) : 1.f;
          r.v[j] = static_cast<T>(one(static_cast<float>(a.v[j]), static_cast<float>(d.v[j]), ks));
        }
      }
      *reinterpret_cast<V*>(o + c * VEC) = r;
    }
    for (int64_t h = nfull * VEC + lane; h < s.H; h += kWave) {
      const float ks = drop_p > 0.f ? drop_scale(seed, (uint64_t)(row * s.H + h), drop_p, keep_scale) : 1.f;
      o[h] = static_cast<T>(one(static_cast<float>(fr[h]), static_cast<float>(gr[h]), ks));
    }
  }
}

// mode: 0 = plain add, 1 = relu (+dropout): mask = out > 0, 2 = dropout only: mask = out != 0
template <typename T>
__device__ __forceinline__ float masked(float dh, float ho, int mode, float scale) {
  if (mode == 1) return ho > 0.f ? dh * scale : 0.f;
  if (mode == 2) return ho != 0.f ? dh * scale : 0.f;
  return dh;
}

// REDUCE_U = true : df[b,t,:] = sum_u dz ; grid (T, B)
// REDUCE_U = false: dg[b,u,:] = sum_t dz ; grid (U, B)
template <typename T, int VEC, bool REDUCE_U>
__global__ __launch_bounds__(128) void joint_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ ho,
                                                        JointShape s, int mode, float scale,
                                                        T* __restrict__ dout) {
  using V = vecj<T, VEC>;
  const int b = blockIdx.y;
  const int64_t i = blockIdx.x;  // t (REDUCE_U) or u
  const int64_t fl = s.f_len[b], gl = s.g_len[b];
  const int64_t base = s.packed ? (b == 0 ? 0 : s.batch_offset[b - 1]) : (int64_t)b * s.T * s.U;
  const int64_t ustride = s.packed ? gl : s.U;
  const int64_t nout = REDUCE_U ? s.T : s.U;
  T* o = dout + ((int64_t)b * nout + i) * s.H;
  const bool live = REDUCE_U ? (i < fl) : (i < gl);
  const int64_t nred = REDUCE_U ? gl : fl;
  for (int64_t c = threadIdx.x; c * VEC < s.H; c += blockDim.x) {
    const int64_t h0 = c * VEC;
    const bool fullvec = (h0 + VEC <= s.H);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    if (live && fullvec && mode) {
      // four rows' loads in flight, then their terms added in row order (the sum is the one-row-at-a-time loop's, bit for bit)
      const int64_t row0 = REDUCE_U ? base + i * ustride : base + i;
      const int64_t rstep = (REDUCE_U ? 1 : ustride) * s.H;
      const T* pd = dh + row0 * s.H + h0;
      const T* ph = ho + row0 * s.H + h0;
      int64_t k = 0;
      for (; k + 4 <= nred; k += 4) {
        V a[4], m[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a[q] = *reinterpret_cast<const V*>(pd + (k + q) * rstep);
          m[q] = *reinterpret_cast<const V*>(ph + (k + q) * rstep);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < VEC; ++j)
            acc[j] += masked<T>(static_cast<float>(a[q].v[j]), static_cast<float>(m[q].v[j]), mode, scale);
      }
      for (; k < nred; ++k) {
        const V a = *reinterpret_cast<const V*>(pd + k * rstep);
        const V m = *reinterpret_cast<const V*>(ph + k * rstep);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          acc[j] += masked<T>(static_cast<float>(a.v[j]), static_cast<float>(m.v[j]), mode, scale);
      }
    } else if (live) {
      for (int64_t k = 0; k < nred; ++k) {
        const int64_t row = REDUCE_U ? base + i * ustride + k : base + k * ustride + i;
        const T* pd = dh + row * s.H + h0;
        const T* ph = ho + row * s.H + h0;
        if (fullvec) {
          const V a = *reinterpret_cast<const V*>(pd);
          V m;
          if (mode) m = *reinterpret_cast<const V*>(ph);
#pragma unroll
          for (int j = 0; j < VEC; ++j)
            acc[j] += masked<T>(static_cast<float>(a.v[j]), mode ? static_cast<float>(m.v[j]) : 0.f, mode, scale);
        } else {
          for (int j = 0; h0 + j < s.H; ++j)
            acc[j] += masked<T>(static_cast<float>(pd[j]), mode ? static_cast<float>(ph[j]) : 0.f, mode, scale);
        }
      }
    }
    if (fullvec) {
      V r;
#pragma unroll
      for (int j = 0; j < VEC; ++j) r.v[j] = static_cast<T>(acc[j]);
      *reinterpret_cast<V*>(o + h0) = r;
    } else {
      for (int j = 0; h0 + j < s.H; ++j) o[h0 + j] = static_cast<T>(acc[j]);
    }
  }
}

}  // namespace
}  // namespace caiman

extern "C" int caiman_joint_forward(const void* f, const void* g, const int32_t* f_len, const int32_t* g_len,
                                    const int64_t* batch_offset, int64_t B, int64_t T, int64_t U, int64_t H,
                                    int64_t total_rows, int packed, int relu, double dropout_p, uint64_t seed,
                                    int dtype, void* out, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(B >= 1 && T >= 1 && U >= 1 && H >= 1 && total_rows >= 0, "joint_forward: bad extents");
  CAIMAN_CHECK(dropout_p >= 0.0 && dropout_p < 1.0, "joint_forward: dropout_p must be in [0,1)");
  CAIMAN_CHECK(!packed || batch_offset, "joint_forward: packed output needs batch_offset");
  CAIMAN_CHECK(packed || total_rows == B * T * U, "joint_forward: padded output must have B*T*U rows");
  if (total_rows == 0) return CAIMAN_OK;
  CAIMAN_CHECK(f && g && f_len && g_len && out, "joint_forward: null pointer");
  CAIMAN_CHECK(B <= 65535 && T < ((int64_t)1 << 31), "joint_forward: batch / frame count too large for one launch");
  const dim3 grid((unsigned)T, (unsigned)B);
  JointShape s{f_len, g_len, batch_offset, B, T, U, H, total_rows, packed};
  hipStream_t st = static_cast<hipStream_t>(stream);
  return CAIMAN_DISPATCH(dtype, "joint_forward", [&]() -> int {
    constexpr int VEC = 16 / sizeof(scalar_t);
    const bool aligned = (H * (int64_t)sizeof(scalar_t)) % 16 == 0 && reinterpret_cast<uintptr_t>(f) % 16 == 0 &&
                         reinterpret_cast<uintptr_t>(g) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
    auto fp = static_cast<const scalar_t*>(f);
    auto gp = static_cast<const scalar_t*>(g);
    auto op = static_cast<scalar_t*>(out);
    if (aligned)
      hipLaunchKernelGGL((joint_fwd_kernel<scalar_t, VEC>), grid, dim3(256), 0, st, fp, gp, s, relu,
                         (float)dropout_p, seed, op);
    else
      hipLaunchKernelGGL((joint_fwd_kernel<scalar_t, 1>), grid, dim3(256), 0, st, fp, gp, s, relu,
                         (float)dropout_p, seed, op);
    return check_launch("caiman_joint_forward");
  });
}

extern "C" int caiman_joint_backward(const void* dh, const void* h_out, const int32_t* f_len, const int32_t* g_len,
                                     const int64_t* batch_offset, int64_t B, int64_t T, int64_t U, int64_t H,
                                     int packed, int mask_mode, double scale, int dtype, void* df, void* dg,
                                     caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(B >= 1 && T >= 1 && U >= 1 && H >= 1, "joint_backward: bad extents");
  CAIMAN_CHECK(B <= 65535, "joint_backward: batch too large for one launch");
  CAIMAN_CHECK(mask_mode >= 0 && mask_mode <= 2, "joint_backward: bad mask mode");
  CAIMAN_CHECK(dh && f_len && g_len && df && dg && (mask_mode == 0 || h_out), "joint_backward: null pointer");
  CAIMAN_CHECK(!packed || batch_offset, "joint_backward: packed input needs batch_offset");
  JointShape s{f_len, g_len, batch_offset, B, T, U, H, 0, packed};
  hipStream_t st = static_cast<hipStream_t>(stream);
  return CAIMAN_DISPATCH(dtype, "joint_backward", [&]() -> int {
    constexpr int VEC = 16 / sizeof(scalar_t);
    const bool aligned = (H * (int64_t)sizeof(scalar_t)) % 16 == 0 && reinterpret_cast<uintptr_t>(dh) % 16 == 0 &&
                         reinterpret_cast<uintptr_t>(h_out) % 16 == 0 && reinterpret_cast<uintptr_t>(df) % 16 == 0 &&
                         reinterpret_cast<uintptr_t>(dg) % 16 == 0;
    auto dp = static_cast<const scalar_t*>(dh);
    auto hp = static_cast<const scalar_t*>(h_out);
    auto dfp = static_cast<scalar_t*>(df);
    auto dgp = static_cast<scalar_t*>(dg);
    const dim3 gt((unsigned)T, (unsigned)B), gu((unsigned)U, (unsigned)B);
    if (aligned) {
      hipLaunchKernelGGL((joint_bwd_kernel<scalar_t, VEC, true>), gt, dim3(128), 0, st, dp, hp, s, mask_mode,
                         (float)scale, dfp);
      hipLaunchKernelGGL((joint_bwd_kernel<scalar_t, VEC, false>), gu, dim3(128), 0, st, dp, hp, s, mask_mode,
                         (float)scale, dgp);
    } else {
      hipLaunchKernelGGL((joint_bwd_kernel<scalar_t, 1, true>), gt, dim3(128), 0, st, dp, hp, s, mask_mode,
                         (float)scale, dfp);
      hipLaunchKernelGGL((joint_bwd_kernel<scalar_t, 1, false>), gu, dim3(128), 0, st, dp, hp, s, mask_mode,
                         (float)scale, dgp);
    }
    return check_launch("caiman_joint_backward");
  });
}
