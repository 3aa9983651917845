// Shared host/device helpers for the gfx950 RNN-T kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <type_traits>

#include "../../include/caiman_rnnt.h"

namespace caiman {

constexpr int kWave = 64;  // CDNA4 wavefront width; never 32.

// ---- element types --------------------------------------------------------
using f16_t = _Float16;
using bf16_t = __bf16;

template <typename T>
struct acc_of {
  using type = float;
};
template <>
struct acc_of<double> {
  using type = double;
};
template <typename T>
using acc_t = typename acc_of<T>::type;

template <typename A, typename T>
__device__ __forceinline__ A to_acc(T v) {
  return static_cast<A>(v);
}
template <typename T, typename A>
__device__ __forceinline__ T from_acc(A v) {
  return static_cast<T>(v);
}

// ---- error plumbing ---------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Device-visible address of the current device's resident-LSTM failure word (csrc/lstm.hip: host-mapped memory that
// counts hand-off timeouts), or nullptr while no resident launch has been attempted on this device.
const unsigned* resident_fail_word();

#define CAIMAN_CHECK(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ::caiman::set_error(__VA_ARGS__);    \
      return CAIMAN_ERR_INVALID;           \
    }                                      \
  } while (0)

// Dispatch a generic lambda over the boundary's dtype tag.
#define CAIMAN_DISPATCH(dtype, NAME, ...)                                  \
  [&]() -> int {                                                           \
    switch (dtype) {                                                       \
      case CAIMAN_F64: { using scalar_t = double; return __VA_ARGS__(); }  \
      case CAIMAN_F32: { using scalar_t = float; return __VA_ARGS__(); }   \
      case CAIMAN_F16: { using scalar_t = ::caiman::f16_t; return __VA_ARGS__(); }  \
      case CAIMAN_BF16: { using scalar_t = ::caiman::bf16_t; return __VA_ARGS__(); } \
      default:                                                             \
        ::caiman::set_error("%s: unsupported dtype tag %d", NAME, (int)(dtype)); \
        return CAIMAN_ERR_UNSUPPORTED;                                     \
    }                                                                      \
  }()

inline size_t dtype_size(int dtype) {
  switch (dtype) {
    case CAIMAN_F64: return 8;
    case CAIMAN_F32: return 4;
    default: return 2;
  }
}

// ---- wave64 reductions ------------------------------------------------------
template <typename T, typename F>
__device__ __forceinline__ T wave_reduce(T v, F op) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off, kWave));
  return v;
}

// Block-wide reduce for blocks of NW waves; result valid in every thread.
// `smem` must hold NW elements of T.
template <int NW, typename T, typename F>
__device__ __forceinline__ T block_reduce(T v, F op, T* smem) {
  v = wave_reduce(v, op);
  if constexpr (NW == 1) return v;
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = threadIdx.x / kWave;
  __syncthreads();  // protect smem reuse across consecutive reductions
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  T r = smem[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) r = op(r, smem[i]);
  return r;
}


// Dropout without mask tensors: keep(seed, element counter) is a pure function of the seed and the element's position in
// its tensor, evaluated wherever the mask is needed (forward epilogues, and again in the backward of the layer below).
// One splitmix64 hash serves FOUR consecutive elements (16 bits each: P(drop) = floor(65536 p) / 65536): a 64-bit hash is
// three 64-bit multiplies = twelve quarter-rate 32-bit multiplies, ~250 cycles of a wave -- per element that made the
// joint's forward kernel ALU-bound (0.33 ms for 0.47 GB) and sat in the LSTM forward chain's epilogue.
__device__ __forceinline__ uint64_t drop_hash(uint64_t seed, uint64_t group) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (group + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned drop_threshold(float p) { return (unsigned)(p * 65536.f); }
__device__ __forceinline__ bool drop_pick(uint64_t z, unsigned sub, unsigned threshold) {   // true: the element is dropped
  return ((unsigned)(z >> (16u * sub)) & 0xFFFFu) < threshold;
}
// one element: 0 (dropped) or inv_keep
__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
  return drop_pick(drop_hash(seed, idx >> 2), (unsigned)(idx & 3), drop_threshold(p)) ? 0.f : inv_keep;
}
// elements idx .. idx + 3: one hash when idx is a multiple of four (every caller's case: rows of H % 4 == 0 elements)
__device__ __forceinline__ void drop_scale4(uint64_t seed, uint64_t idx, float p, float inv_keep, float (&out)[4]) {
  if ((idx & 3) == 0) {
    const uint64_t z = drop_hash(seed, idx >> 2);
    const unsigned thr = drop_threshold(p);
#pragma unroll
    for (unsigned q = 0; q < 4; ++q) out[q] = drop_pick(z, q, thr) ? 0.f : inv_keep;
  } else {
#pragma unroll
    for (unsigned q = 0; q < 4; ++q) out[q] = drop_scale(seed, idx + q, p, inv_keep);
  }
}

}  // namespace caiman
