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

}  // namespace caiman
