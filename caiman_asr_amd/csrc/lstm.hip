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
//   fwd tile : 32 batch rows x (4 gates x 4 hidden units), v_mfma_f32_16x16x32, K = H dealt
//              round-robin to the 4 waves of the workgroup;
//   bwd tile : 32 batch rows x 16 hidden units, K = 4H dealt to 16 waves;
//   both read fragment-major ("tiled") operand images straight from L2 into registers (every
//   element is used once per workgroup, so LDS staging would be pure overhead — guide §5
//   "GEMV / M<=16" row); see the MFMA section below for the layout and the measurements.
// A generic scalar kernel covers f32 / f64 and sizes the MFMA tiles do not divide.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>

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
// MFMA path (bf16 / f16, H % 32 == 0).
//
// Fragment-major ("tiled") operand images.  A v_mfma_f32_16x16x32 operand fragment is
// 16 rows x 64 bytes; read from a row-major [rows, K] matrix whose row stride is 2-8 KB that
// is 16 half-used cache lines per wave-instruction, all on one or two L2 channels.  Measured
// on MI355X (tools/lstm_microbench.hip, tools/lstm_bwd_microbench.hip): 6.1 -> 4.8 us per
// forward step and 13.0 -> 7.8 us per backward step from re-laying BOTH operands so that every
// wave-instruction reads one contiguous 1 KB block:
//   weights : tiled once per call into the workspace (they are constant over the time loop),
//   h / dG  : each step's epilogue writes, next to the row-major result the caller keeps, a
//             tiled copy [batch tile][k-step][32 rows][32] into a 2-deep ring that only the
//             next step reads.
// Every global load of a step (operands AND epilogue inputs) is issued before the first MFMA:
// the step is latency-bound, so exactly one memory round trip is exposed.
// Kernel boundaries (~1.5-2.5 us) are the per-timestep synchronisation on purpose: an in-launch
// grid barrier costs 4+ us on this chip (MI355X_MICROARCH.md, barrier-xcd row).
// ===========================================================================

// Rtile[((blk*nk + s)*16 + n)*32 + kk] = R[(gate(n)*H + blk*4 + unit(n))*H + s*32 + kk]
template <typename T>
__global__ __launch_bounds__(256) void tile_R_fwd_kernel(const T* __restrict__ R, T* __restrict__ out, int H) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)4 * H * H;
  if (i >= total) return;
  const int nk = H >> 5;
  const int kk = (int)(i & 31);
  const int n = (int)((i >> 5) & 15);
  const int64_t rest = i >> 9;
  const int s = (int)(rest % nk);
  const int blk = (int)(rest / nk);
  out[i] = R[(int64_t)((n >> 2) * H + blk * 4 + (n & 3)) * H + s * 32 + kk];
}

// Rttile[((blk*nk4 + s)*16 + n)*32 + kk] = R[row(s*32 + kk)*H + blk*16 + n]   (nk4 = 4H/32)
// row(k) = k for the reference gate layout [gate][unit]; for the interleaved layout [unit][gate]
// the GEMM's K index is k = unit*4 + gate, i.e. row(k) = (k & 3)*H + (k >> 2).
template <typename T, bool IL>
__global__ __launch_bounds__(256) void tile_Rt_bwd_kernel(const T* __restrict__ R, T* __restrict__ out, int H) {
  __shared__ T t[32][17];
  const int nk4 = (4 * H) >> 5;
  const int s = blockIdx.x % nk4, blk = blockIdx.x / nk4;
  // read R[s*32 + kk][blk*16 + n] : 32 rows x 16 cols
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int kk = e >> 4, n = e & 15;
    const int k = s * 32 + kk;
    const int row = IL ? (k & 3) * H + (k >> 2) : k;
    t[kk][n] = R[(int64_t)row * H + blk * 16 + n];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int n = e >> 5, kk = e & 31;
    out[((int64_t)blockIdx.x * 16 + n) * 32 + kk] = t[kk][n];
  }
}

// rows [B, W] row-major -> tiled [ceil(B/32)][W/32][32][32]; padding rows are left untouched
template <typename T>
__global__ __launch_bounds__(256) void tile_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int W) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * W) return;
  const int b = (int)(i / W), k = (int)(i % W);
  dst[(((int64_t)(b >> 5) * (W >> 5) + (k >> 5)) * 32 + (b & 31)) * 32 + (k & 31)] = src[i];
}

__device__ __forceinline__ int64_t tiled_index(int b, int k, int nk) {
  return (((int64_t)(b >> 5) * nk + (k >> 5)) * 32 + (b & 31)) * 32 + (k & 31);
}

// Inter-layer dropout: drop_scale / drop_scale4 (common.h) -- evaluated in the forward epilogue (to emit the masked copy the
// next layer's input GEMM reads) and again in the backward epilogue of the layer below (to mask the incoming gradient).

// A launch ("wave") can advance SEVERAL independent recurrences at once: blockIdx.z selects a slot
// (one LSTM layer working on its own timestep).  That is how a stack of layers is pipelined: layer l
// runs a chunk of timesteps behind layer l-1, so the number of dependent kernel boundaries is
// T + (L-1)*chunk instead of L*T while every launch carries L times the work.
constexpr int kMaxSlots = 8;

template <typename T>
struct FwdSlots {
  const T* Rtile[kMaxSlots];
  T* g[kMaxSlots];      // gates of the slot's first step (rows advance by B*4H per launch)
  T* c[kMaxSlots];      // c row of the slot's first step INPUT; output row = +B*H
  T* y[kMaxSlots];      // y row matching c (row-major output goes to the next row)
  T* hring[kMaxSlots];  // 2 x pad32(B) x H tiled
  int parity[kMaxSlots];  // ring half that holds h of the slot's first step input
  int nsteps[kMaxSlots];
  // optional masked copy of the output for the layer above: ymask row of the first step OUTPUT (NULL: none),
  // element counter of that row's first element, dropout probability
  T* ymask[kMaxSlots];
  uint64_t drop_base[kMaxSlots];
  float drop_p[kMaxSlots];
  int hidden[kMaxSlots];  // 0: the launch's H; otherwise this slot's own (smaller) hidden size
  uint64_t seed;
  // B <= 32 resident kernels working on a 32-row slice of a larger batch (res_batch_slice): rows of the whole batch
  // between two timesteps of g / c / y / ymask, and the batch the rings are sized for.  0: the launch's B.
  int batch_stride;
};

template <typename T>
struct BwdSlots {
  const T* Rttile[kMaxSlots];
  const T* g[kMaxSlots];      // activated gates of the slot's LAST timestep t_hi (rows go down per launch)
  const T* c[kMaxSlots];      // c row t_hi (c_prev); c_cur = +B*H
  const T* delta[kMaxSlots];  // upstream gradient row t_hi
  int64_t d_st[kMaxSlots], d_sb[kMaxSlots];
  T* dG[kMaxSlots];           // dG row t_hi
  T* dring[kMaxSlots];        // 2 x pad32(B) x 4H tiled
  float* dC[kMaxSlots];
  int parity[kMaxSlots];      // t_hi & 1
  int nsteps[kMaxSlots];
  int has_in0[kMaxSlots];     // 0 when t_hi is the last timestep of the sequence (no dG[t+1])
  // delta is the UNMASKED gradient w.r.t. the dropped-out copy of this layer's output: multiply by the same
  // keep/scale the forward applied (drop_p == 0: delta is used as is); counter of row t_hi's first element
  uint64_t drop_base[kMaxSlots];
  float drop_p[kMaxSlots];
  int hidden[kMaxSlots];  // as in FwdSlots
  float* dbias[kMaxSlots];  // NULL, or fp32 [4H] (interleaved layout): += sum over the call's dG rows
  uint64_t seed;
  int batch_stride;  // as in FwdSlots
};

// ---- forward step: grid (H/4, ceil(B/32), slots), 256 threads = 4 waves -----------------------
//   output tile: 32 batch rows x 16 gate columns (4 gates x 4 hidden units); the H/32 k-steps
//   are dealt round-robin to the 4 waves; NK = k-steps per wave (0 = runtime loop).
// IL: gates are stored [B, H, 4] (unit-major, the 4 gates of a unit adjacent) instead of the reference's
// [B, 4, H]: the epilogue then moves one 8-byte vector per (row, unit) instead of four 2-byte elements a
// whole H apart (PMC: the scattered form fetched ~40 % more HBM bytes than the algorithmic count).
template <typename T, bool HARD, int NK, bool IL>
__global__ __launch_bounds__(256) void lstm_fwd_step_mfma(FwdSlots<T> w, int step, int B, int H_launch) {
  using frag = typename frag8<T>::type;
  __shared__ float tile[4][2][16][17];
  const int slot = blockIdx.z;
  if (step >= w.nsteps[slot]) return;
  const int H = w.hidden[slot] ? w.hidden[slot] : H_launch;   // slots of one launch may differ in width
  if ((int)blockIdx.x * 4 >= H) return;
  const int64_t go = (int64_t)B * 4 * H, so = (int64_t)B * H;
  const int64_t hsz = (int64_t)((B + 31) / 32 * 32) * H;
  const T* __restrict__ Rtile = w.Rtile[slot];
  T* __restrict__ g = w.g[slot] + go * step;
  const T* __restrict__ c0 = w.c[slot] + so * step;
  T* __restrict__ c1 = w.c[slot] + so * (step + 1);
  T* __restrict__ y1 = w.y[slot] + so * (step + 1);
  const T* __restrict__ h_in = w.hring[slot] + ((w.parity[slot] + step) & 1) * hsz;
  T* __restrict__ h_out = w.hring[slot] + ((w.parity[slot] + step + 1) & 1) * hsz;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int j0 = blockIdx.x * 4, mt = blockIdx.y, m0 = mt * 32;
  const int nk = H >> 5;

  // epilogue inputs first: their latency hides under the operand loads
  const int eb = tid >> 2, eu = tid & 3;
  const int be = m0 + eb, ne = j0 + eu;
  const bool ep = (tid < 128) && (be < B);
  const int64_t gb = IL ? ((int64_t)be * H + ne) * 4 : (int64_t)be * 4 * H + ne;
  const int64_t gstep = IL ? 1 : H;
  using g4 = __attribute__((ext_vector_type(4))) T;
  float pre[4] = {0.f, 0.f, 0.f, 0.f};
  float cprev = 0.f;
  if (ep) {
    if constexpr (IL) {
      const g4 v = *reinterpret_cast<const g4*>(g + gb);
#pragma unroll
      for (int q = 0; q < 4; ++q) pre[q] = static_cast<float>(v[q]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) pre[q] = static_cast<float>(g[gb + (int64_t)q * gstep]);
    }
    cprev = static_cast<float>(c0[(int64_t)be * H + ne]);
  }

  const T* Bbase = Rtile + ((int64_t)blockIdx.x * nk * 16 + r) * 32 + 8 * kq;  // + s*512
  const T* Abase = h_in + ((int64_t)mt * nk * 32 + r) * 32 + 8 * kq;           // + s*1024 (+512: rows 16..31)
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (NK > 0) {
    // NK fragments of each operand are in flight at a time; wider layers take several such batches
    for (int i0 = 0; 4 * i0 < nk; i0 += NK) {
      frag bf[NK], a0[NK], a1[NK];
#pragma unroll
      for (int i = 0; i < NK; ++i) {
        const int s = wave + 4 * (i0 + i);
        const int sc = s < nk ? s : 0;  // past the end: clamp the address, zero the product below
        bf[i] = *reinterpret_cast<const frag*>(Bbase + (int64_t)sc * 512);
        a0[i] = *reinterpret_cast<const frag*>(Abase + (int64_t)sc * 1024);
        a1[i] = *reinterpret_cast<const frag*>(Abase + (int64_t)sc * 1024 + 512);
        if (s >= nk) {
#pragma unroll
          for (int q = 0; q < 8; ++q) bf[i][q] = static_cast<T>(0.f);
        }
      }
#pragma unroll
      for (int i = 0; i < NK; ++i) {
        acc0 = mfma16(a0[i], bf[i], acc0);
        acc1 = mfma16(a1[i], bf[i], acc1);
      }
    }
  } else {
    for (int s = wave; s < nk; s += 4) {
      const frag bb = *reinterpret_cast<const frag*>(Bbase + (int64_t)s * 512);
      const frag a0 = *reinterpret_cast<const frag*>(Abase + (int64_t)s * 1024);
      const frag a1 = *reinterpret_cast<const frag*>(Abase + (int64_t)s * 1024 + 512);
      acc0 = mfma16(a0, bb, acc0);
      acc1 = mfma16(a1, bb, acc1);
    }
  }
  // C layout 16x16: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    tile[wave][0][kq * 4 + q][r] = acc0[q];
    tile[wave][1][kq * 4 + q][r] = acc1[q];
  }
  __syncthreads();
  if (!ep) return;
  const int half = eb >> 4, rr = eb & 15;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = q * 4 + eu;
    pre[q] += tile[0][half][rr][col] + tile[1][half][rr][col] + tile[2][half][rr][col] + tile[3][half][rr][col];
  }
  const float i = Act<float, HARD>::sigm(pre[0]), f = Act<float, HARD>::sigm(pre[1]);
  const float gg = Act<float, HARD>::tanhv(pre[2]), o = Act<float, HARD>::sigm(pre[3]);
  const float c = i * gg + f * cprev;
  const T yv = static_cast<T>(o * Act<float, HARD>::tanhv(c));
  h_out[tiled_index(be, ne, nk)] = yv;  // what the next step reads
  y1[(int64_t)be * H + ne] = yv;
  if (w.ymask[slot]) {
    const float pd = w.drop_p[slot];
    const uint64_t ctr = w.drop_base[slot] + (uint64_t)step * (uint64_t)so + (uint64_t)be * H + ne;
    w.ymask[slot][so * step + (int64_t)be * H + ne] =
        static_cast<T>(static_cast<float>(yv) * drop_scale(w.seed, ctr, pd, 1.f / (1.f - pd)));
  }
  c1[(int64_t)be * H + ne] = static_cast<T>(c);
  if constexpr (IL) {
    g4 v;
    v[0] = static_cast<T>(i); v[1] = static_cast<T>(f); v[2] = static_cast<T>(gg); v[3] = static_cast<T>(o);
    *reinterpret_cast<g4*>(g + gb) = v;
  } else {
    g[gb] = static_cast<T>(i);
    g[gb + H] = static_cast<T>(f);
    g[gb + 2 * (int64_t)H] = static_cast<T>(gg);
    g[gb + 3 * (int64_t)H] = static_cast<T>(o);
  }
}

// ---- backward step: grid (H/16, ceil(B/32), slots), 1024 threads = 16 waves ---------------------
//   dh tile: 32 batch rows x 16 hidden units, K = 4H dealt round-robin to the 16 waves.
template <typename T, bool HARD, int NK, bool IL>
__global__ __launch_bounds__(1024) void lstm_bwd_step_mfma(BwdSlots<T> w, int step, int B, int H_launch) {
  using frag = typename frag8<T>::type;
  __shared__ float tile[16][2][16][17];
  const int slot = blockIdx.z;
  if (step >= w.nsteps[slot]) return;
  const int H = w.hidden[slot] ? w.hidden[slot] : H_launch;
  if ((int)blockIdx.x * 16 >= H) return;
  const int64_t go = (int64_t)B * 4 * H, so = (int64_t)B * H;
  const int64_t dsz = (int64_t)((B + 31) / 32 * 32) * 4 * H;
  const T* __restrict__ Rttile = w.Rttile[slot];
  const T* __restrict__ g = w.g[slot] - go * step;
  const T* __restrict__ c_prev = w.c[slot] - so * step;
  const T* __restrict__ c_cur = c_prev + so;
  const T* __restrict__ delta = w.delta[slot] - w.d_st[slot] * step;
  const int64_t d_sb = w.d_sb[slot];
  T* __restrict__ dG = w.dG[slot] - go * step;
  const bool has_in = step > 0 || w.has_in0[slot];
  const T* __restrict__ dG_in = w.dring[slot] + ((w.parity[slot] + step + 1) & 1) * dsz;
  T* __restrict__ dG_out = w.dring[slot] + ((w.parity[slot] + step) & 1) * dsz;
  float* __restrict__ dC = w.dC[slot];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n0 = blockIdx.x * 16, mt = blockIdx.y, m0 = mt * 32;
  const int nk4 = (4 * H) >> 5;

  const int eb = tid >> 4, eu = tid & 15;  // 32 rows x 16 units on the first 512 threads
  const int be = m0 + eb, ne = n0 + eu;
  const bool ep = (tid < 512) && (be < B);
  const int64_t gb = IL ? ((int64_t)be * H + ne) * 4 : (int64_t)be * 4 * H + ne;
  using g4 = __attribute__((ext_vector_type(4))) T;
  float dy = 0.f, gi = 0.f, gf = 0.f, gg = 0.f, go_ = 0.f, cp = 0.f, cc = 0.f, dcf = 0.f;
  if (ep) {
    dy = static_cast<float>(delta[(int64_t)be * d_sb + ne]);
    if (w.drop_p[slot] > 0.f) {
      const float pd = w.drop_p[slot];
      const uint64_t ctr = w.drop_base[slot] - (uint64_t)step * (uint64_t)so + (uint64_t)be * H + ne;
      dy *= drop_scale(w.seed, ctr, pd, 1.f / (1.f - pd));
    }
    if constexpr (IL) {
      const g4 v = *reinterpret_cast<const g4*>(g + gb);
      gi = static_cast<float>(v[0]); gf = static_cast<float>(v[1]);
      gg = static_cast<float>(v[2]); go_ = static_cast<float>(v[3]);
    } else {
      gi = static_cast<float>(g[gb]);
      gf = static_cast<float>(g[gb + H]);
      gg = static_cast<float>(g[gb + 2 * (int64_t)H]);
      go_ = static_cast<float>(g[gb + 3 * (int64_t)H]);
    }
    cp = static_cast<float>(c_prev[(int64_t)be * H + ne]);
    cc = static_cast<float>(c_cur[(int64_t)be * H + ne]);
    dcf = dC[(int64_t)be * H + ne];
  }

  if (has_in) {
    const int r = lane & 15, kq = lane >> 4;
    const T* Bbase = Rttile + ((int64_t)blockIdx.x * nk4 * 16 + r) * 32 + 8 * kq;  // + s*512
    const T* Abase = dG_in + ((int64_t)mt * nk4 * 32 + r) * 32 + 8 * kq;           // + s*1024 (+512)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NK > 0) {
      for (int i0 = 0; 16 * i0 < nk4; i0 += NK) {
        frag bf[NK], a0[NK], a1[NK];
#pragma unroll
        for (int i = 0; i < NK; ++i) {
          const int s = wave + 16 * (i0 + i);
          const int sc = s < nk4 ? s : 0;
          bf[i] = *reinterpret_cast<const frag*>(Bbase + (int64_t)sc * 512);
          a0[i] = *reinterpret_cast<const frag*>(Abase + (int64_t)sc * 1024);
          a1[i] = *reinterpret_cast<const frag*>(Abase + (int64_t)sc * 1024 + 512);
          if (s >= nk4) {
#pragma unroll
            for (int q = 0; q < 8; ++q) bf[i][q] = static_cast<T>(0.f);
          }
        }
#pragma unroll
        for (int i = 0; i < NK; ++i) {
          acc0 = mfma16(a0[i], bf[i], acc0);
          acc1 = mfma16(a1[i], bf[i], acc1);
        }
      }
    } else {
      for (int s = wave; s < nk4; s += 16) {
        const frag bb = *reinterpret_cast<const frag*>(Bbase + (int64_t)s * 512);
        const frag a0 = *reinterpret_cast<const frag*>(Abase + (int64_t)s * 1024);
        const frag a1 = *reinterpret_cast<const frag*>(Abase + (int64_t)s * 1024 + 512);
        acc0 = mfma16(a0, bb, acc0);
        acc1 = mfma16(a1, bb, acc1);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      tile[wave][0][kq * 4 + q][r] = acc0[q];
      tile[wave][1][kq * 4 + q][r] = acc1[q];
    }
  }
  __syncthreads();
  if (!ep) return;
  if (has_in) {
    const int half = eb >> 4, rr = eb & 15;
#pragma unroll
    for (int q = 0; q < 16; ++q) dy += tile[q][half][rr][eu];
  }
  const float ct = Act<float, HARD>::tanhv(cc);
  const float dc = dy * go_ * Act<float, HARD>::tanh_prime(ct) + dcf;
  const T vI = static_cast<T>(dc * gg * Act<float, HARD>::sigm_prime(gi));
  const T vF = static_cast<T>(dc * cp * Act<float, HARD>::sigm_prime(gf));
  const T vG = static_cast<T>(dc * gi * Act<float, HARD>::tanh_prime(gg));
  const T vO = static_cast<T>(dy * ct * Act<float, HARD>::sigm_prime(go_));
  if constexpr (IL) {
    g4 v;
    v[0] = vI; v[1] = vF; v[2] = vG; v[3] = vO;
    *reinterpret_cast<g4*>(dG_out + tiled_index(be, ne * 4, nk4)) = v;  // K index = unit*4 + gate
    *reinterpret_cast<g4*>(dG + gb) = v;
  } else {
    // tiled copy for the next (earlier-in-time) step
    dG_out[tiled_index(be, ne, nk4)] = vI;
    dG_out[tiled_index(be, H + ne, nk4)] = vF;
    dG_out[tiled_index(be, 2 * H + ne, nk4)] = vG;
    dG_out[tiled_index(be, 3 * H + ne, nk4)] = vO;
    dG[gb] = vI;
    dG[gb + H] = vF;
    dG[gb + 2 * (int64_t)H] = vG;
    dG[gb + 3 * (int64_t)H] = vO;
  }
  dC[(int64_t)be * H + ne] = dc * gf;
}

// Soft activations on the hardware transcendental units (v_exp_f32, v_rcp_f32) for the resident kernels, where one
// lane updates four cells per timestep inside the dependent chain: the library expf / tanhf cost 2.6 us of a 7 us
// forward timestep (phase timers, tools/lstm_resident_bench.py).  Error ~1e-6 relative, far below the storage
// type's resolution; hard activations are plain arithmetic and stay as they are.
template <bool HARD>
struct FastAct : Act<float, HARD> {};
template <>
struct FastAct<false> : Act<float, false> {
  __device__ __forceinline__ static float sigm(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
  __device__ __forceinline__ static float tanhv(float z) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * z)); }
};

// ===========================================================================
// Weight-resident chunk kernels.
//
// The step kernels above re-read every slot's recurrent matrix (8 MB at H = 1024) on every launch: with the eight
// encoder layers of a pipeline tick in flight that is 64 MB per timestep, and the launch runs at the rate the
// Infinity Cache delivers it (12 us; profiles/r01_pmc_traffic.json).  Here ONE launch covers all timesteps of the
// tick.  A slot is served by H/32 workgroups of 4 waves; a wave owns 8 hidden units (32 gate rows) and keeps their
// rows of R, all K, in registers for the whole launch (2 x H/32 MFMA A-fragments = 256 VGPRs at H = 1024), so a
// workgroup needs a CU to itself and the grid (slots x H/32 <= number of CUs) is co-resident by construction.
// Per timestep the workgroups of one slot exchange h through the y row they write anyway:
//   write-through (sc1) 8-byte stores -> every wave drains -> barrier -> one lane adds to the slot's counter;
//   one lane polls that counter relaxed, barrier, then sc1 16-byte loads of the whole row into LDS
//   (MI355X_MICROARCH.md "inter-workgroup visibility": every store and every load of the handed-off bytes is sc1,
//   first row of the table; an acquire fence + plain loads measured 0.4 us slower per hand-off).
// Rows are written once and never overwritten inside a launch, so there is no second (read-done) barrier.
// The MFMA is transposed with respect to the step kernel: A = 16 gate rows of R ordered [unit][gate], B = h^T, so
// a lane ends up with the four gates of ONE (unit, batch row) in its four accumulator registers and the cell
// update needs no cross-lane traffic; c stays in a register between steps (rounded to the storage type each step,
// as the step kernel's round trip through memory does).
// Every spin is bounded by a wall-clock limit; a timeout raises `fail` (host-visible) and the workgroup leaves.
// ===========================================================================
// wall_clock64 runs at 100 MHz: 5 s.  A hand-off normally takes microseconds, but a workgroup that has not been placed
// yet (a collective's kernel of another stream holds its CU: neither kernel can share a CU with a resident workgroup,
// whose four waves fill the register file) keeps its peers waiting for as long as that kernel runs, and a
// collective waits for the slowest rank.
constexpr int kResTimeoutTicks = 500000000;
constexpr int kResCounterStride = 32;       // one 128-byte line per slot counter
constexpr int kRes2MaxGroups = 12;           // 2-D split backward kernel: groups of 128 columns per layer (H <= 1536)
constexpr int kRes2CtrPerSlot = 4 + kRes2MaxGroups;   // its counters per slot: qc[4] + gc[groups]
constexpr int kResMaxTiles = 4;              // batch-tile kernels: 32-row tiles per workgroup
// a launch's counter block (+ the abort word): 12 counters per slot for the 2-D split backward kernel, one for the others
constexpr size_t kResSyncBytes = (size_t)(kMaxSlots * kRes2CtrPerSlot + 1) * kResCounterStride * sizeof(unsigned);
constexpr size_t kResSyncBytesBT = (size_t)(kMaxSlots * kResMaxTiles * 12 + 1) * kResCounterStride * sizeof(unsigned);
// A launch's counters start at zero.  A memset in front of every resident launch cost ~90 fill commands per step
// (4 us each plus the command-processor gap around them); instead workgroup (0, 0) of every resident launch clears the
// block that the launch HALF A POOL LATER will use.  Resident launches are totally ordered (res_begin), so that block
// belongs to no launch in flight: its last user finished half a pool ago and its next user starts after this launch
// has completed (the kernel-end release makes the stores visible to it).
// Workgroup -> (slot, position in the slot).  Default: grid (NKS, slots).  XCD-local roles: a flat grid of 8 * NKS
// workgroups, slot = block % 8: blocks are dealt round-robin over the 8 XCDs (observed, not promised -- and nothing here
// depends on it for correctness: the hand-off protocol is placement-independent), so the NKS workgroups of a slot share
// one XCD and its L2: tools/handoff_xcd_microbench.hip measures 2.29 instead of 2.66 us per 64 KB hand-off (1.96 with plain
// instead of sc1 stores, which is only valid when the co-location holds).  In the real kernels the placement alone gains
// nothing (forward 4.70 vs 4.78, backward 6.85 vs 6.77 us per timestep; training step 28.0-28.3 vs 27.9 ms: the sc1 stores
// drop the line from the XCD's L2 either way), so it is OFF by default.  Slots past the launch's count have nsteps = 0
// and leave at once.
template <int NKS>
__device__ __forceinline__ void res_role(int& slot, int& j) {
  if (gridDim.y == 1 && gridDim.x == (unsigned)(NKS * 8)) {
    slot = (int)(blockIdx.x & 7u);
    j = (int)(blockIdx.x >> 3);
  } else {
    slot = (int)blockIdx.y;
    j = (int)blockIdx.x;
  }
}
__device__ inline void res_scrub(unsigned* blk, int words) {
  if (blockIdx.x == 0 && blockIdx.y == 0)
    for (int i = threadIdx.x; i < words; i += blockDim.x)
      __hip_atomic_store(blk + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool res_wait(unsigned* cnt, unsigned target, unsigned* fail_dev, unsigned* fail_host) {
  const long long t0 = wall_clock64();
  unsigned spins = 0;
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 63u) == 0u) {
      if (__hip_atomic_load(fail_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      // after a first failure in this process every later wait gives up after 50 ms: a broken set-up cannot turn each of
      // the ~56 launches of a training step into a 5 s stall (the long limit is for a collective waiting on a slow rank)
      const long long waited = wall_clock64() - t0;
      if (waited > kResTimeoutTicks ||
          (waited > kResTimeoutTicks / 100 && (spins & 1023u) == 0u &&
           __hip_atomic_load(fail_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)) {
        __hip_atomic_store(fail_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(fail_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // (rare path) leave nothing outstanding: at the join with the normal path the compiler's waitcnt bookkeeping
        // takes the worst case, and would protect the registers of these two with vmcnt waits in the caller's hot loop
        __builtin_amdgcn_s_waitcnt(0x0F70);
        return false;
      }
    }
  }
  return true;
}

// 16-byte write-through (sc1) buffer accesses for rows that workgroups hand to each other inside a launch
__device__ __forceinline__ __amdgpu_buffer_rsrc_t res_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
template <typename T>
__device__ __forceinline__ typename frag8<T>::type res_load16(__amdgpu_buffer_rsrc_t rs, int byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);   // aux 16 = sc1
  typename frag8<T>::type f;
  __builtin_memcpy(&f, &v, 16);
  return f;
}
template <typename V>
__device__ __forceinline__ void res_store16(const V& val, __amdgpu_buffer_rsrc_t rs, int byte_off) {
  static_assert(sizeof(V) == 16, "16-byte store");
  u32x4 v;
  __builtin_memcpy(&v, &val, 16);
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, 16);
}

// PROF: thread 0 of workgroup (0, slot 0) adds the 10 ns ticks it spends per phase to fail_host[kResProfFwd + ..]
// (wait for the peers | h into LDS | MFMA + cell update | drain of the stores + barrier) and the timestep count.
constexpr int kResProfFwd = 2, kResProfBwd = 8;
template <typename T, bool HARD, int NKS, bool PROF>
__global__ __launch_bounds__(256, 1) void lstm_fwd_resident(FwdSlots<T> w, int B, unsigned* sync, unsigned* fail_host, unsigned* scrub) {
  using frag = typename frag8<T>::type;
  using g4 = __attribute__((ext_vector_type(4))) T;
  constexpr int H = NKS * 32, LDH = H + 8;   // LDS row pitch: +16 bytes keeps the 16 lanes of a b128 read on distinct banks
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* hs = reinterpret_cast<T*>(smem);                       // [32][LDH]
  T* tr = hs + 32 * LDH;                                    // [4 waves][2: h, c][32 rows][8 units]
  int* flag = reinterpret_cast<int*>(tr + 4 * 2 * 32 * 8);  // abort broadcast

  int slot, j;
  res_role<NKS>(slot, j);
  const int nsteps = w.nsteps[slot];
  res_scrub(scrub, (int)(kResSyncBytes / sizeof(unsigned)));
  if (nsteps <= 0) return;
  const int nwg = NKS;
  unsigned* cnt = sync + slot * kResCounterStride;
  unsigned* fail_dev = sync + kMaxSlots * kResCounterStride;
  const int Bs = w.batch_stride ? w.batch_stride : B;   // rows between timesteps (a slice of a larger batch: res_batch_slice)
  const int64_t go = (int64_t)Bs * 4 * H, so = (int64_t)Bs * H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int u0 = j * 32 + wave * 8;   // first hidden unit of this wave

  // resident A fragments: row m of a 16-row tile = (unit m>>2, gate m&3); tile_R_fwd_kernel stores n = gate*4 + unit
  frag wreg[2][NKS];
  {
    const T* Rt = w.Rtile[slot];
    const int n = (r & 3) * 4 + (r >> 2);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int64_t blk = (int64_t)j * 8 + wave * 2 + rt;
#pragma unroll
      for (int s = 0; s < NKS; ++s) wreg[rt][s] = *reinterpret_cast<const frag*>(Rt + ((blk * NKS + s) * 16 + n) * 32 + 8 * kq);
    }
  }
  // this lane's cells: unit u0 + rt*4 + kq, batch row ct*16 + r
  float creg[2][2];
  g4 gcur[2][2], gnext[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) gcur[i >> 1][i & 1][q] = static_cast<T>(0.f);
  {
    const T* c0 = w.c[slot];
    const T* g = w.g[slot];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int b = ct * 16 + r, u = u0 + rt * 4 + kq;
        creg[rt][ct] = b < B ? static_cast<float>(c0[(int64_t)b * H + u]) : 0.f;
        if (b < B) gcur[rt][ct] = *reinterpret_cast<const g4*>(g + ((int64_t)b * H + u) * 4);
        gnext[rt][ct] = gcur[rt][ct];
      }
  }
  if (tid == 0) *flag = 0;
  T* trh = tr + wave * (2 * 32 * 8);
  T* trc = trh + 32 * 8;
  const float pd = w.drop_p[slot];
  const float inv_keep = 1.f / (1.f - pd);

  const bool prof = PROF && tid == 0 && slot == 0 && j == 0;
  long long tp[5] = {0, 0, 0, 0, 0}, pt = 0;
  if (prof) pt = wall_clock64();
#define CAIMAN_PROF_MARK(i)                  \
  if (prof) {                                \
    const long long now_ = wall_clock64();   \
    tp[i] += now_ - pt;                      \
    pt = now_;                               \
  }
  for (int s = 0; s < nsteps; ++s) {
    T* g = w.g[slot] + go * s;
    if (s + 1 < nsteps) {   // next step's pre-activations: independent of h, in flight across the wait
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const int b = ct * 16 + r, u = u0 + rt * 4 + kq;
          if (b < B) gnext[rt][ct] = *reinterpret_cast<const g4*>(g + go + ((int64_t)b * H + u) * 4);
        }
    }
    if (s > 0 && tid == 0) {
      if (!res_wait(cnt, (unsigned)nwg * (unsigned)s, fail_dev, fail_host)) *flag = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __syncthreads();
    if (*flag) break;
    CAIMAN_PROF_MARK(0)
    // h of this step: row s of y, [B][H] row-major -> LDS (rows >= B are zero).  Every load of handed-off bytes is
    // an sc1 load to registers, which stands in for the agent acquire (visibility table, first row).
    {
      const __amdgpu_buffer_rsrc_t rs = res_rsrc(w.y[slot] + so * s);
      constexpr int PER = (32 * H / 8 + 255) / 256;   // 16-byte pieces per thread
      frag v[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int idx = tid + 256 * i, b = idx / (H / 8), k8 = idx % (H / 8);
        if (b < B) v[i] = res_load16<T>(rs, (b * H + k8 * 8) * (int)sizeof(T));
        else {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[i][q] = static_cast<T>(0.f);
        }
      }
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int idx = tid + 256 * i, b = idx / (H / 8), k8 = idx % (H / 8);
        if (b < 32) *reinterpret_cast<frag*>(hs + b * LDH + k8 * 8) = v[i];
      }
    }
    __syncthreads();
    CAIMAN_PROF_MARK(1)
    f32x4 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    // LDS reads run a batch of KB k-steps ahead of the MFMAs that consume them: left to itself the compiler, short
    // of registers, put one read and a full wait in front of every pair of MFMAs (2.7 us of exposed LDS latency per
    // timestep).  The empty asm pins the order: the next batch's reads are issued before this batch's MFMAs.
    {
      constexpr int KB = (NKS % 4 == 0) ? 4 : 2, NB_ = NKS / KB;
      frag bb[2][KB][2];
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        bb[0][i][0] = *reinterpret_cast<const frag*>(hs + r * LDH + i * 32 + kq * 8);
        bb[0][i][1] = *reinterpret_cast<const frag*>(hs + (16 + r) * LDH + i * 32 + kq * 8);
      }
#pragma unroll
      for (int nb = 0; nb < NB_; ++nb) {
        if (nb + 1 < NB_) {
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            const int ks = (nb + 1) * KB + i;
            bb[(nb + 1) & 1][i][0] = *reinterpret_cast<const frag*>(hs + r * LDH + ks * 32 + kq * 8);
            bb[(nb + 1) & 1][i][1] = *reinterpret_cast<const frag*>(hs + (16 + r) * LDH + ks * 32 + kq * 8);
          }
        }
        // the MFMAs below take their operands from this asm, so they cannot be hoisted back above the reads just issued
        if constexpr (KB == 4)
          asm volatile("" : "+v"(bb[nb & 1][0][0]), "+v"(bb[nb & 1][0][1]), "+v"(bb[nb & 1][1][0]), "+v"(bb[nb & 1][1][1]),
                            "+v"(bb[nb & 1][2][0]), "+v"(bb[nb & 1][2][1]), "+v"(bb[nb & 1][3][0]), "+v"(bb[nb & 1][3][1])
                       :: "memory");
        else
          asm volatile("" : "+v"(bb[nb & 1][0][0]), "+v"(bb[nb & 1][0][1]), "+v"(bb[nb & 1][1][0]), "+v"(bb[nb & 1][1][1])
                       :: "memory");
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const int ks = nb * KB + i;
          acc[0][0] = mfma16(wreg[0][ks], bb[nb & 1][i][0], acc[0][0]);
          acc[0][1] = mfma16(wreg[0][ks], bb[nb & 1][i][1], acc[0][1]);
          acc[1][0] = mfma16(wreg[1][ks], bb[nb & 1][i][0], acc[1][0]);
          acc[1][1] = mfma16(wreg[1][ks], bb[nb & 1][i][1], acc[1][1]);
        }
      }
    }
    // cell update, lane-local: acc register q = gate q of (unit kq of the row tile, batch row r of the column tile)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int b = ct * 16 + r, ul = rt * 4 + kq;
        const float pi = static_cast<float>(gcur[rt][ct][0]) + acc[rt][ct][0];
        const float pf = static_cast<float>(gcur[rt][ct][1]) + acc[rt][ct][1];
        const float pg = static_cast<float>(gcur[rt][ct][2]) + acc[rt][ct][2];
        const float po = static_cast<float>(gcur[rt][ct][3]) + acc[rt][ct][3];
        const float ig = FastAct<HARD>::sigm(pi), fg = FastAct<HARD>::sigm(pf);
        const float gg = FastAct<HARD>::tanhv(pg), og = FastAct<HARD>::sigm(po);
        const float c = ig * gg + fg * creg[rt][ct];
        const T cv = static_cast<T>(c);
        const T yv = static_cast<T>(og * FastAct<HARD>::tanhv(c));
        creg[rt][ct] = static_cast<float>(cv);
        trh[b * 8 + ul] = yv;
        trc[b * 8 + ul] = cv;
        if (b < B) {
          g4 v;
          v[0] = static_cast<T>(ig); v[1] = static_cast<T>(fg); v[2] = static_cast<T>(gg); v[3] = static_cast<T>(og);
          *reinterpret_cast<g4*>(g + ((int64_t)b * H + u0 + ul) * 4) = v;
        }
        gcur[rt][ct] = gnext[rt][ct];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      // lane -> (batch row lane>>1, 4 units): one 8-byte piece of the h row (write-through) and of the c row
      const int b = lane >> 1, half = lane & 1;
      if (b < B) {
        const g4 hv = *reinterpret_cast<const g4*>(trh + b * 8 + half * 4);
        const g4 cv = *reinterpret_cast<const g4*>(trc + b * 8 + half * 4);
        const int64_t e = (int64_t)b * H + u0 + half * 4;
        unsigned long long hbits;
        __builtin_memcpy(&hbits, &hv, 8);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(w.y[slot] + so * (s + 1) + e), hbits, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        *reinterpret_cast<g4*>(w.c[slot] + so * (s + 1) + e) = cv;
        if (w.ymask[slot]) {
          g4 mv;
          float ks[4];
          drop_scale4(w.seed, w.drop_base[slot] + (uint64_t)s * (uint64_t)so + (uint64_t)e, pd, inv_keep, ks);
#pragma unroll
          for (int q = 0; q < 4; ++q) mv[q] = static_cast<T>(static_cast<float>(hv[q]) * ks[q]);
          *reinterpret_cast<g4*>(w.ymask[slot] + so * s + e) = mv;
        }
        if (s == nsteps - 1) {   // leave the ring as the step kernels expect it
          const int64_t hsz = (int64_t)((Bs + 31) / 32 * 32) * H;
          T* h_out = w.hring[slot] + ((w.parity[slot] + nsteps) & 1) * hsz;
          *reinterpret_cast<g4*>(h_out + tiled_index(b, u0 + half * 4, NKS)) = hv;
        }
      }
    }
    CAIMAN_PROF_MARK(2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the workgroup signals
    __syncthreads();
    if (tid == 0 && s + 1 < nsteps) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    CAIMAN_PROF_MARK(3)
  }
  if (prof) {
    for (int i = 0; i < 4; ++i)
      __hip_atomic_fetch_add(fail_host + kResProfFwd + i, (unsigned)tp[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_add(fail_host + kResProfFwd + 4, (unsigned)nsteps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Batch tiles (B > 32): the same kernel with an inner loop over tiles of 32 batch rows.  The resident weights serve every
// tile; each tile is an INDEPENDENT recurrence with its own hand-off counter, so while the peers of tile bt are still
// on their way the workgroup multiplies tile bt + 1: the wait for the slowest peer (1.2 of the 4.8 us of a B = 32
// timestep) disappears behind the other tiles' work, and the poll of a tile's counter is issued one tile ahead.  c is
// re-read from the row the workgroup wrote a timestep earlier (rounded to the storage type there, exactly what the
// single-tile kernel keeps in its registers).  Per-timestep launches at B = 128 spend 16.4 ms per training step in the
// forward recurrence; this kernel ~ 4 tiles x 3.6 us x 560 timesteps.
template <typename T, bool HARD, int NKS>
__global__ __launch_bounds__(256, 1) void lstm_fwd_resident_bt(FwdSlots<T> w, int B, unsigned* sync, unsigned* fail_host, unsigned* scrub) {
  using frag = typename frag8<T>::type;
  using g4 = __attribute__((ext_vector_type(4))) T;
  constexpr int H = NKS * 32, LDH = H + 8;
  constexpr bool PROF = false;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* hs = reinterpret_cast<T*>(smem);                       // [32][LDH]
  T* tr = hs + 32 * LDH;                                    // [4 waves][2: h, c][32 rows][8 units]
  int* flag = reinterpret_cast<int*>(tr + 4 * 2 * 32 * 8);  // abort broadcast

  int slot, j;
  res_role<NKS>(slot, j);
  const int nsteps = w.nsteps[slot];
  res_scrub(scrub, (int)(kResSyncBytesBT / sizeof(unsigned)));
  if (nsteps <= 0) return;
  const int nwg = NKS;
  const int ntiles = (B + 31) / 32;
  unsigned* fail_dev = sync + kMaxSlots * kResMaxTiles * 12 * kResCounterStride;
  auto counter = [&](int bt) -> unsigned* { return sync + ((slot * kResMaxTiles + bt) * 12) * kResCounterStride; };
  const int64_t go = (int64_t)B * 4 * H, so = (int64_t)B * H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int u0 = j * 32 + wave * 8;   // first hidden unit of this wave

  // resident A fragments: row m of a 16-row tile = (unit m>>2, gate m&3); tile_R_fwd_kernel stores n = gate*4 + unit
  frag wreg[2][NKS];
  {
    const T* Rt = w.Rtile[slot];
    const int n = (r & 3) * 4 + (r >> 2);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int64_t blk = (int64_t)j * 8 + wave * 2 + rt;
#pragma unroll
      for (int s = 0; s < NKS; ++s) wreg[rt][s] = *reinterpret_cast<const frag*>(Rt + ((blk * NKS + s) * 16 + n) * 32 + 8 * kq);
    }
  }
  // this lane's cells: unit u0 + rt*4 + kq, batch row (32 bt) + ct*16 + r
  g4 gcur[2][2], gnext[2][2];
  auto load_gates = [&](int s_, int bt_, g4 (&dst)[2][2]) {
    const T* g_ = w.g[slot] + go * s_;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int b = bt_ * 32 + ct * 16 + r, u = u0 + rt * 4 + kq;
        if (b < B) dst[rt][ct] = *reinterpret_cast<const g4*>(g_ + ((int64_t)b * H + u) * 4);
        else {
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[rt][ct][q] = static_cast<T>(0.f);
        }
      }
  };
  load_gates(0, 0, gnext);
  unsigned pre_next = 0;   // (thread 0) the next tile-step's counter, polled a tile ahead
  if (tid == 0) *flag = 0;
  T* trh = tr + wave * (2 * 32 * 8);
  T* trc = trh + 32 * 8;
  const float pd = w.drop_p[slot];
  const float inv_keep = 1.f / (1.f - pd);

  bool aborted = false;
  for (int s = 0; s < nsteps && !aborted; ++s) {
   for (int bt = 0; bt < ntiles; ++bt) {
    T* g = w.g[slot] + go * s;
    unsigned* cnt = counter(bt);
    const int row0 = bt * 32;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) gcur[rt][ct] = gnext[rt][ct];
    // the next tile-step in program order: its pre-activations and its c row do not depend on this one
    const int bt_n = bt + 1 < ntiles ? bt + 1 : 0, s_n = bt + 1 < ntiles ? s : s + 1;
    if (s_n < nsteps) load_gates(s_n, bt_n, gnext);
    float creg[2][2];
    {
      const T* cp = w.c[slot] + so * s;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const int b = row0 + ct * 16 + r, u = u0 + rt * 4 + kq;
          creg[rt][ct] = b < B ? static_cast<float>(cp[(int64_t)b * H + u]) : 0.f;
        }
    }
    if (s > 0 && tid == 0) {
      const unsigned target = (unsigned)nwg * (unsigned)s;
      if (pre_next < target && !res_wait(cnt, target, fail_dev, fail_host)) *flag = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __syncthreads();
    if (*flag) { aborted = true; break; }
    // h of this step: row s of y, [B][H] row-major -> LDS (rows >= B are zero).  Every load of handed-off bytes is
    // an sc1 load to registers, which stands in for the agent acquire (visibility table, first row).
    {
      const __amdgpu_buffer_rsrc_t rs = res_rsrc(w.y[slot] + so * s);
      constexpr int PER = (32 * H / 8 + 255) / 256;   // 16-byte pieces per thread
      frag v[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int idx = tid + 256 * i, b = idx / (H / 8), k8 = idx % (H / 8);
        if (row0 + b < B) v[i] = res_load16<T>(rs, ((row0 + b) * H + k8 * 8) * (int)sizeof(T));
        else {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[i][q] = static_cast<T>(0.f);
        }
      }
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int idx = tid + 256 * i, b = idx / (H / 8), k8 = idx % (H / 8);
        if (b < 32) *reinterpret_cast<frag*>(hs + b * LDH + k8 * 8) = v[i];
      }
    }
    if (tid == 0 && s_n > 0 && s_n < nsteps)   // poll the next tile-step's counter now: the answer travels under the MFMAs
      pre_next = __hip_atomic_load(counter(bt_n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    f32x4 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    // LDS reads run a batch of KB k-steps ahead of the MFMAs that consume them: left to itself the compiler, short
    // of registers, put one read and a full wait in front of every pair of MFMAs (2.7 us of exposed LDS latency per
    // timestep).  The empty asm pins the order: the next batch's reads are issued before this batch's MFMAs.
    {
      constexpr int KB = (NKS % 4 == 0) ? 4 : 2, NB_ = NKS / KB;
      frag bb[2][KB][2];
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        bb[0][i][0] = *reinterpret_cast<const frag*>(hs + r * LDH + i * 32 + kq * 8);
        bb[0][i][1] = *reinterpret_cast<const frag*>(hs + (16 + r) * LDH + i * 32 + kq * 8);
      }
#pragma unroll
      for (int nb = 0; nb < NB_; ++nb) {
        if (nb + 1 < NB_) {
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            const int ks = (nb + 1) * KB + i;
            bb[(nb + 1) & 1][i][0] = *reinterpret_cast<const frag*>(hs + r * LDH + ks * 32 + kq * 8);
            bb[(nb + 1) & 1][i][1] = *reinterpret_cast<const frag*>(hs + (16 + r) * LDH + ks * 32 + kq * 8);
          }
        }
        // the MFMAs below take their operands from this asm, so they cannot be hoisted back above the reads just issued
        if constexpr (KB == 4)
          asm volatile("" : "+v"(bb[nb & 1][0][0]), "+v"(bb[nb & 1][0][1]), "+v"(bb[nb & 1][1][0]), "+v"(bb[nb & 1][1][1]),
                            "+v"(bb[nb & 1][2][0]), "+v"(bb[nb & 1][2][1]), "+v"(bb[nb & 1][3][0]), "+v"(bb[nb & 1][3][1])
                       :: "memory");
        else
          asm volatile("" : "+v"(bb[nb & 1][0][0]), "+v"(bb[nb & 1][0][1]), "+v"(bb[nb & 1][1][0]), "+v"(bb[nb & 1][1][1])
                       :: "memory");
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const int ks = nb * KB + i;
          acc[0][0] = mfma16(wreg[0][ks], bb[nb & 1][i][0], acc[0][0]);
          acc[0][1] = mfma16(wreg[0][ks], bb[nb & 1][i][1], acc[0][1]);
          acc[1][0] = mfma16(wreg[1][ks], bb[nb & 1][i][0], acc[1][0]);
          acc[1][1] = mfma16(wreg[1][ks], bb[nb & 1][i][1], acc[1][1]);
        }
      }
    }
    // cell update, lane-local: acc register q = gate q of (unit kq of the row tile, batch row r of the column tile)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int b = ct * 16 + r, ul = rt * 4 + kq;   // row within the tile
        const float pi = static_cast<float>(gcur[rt][ct][0]) + acc[rt][ct][0];
        const float pf = static_cast<float>(gcur[rt][ct][1]) + acc[rt][ct][1];
        const float pg = static_cast<float>(gcur[rt][ct][2]) + acc[rt][ct][2];
        const float po = static_cast<float>(gcur[rt][ct][3]) + acc[rt][ct][3];
        const float ig = FastAct<HARD>::sigm(pi), fg = FastAct<HARD>::sigm(pf);
        const float gg = FastAct<HARD>::tanhv(pg), og = FastAct<HARD>::sigm(po);
        const float c = ig * gg + fg * creg[rt][ct];
        const T cv = static_cast<T>(c);
        const T yv = static_cast<T>(og * FastAct<HARD>::tanhv(c));
        creg[rt][ct] = static_cast<float>(cv);
        trh[b * 8 + ul] = yv;
        trc[b * 8 + ul] = cv;
        if (row0 + b < B) {
          g4 v;
          v[0] = static_cast<T>(ig); v[1] = static_cast<T>(fg); v[2] = static_cast<T>(gg); v[3] = static_cast<T>(og);
          *reinterpret_cast<g4*>(g + ((int64_t)(row0 + b) * H + u0 + ul) * 4) = v;
        }
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      // lane -> (batch row lane>>1, 4 units): one 8-byte piece of the h row (write-through) and of the c row
      const int bl = lane >> 1, half = lane & 1, b = row0 + bl;
      if (b < B) {
        const g4 hv = *reinterpret_cast<const g4*>(trh + bl * 8 + half * 4);
        const g4 cv = *reinterpret_cast<const g4*>(trc + bl * 8 + half * 4);
        const int64_t e = (int64_t)b * H + u0 + half * 4;
        unsigned long long hbits;
        __builtin_memcpy(&hbits, &hv, 8);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(w.y[slot] + so * (s + 1) + e), hbits, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        *reinterpret_cast<g4*>(w.c[slot] + so * (s + 1) + e) = cv;
        if (w.ymask[slot]) {
          g4 mv;
          float ks[4];
          drop_scale4(w.seed, w.drop_base[slot] + (uint64_t)s * (uint64_t)so + (uint64_t)e, pd, inv_keep, ks);
#pragma unroll
          for (int q = 0; q < 4; ++q) mv[q] = static_cast<T>(static_cast<float>(hv[q]) * ks[q]);
          *reinterpret_cast<g4*>(w.ymask[slot] + so * s + e) = mv;
        }
        if (s == nsteps - 1) {   // leave the ring as the step kernels expect it
          const int64_t hsz = (int64_t)((B + 31) / 32 * 32) * H;
          T* h_out = w.hring[slot] + ((w.parity[slot] + nsteps) & 1) * hsz;
          *reinterpret_cast<g4*>(h_out + tiled_index(b, u0 + half * 4, NKS)) = hv;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the workgroup signals
    __syncthreads();
    if (tid == 0 && s + 1 < nsteps) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
  }
}

// Batch tiles with the operands of tile-step i + 1 arriving under the MFMAs of tile-step i (H = 512, 1024; 32 < B <= 128).
// lstm_fwd_resident_bt spends 1.25 of its 3.6 us per tile-step pulling the 64 KB h tile through staging registers before
// the first MFMA can issue.  Here every operand of a tile-step comes by LDS-DMA into one of TWO buffer sets -- the h tile
// (as in lstm_fwd_resident_dma) and the tile's pre-activations (32 rows x 64 B per wave) -- and c never leaves the CU (a
// lane keeps the four cells it updates in LDS, one slot per tile).  While the workgroup multiplies tile-step i out of one
// set, the DMAs of tile-step i + 1 fill the other, provided that tile's hand-off counter has already reached its target
// when tile-step i starts (one relaxed load by one lane, broadcast through LDS behind a barrier).  With 2 - 4 tiles per
// timestep it almost always has: the peers produced tile bt + 1 of the previous timestep ntiles - 1 tile-steps ago.  When
// it has not, the next tile-step spins and gathers at its start, as lstm_fwd_resident_bt does.
// No register-destination global load is left in the loop: next to LDS-DMAs in flight the compiler answers any wait for
// such a load with vmcnt(0), which would drain the prefetch (guide 5, "three .s-level traps", b).  For the same reason the
// DMA count per tile-step is fixed (a gather under a condition makes the wait counts fall back to vmcnt(0) at the join):
// when the next tile is not complete, the current one is gathered once more into the other set, harmlessly.  The buffer
// sets are separate LDS objects and the loop is unrolled by two with the sets named statically.
template <typename T, bool HARD, int NKS>
__global__ __launch_bounds__(256, 1) void lstm_fwd_resident_bt_dma(FwdSlots<T> w, int B, unsigned* sync, unsigned* fail_host, unsigned* scrub) {
  using frag = typename frag8<T>::type;
  using g4 = __attribute__((ext_vector_type(4))) T;
  constexpr int H = NKS * 32;
  constexpr int NST = H / 512, LDW = 512 + 8, KPS = 16;
  static_assert(H % 512 == 0 && NST >= 1 && NST <= 2, "double-buffered batch-tile kernel: H = 512 or 1024");
  // ONE LDS object per buffer set (stage buffers, then the pre-activations [wave][row][8 units x 4 gates]): with DMAs in flight to
  // more than two distinct LDS objects the compiler puts a vmcnt(0) in front of the next DMA (seen in the ISA), which would
  // serialise the gather
  __shared__ __attribute__((aligned(16))) T bufE[NST * 32 * LDW + 4 * 32 * 32];
  __shared__ __attribute__((aligned(16))) T bufO[NST * 32 * LDW + 4 * 32 * 32];
  __shared__ __attribute__((aligned(16))) T cells[kResMaxTiles * 256 * 4];            // [tile][thread][rt][ct]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tr = reinterpret_cast<T*>(smem);                       // [4 waves][2: h, c][32 rows][8 units]
  int* flag = reinterpret_cast<int*>(tr + 4 * 2 * 32 * 8);  // [0] abort, [1] next tile-step's rows are ready

  int slot, j;
  res_role<NKS>(slot, j);
  const int nsteps = w.nsteps[slot];
  res_scrub(scrub, (int)(kResSyncBytesBT / sizeof(unsigned)));
  if (nsteps <= 0) return;
  const int nwg = NKS;
  const int ntiles = (B + 31) / 32;
  unsigned* fail_dev = sync + kMaxSlots * kResMaxTiles * 12 * kResCounterStride;
  auto counter = [&](int bt) -> unsigned* { return sync + ((slot * kResMaxTiles + bt) * 12) * kResCounterStride; };
  const int64_t go = (int64_t)B * 4 * H, so = (int64_t)B * H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int u0 = j * 32 + wave * 8;   // first hidden unit of this wave

  frag wreg[2][NKS];
  {
    const T* Rt = w.Rtile[slot];
    const int n = (r & 3) * 4 + (r >> 2);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int64_t blk = (int64_t)j * 8 + wave * 2 + rt;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) wreg[rt][ks] = *reinterpret_cast<const frag*>(Rt + ((blk * NKS + ks) * 16 + n) * 32 + 8 * kq);
    }
  }
  // this lane's cells of every tile: unit u0 + rt*4 + kq, batch row 32 bt + ct*16 + r (rows past B: the last row's, never stored)
  {
    const T* c0 = w.c[slot];
    for (int t = 0; t < ntiles; ++t)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const int b = t * 32 + ct * 16 + r, bc = b < B ? b : B - 1;
          cells[(t * 256 + tid) * 4 + rt * 2 + ct] = c0[(int64_t)bc * H + u0 + rt * 4 + kq];
        }
  }
  if (tid == 0) { flag[0] = 0; flag[1] = 0; }
  T* trh = tr + wave * (2 * 32 * 8);
  T* trc = trh + 32 * 8;
  const float pd = w.drop_p[slot];
  const float inv_keep = 1.f / (1.f - pd);

  // operands of tile-step (s_, bt_) -> one buffer set: the h tile (8 DMA instructions per wave and stage) and this wave's
  // 32 x 64 B of pre-activations (2 instructions: lane = (row lane >> 2 of 16, 16-byte piece lane & 3)).  Rows past B re-read
  // the last row (never stored).
  auto gather = [&](T* b0, int s_, int bt_, int sg_, int btg_) {
    T* gb = b0 + NST * 32 * LDW;
    const T* src = w.y[slot] + so * s_;
#pragma unroll
    for (int q = 0; q < NST; ++q) {
      T* bq = b0 + q * 32 * LDW;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int bl = wave + 4 * i, b = bt_ * 32 + bl, bs = b < B ? b : B - 1;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + (int64_t)bs * H + q * 512 + lane * 8),
            (__attribute__((address_space(3))) void*)(bq + bl * LDW), 16, 0, 16);
      }
    }
    const T* gs = w.g[slot] + go * sg_;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int bl = i * 16 + (lane >> 2), b = btg_ * 32 + bl, bs = b < B ? b : B - 1;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(gs + ((int64_t)bs * H + u0) * 4 + (lane & 3) * 8),
          (__attribute__((address_space(3))) void*)(gb + (wave * 32 + i * 16) * 32), 16, 0, 0);
    }
  };
  // the same 18 pieces one at a time (p = 0 .. 15: piece i = p & 7 of stage p >> 3; 16, 17: the pre-activations), for the
  // tile-steps that issue them BETWEEN the MFMA blocks: a piece costs its wave 60 - 185 cycles of issue (guide, cycle table),
  // eighteen in a row in front of the first MFMA were 0.5 - 1 us of every 4 us tile-step with the matrix cores idle
  auto gather_piece = [&](T* b0, int s_, int bt_, int sg_, int btg_, int p) {
    if (p < 8 * NST) {
      const int q = p >> 3, i = p & 7;
      const int bl = wave + 4 * i, b = bt_ * 32 + bl, bs = b < B ? b : B - 1;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(w.y[slot] + so * s_ + (int64_t)bs * H + q * 512 + lane * 8),
          (__attribute__((address_space(3))) void*)(b0 + q * 32 * LDW + bl * LDW), 16, 0, 16);
    } else {
      const int i = p - 8 * NST;
      T* gb = b0 + NST * 32 * LDW;
      const int bl = i * 16 + (lane >> 2), b = btg_ * 32 + bl, bs = b < B ? b : B - 1;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(w.g[slot] + go * sg_ + ((int64_t)bs * H + u0) * 4 + (lane & 3) * 8),
          (__attribute__((address_space(3))) void*)(gb + (wave * 32 + i * 16) * 32), 16, 0, 0);
    }
  };

  int s = 0, bt = 0;
  bool have = false, aborted = false;     // have: this tile-step's operands were gathered during the previous one
#ifdef BT_PROF
  const bool prof = tid == 0 && slot == 0 && j == 0;
  long long tp[4] = {0, 0, 0, 0}, pt = 0;
  unsigned n_slow = 0, n_notready = 0, n_ts = 0;
  if (prof) pt = wall_clock64();
#define BT_MARK(i) if (prof) { const long long now_ = wall_clock64(); tp[i] += now_ - pt; pt = now_; }
#else
#define BT_MARK(i)
#endif
  auto body = [&](T* c0, T* n0) {
    const T* cg = c0 + NST * 32 * LDW;
    T* g = w.g[slot] + go * s;
    unsigned* cnt = counter(bt);
    const int row0 = bt * 32;
    const int bt_n = bt + 1 < ntiles ? bt + 1 : 0, s_n = bt + 1 < ntiles ? s : s + 1;
#ifdef BT_PROF
    if (prof) { ++n_ts; if (!have) ++n_slow; }
#endif
    if (!have) {   // (first tile-step, or the tile was not complete when the previous tile-step looked) wait, then gather now
      if (s > 0 && tid == 0) {
        if (!res_wait(cnt, (unsigned)nwg * (unsigned)s, fail_dev, fail_host)) flag[0] = 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __syncthreads();
      if (flag[0]) { aborted = true; return; }
      gather(c0, s, bt, s, bt);
      __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this wave's share has landed
    }
    // (have: the DMAs of this tile-step were issued a tile-step ago and every wave drained them before that one's last barrier)
    if (tid == 0) {   // is the next tile-step's tile complete already?  (one relaxed load; the rows themselves come by sc1 DMA)
      int rdy = 0;
      if (s_n < nsteps)
        rdy = s_n == 0 || __hip_atomic_load(counter(bt_n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)nwg * (unsigned)s_n;
      flag[1] = rdy;
    }
    // waits through the builtin, not asm: the compiler's own wait-count bookkeeping must SEE that nothing is outstanding here --
    // after an asm wait it still believes the previous tile-step's stores pending and protects their data registers with a
    // vmcnt(0) of its own in the middle of the DMAs issued below (seen in the ISA: it drained the prefetch)
    __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    const bool ready = flag[1] != 0;
    const bool more = s_n < nsteps;
#ifdef BT_PROF
    if (prof && !ready) ++n_notready;
#endif
    BT_MARK(0)
    const int gs_ = ready ? s_n : s, gbt_ = ready ? bt_n : bt, gsg_ = more ? s_n : s, gbtg_ = more ? bt_n : bt;
#ifdef BT_GATHER_UPFRONT
    gather(n0, gs_, gbt_, gsg_, gbtg_);     // flies under the MFMAs below
#endif
    f32x4 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NST; ++q) {
      const T* bq = c0 + q * 32 * LDW;
      constexpr int KB = 4, NB_ = KPS / KB;
      frag bb[2][KB][2];
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        bb[0][i][0] = *reinterpret_cast<const frag*>(bq + r * LDW + i * 32 + kq * 8);
        bb[0][i][1] = *reinterpret_cast<const frag*>(bq + (16 + r) * LDW + i * 32 + kq * 8);
      }
#pragma unroll
      for (int nb = 0; nb < NB_; ++nb) {
#ifndef BT_GATHER_UPFRONT
        {   // this block's share of the next tile-step's gather: 18 pieces over NST * NB_ blocks of 16 MFMAs
#ifndef BT_GATHER_BLOCKS
#define BT_GATHER_BLOCKS (NST * NB_)
#endif
          constexpr int NBLK = BT_GATHER_BLOCKS, NP = 8 * NST + 2;
          const int blk = q * NB_ + nb;
#pragma unroll
          for (int p = 0; p < NP; ++p)
            if (p * NBLK / NP == blk) gather_piece(n0, gs_, gbt_, gsg_, gbtg_, p);
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
        if (nb + 1 < NB_) {
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            const int ks = (nb + 1) * KB + i;
            bb[(nb + 1) & 1][i][0] = *reinterpret_cast<const frag*>(bq + r * LDW + ks * 32 + kq * 8);
            bb[(nb + 1) & 1][i][1] = *reinterpret_cast<const frag*>(bq + (16 + r) * LDW + ks * 32 + kq * 8);
          }
        }
        asm volatile("" : "+v"(bb[nb & 1][0][0]), "+v"(bb[nb & 1][0][1]), "+v"(bb[nb & 1][1][0]), "+v"(bb[nb & 1][1][1]),
                          "+v"(bb[nb & 1][2][0]), "+v"(bb[nb & 1][2][1]), "+v"(bb[nb & 1][3][0]), "+v"(bb[nb & 1][3][1])
                     :: "memory");
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const int ks = q * KPS + nb * KB + i;
          acc[0][0] = mfma16(wreg[0][ks], bb[nb & 1][i][0], acc[0][0]);
          acc[0][1] = mfma16(wreg[0][ks], bb[nb & 1][i][1], acc[0][1]);
          acc[1][0] = mfma16(wreg[1][ks], bb[nb & 1][i][0], acc[1][0]);
          acc[1][1] = mfma16(wreg[1][ks], bb[nb & 1][i][1], acc[1][1]);
        }
      }
    }
    BT_MARK(1)
    // cell update, lane-local: acc register q = gate q of (unit kq of the row tile, batch row r of the column tile)
    T* cl = cells + (bt * 256 + tid) * 4;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int b = ct * 16 + r, ul = rt * 4 + kq;   // row within the tile
        const g4 gv = *reinterpret_cast<const g4*>(cg + (wave * 32 + b) * 32 + ul * 4);
        const float pi = static_cast<float>(gv[0]) + acc[rt][ct][0];
        const float pf = static_cast<float>(gv[1]) + acc[rt][ct][1];
        const float pg = static_cast<float>(gv[2]) + acc[rt][ct][2];
        const float po = static_cast<float>(gv[3]) + acc[rt][ct][3];
        const float ig = FastAct<HARD>::sigm(pi), fg = FastAct<HARD>::sigm(pf);
        const float gg = FastAct<HARD>::tanhv(pg), og = FastAct<HARD>::sigm(po);
        const float c = ig * gg + fg * static_cast<float>(cl[rt * 2 + ct]);
        const T cv = static_cast<T>(c);
        const T yv = static_cast<T>(og * FastAct<HARD>::tanhv(c));
        cl[rt * 2 + ct] = cv;
        trh[b * 8 + ul] = yv;
        trc[b * 8 + ul] = cv;
        if (row0 + b < B) {
          g4 v;
          v[0] = static_cast<T>(ig); v[1] = static_cast<T>(fg); v[2] = static_cast<T>(gg); v[3] = static_cast<T>(og);
          *reinterpret_cast<g4*>(g + ((int64_t)(row0 + b) * H + u0 + ul) * 4) = v;
        }
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      const int bl = lane >> 1, half = lane & 1, b = row0 + bl;
      if (b < B) {
        const g4 hv = *reinterpret_cast<const g4*>(trh + bl * 8 + half * 4);
        const g4 cv = *reinterpret_cast<const g4*>(trc + bl * 8 + half * 4);
        const int64_t e = (int64_t)b * H + u0 + half * 4;
        unsigned long long hbits;
        __builtin_memcpy(&hbits, &hv, 8);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(w.y[slot] + so * (s + 1) + e), hbits, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        *reinterpret_cast<g4*>(w.c[slot] + so * (s + 1) + e) = cv;
        if (w.ymask[slot]) {
          g4 mv;
          float ks[4];
          drop_scale4(w.seed, w.drop_base[slot] + (uint64_t)s * (uint64_t)so + (uint64_t)e, pd, inv_keep, ks);
#pragma unroll
          for (int q = 0; q < 4; ++q) mv[q] = static_cast<T>(static_cast<float>(hv[q]) * ks[q]);
          *reinterpret_cast<g4*>(w.ymask[slot] + so * s + e) = mv;
        }
        if (s == nsteps - 1) {   // leave the ring as the step kernels expect it
          const int64_t hsz = (int64_t)((B + 31) / 32 * 32) * H;
          T* h_out = w.hring[slot] + ((w.parity[slot] + nsteps) & 1) * hsz;
          *reinterpret_cast<g4*>(h_out + tiled_index(b, u0 + half * 4, NKS)) = hv;
        }
      }
    }
    BT_MARK(2)
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): every storing wave drains (and the next tile-step's DMAs have landed)
    __syncthreads();
    if (tid == 0 && s + 1 < nsteps) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    BT_MARK(3)
    have = ready;
    bt = bt_n;
    s = s_n;
  };
  const int total = nsteps * ntiles;
  for (int i = 0; i < total && !aborted; i += 2) {
    body(bufE, bufO);
    if (i + 1 < total && !aborted) body(bufO, bufE);
  }
#ifdef BT_PROF
  if (prof) {
    for (int i = 0; i < 4; ++i)
      __hip_atomic_fetch_add(fail_host + kResProfFwd + i, (unsigned)tp[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_add(fail_host + kResProfFwd + 4, n_ts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_add(fail_host + kResProfBwd + 0, n_slow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_add(fail_host + kResProfBwd + 1, n_notready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#endif
#undef BT_MARK
}

// Forward resident kernel with the h row gathered by LDS-DMA (H = 512, 1024, 1536): lstm_fwd_resident pulls the row
// through staging registers (32 * H / 8 / 256 16-byte pieces per thread: 96 VGPRs at H = 1536), which, next to
// 2 * H / 32 resident weight fragments per wave (384 VGPRs at H = 1536), does not fit the 512 registers of a wave.  Here
// the row goes straight from L2 into LDS (`global_load_lds_dwordx4 ... sc1`, one 1 KB piece = 512 columns of one batch
// row per wave instruction) in H / 512 stages, each with its own LDS object, and the MFMAs of stage q run while the
// later stages are still in flight -- the gather of lstm_bwd_resident2.  Everything else (roles, hand-off protocol,
// cell update, stores) is lstm_fwd_resident's.  LDS-DMA cannot leave rows out: rows >= B read row B - 1 again and
// their results are never stored.  A layer of H = 1536 takes 48 workgroups, so five layers share the chip
// (try_fwd_resident splits a tick's slots into groups that fit).
template <typename T, bool HARD, int NKS, bool PROF, int KBW = 1>
__global__ __launch_bounds__(256, 1) void lstm_fwd_resident_dma(FwdSlots<T> w, int B, unsigned* sync, unsigned* fail_host, unsigned* scrub) {
  using frag = typename frag8<T>::type;
  using g4 = __attribute__((ext_vector_type(4))) T;
  constexpr int H = NKS * 32;
  constexpr int NST = H / 512;                             // LDS-DMA stages of 512 columns
  constexpr int LDW = 512 + 8;                             // +16 bytes: the 16 lanes of a b128 read hit distinct banks
  constexpr int KPS = 16;                                  // k-steps per stage
  static_assert(H % 512 == 0 && NST >= 1 && NST <= 3, "DMA forward kernel: H = 512, 1024 or 1536");
  __shared__ __attribute__((aligned(16))) T ring0[32 * LDW], ring1[NST > 1 ? 32 * LDW : 8], ring2[NST > 2 ? 32 * LDW : 8];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tr = reinterpret_cast<T*>(smem);                       // [4 waves][2: h, c][32 rows][8 units]
  int* flag = reinterpret_cast<int*>(tr + 4 * 2 * 32 * 8);  // abort broadcast
  auto ring = [&](int k) -> T* { return k == 0 ? ring0 : (k == 1 ? ring1 : ring2); };

  int slot, j;
  res_role<NKS>(slot, j);
  const int nsteps = w.nsteps[slot];
  res_scrub(scrub, (int)(kResSyncBytes / sizeof(unsigned)));
  if (nsteps <= 0) return;
  const int nwg = NKS;
  unsigned* cnt = sync + slot * kResCounterStride;
  unsigned* fail_dev = sync + kMaxSlots * kRes2CtrPerSlot * kResCounterStride;
  const int Bs = w.batch_stride ? w.batch_stride : B;   // rows between timesteps (a slice of a larger batch: res_batch_slice)
  const int64_t go = (int64_t)Bs * 4 * H, so = (int64_t)Bs * H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int u0 = j * 32 + wave * 8;   // first hidden unit of this wave

  // resident A fragments: row m of a 16-row tile = (unit m>>2, gate m&3); tile_R_fwd_kernel stores n = gate*4 + unit
  frag wreg[2][NKS];
  {
    const T* Rt = w.Rtile[slot];
    const int n = (r & 3) * 4 + (r >> 2);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int64_t blk = (int64_t)j * 8 + wave * 2 + rt;
#pragma unroll
      for (int s = 0; s < NKS; ++s) wreg[rt][s] = *reinterpret_cast<const frag*>(Rt + ((blk * NKS + s) * 16 + n) * 32 + 8 * kq);
    }
  }
  // this lane's cells: unit u0 + rt*4 + kq, batch row ct*16 + r
  float creg[2][2];
  g4 gcur[2][2], gnext[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) gcur[i >> 1][i & 1][q] = static_cast<T>(0.f);
  {
    const T* c0 = w.c[slot];
    const T* g = w.g[slot];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int b = ct * 16 + r, u = u0 + rt * 4 + kq;
        creg[rt][ct] = b < B ? static_cast<float>(c0[(int64_t)b * H + u]) : 0.f;
        if (b < B) gcur[rt][ct] = *reinterpret_cast<const g4*>(g + ((int64_t)b * H + u) * 4);
        gnext[rt][ct] = gcur[rt][ct];
      }
  }
  if (tid == 0) *flag = 0;
  T* trh = tr + wave * (2 * 32 * 8);
  T* trc = trh + 32 * 8;
  const float pd = w.drop_p[slot];
  const float inv_keep = 1.f / (1.f - pd);

  const bool prof = PROF && tid == 0 && slot == 0 && j == 0;
  long long tp[5] = {0, 0, 0, 0, 0}, pt = 0;
  if (prof) pt = wall_clock64();
  for (int s = 0; s < nsteps; ++s) {
    T* g = w.g[slot] + go * s;
    if (s + 1 < nsteps) {   // next step's pre-activations: independent of h, in flight across the wait
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const int b = ct * 16 + r, u = u0 + rt * 4 + kq;
          if (b < B) gnext[rt][ct] = *reinterpret_cast<const g4*>(g + go + ((int64_t)b * H + u) * 4);
        }
    }
    if (s > 0 && tid == 0) {
      if (!res_wait(cnt, (unsigned)nwg * (unsigned)s, fail_dev, fail_host)) *flag = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __syncthreads();
    if (*flag) break;
    CAIMAN_PROF_MARK(0)
    // h of this step: row s of y, [B][H] row-major -> LDS by DMA (sc1: the loads of handed-off bytes bypass the
    // non-coherent levels, visibility table, first row).  8 instructions per wave and stage: the vmcnt arithmetic
    // below counts them.
    {
      const T* src = w.y[slot] + so * s;
#pragma unroll
      for (int q = 0; q < NST; ++q) {
        T* bq = ring(q);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int b = wave + 4 * i, bs = b < B ? b : B - 1;
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(src + (int64_t)bs * H + q * 512 + lane * 8),
              (__attribute__((address_space(3))) void*)(bq + b * LDW), 16, 0, 16);
        }
      }
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NST; ++q) {
      // this wave's rows of stage q have landed (the later stages' 8 instructions each may still fly) and its LDS reads
      // of the stage before have returned; behind the bare barrier that holds for every wave.  The waits go through the
      // builtin (simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14): after an asm wait the compiler's own
      // bookkeeping still believed the previous timestep's stores pending and put a vmcnt(0) of its own behind the first
      // barrier -- every stage had landed before the first MFMA issued (seen in the ISA)
      if (q + 2 < NST) { __builtin_amdgcn_s_waitcnt(0x4070); __builtin_amdgcn_s_barrier(); }
      else if (q + 1 < NST) { __builtin_amdgcn_s_waitcnt(0x0078); __builtin_amdgcn_s_barrier(); }
      else { __builtin_amdgcn_s_waitcnt(0x0070); __builtin_amdgcn_s_barrier(); }
      if (q == NST - 1) { CAIMAN_PROF_MARK(1) }
      const T* bq = ring(q);
      constexpr int KB = NKS > 32 ? KBW : 4, NB_ = KPS / KB;
      frag bb[2][KB][2];
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        bb[0][i][0] = *reinterpret_cast<const frag*>(bq + r * LDW + i * 32 + kq * 8);
        bb[0][i][1] = *reinterpret_cast<const frag*>(bq + (16 + r) * LDW + i * 32 + kq * 8);
      }
#pragma unroll
      for (int nb = 0; nb < NB_; ++nb) {
        if (nb + 1 < NB_) {
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            const int ks = (nb + 1) * KB + i;
            bb[(nb + 1) & 1][i][0] = *reinterpret_cast<const frag*>(bq + r * LDW + ks * 32 + kq * 8);
            bb[(nb + 1) & 1][i][1] = *reinterpret_cast<const frag*>(bq + (16 + r) * LDW + ks * 32 + kq * 8);
          }
        }
        // the MFMAs below take their operands from this asm, so they cannot be hoisted back above the reads just issued
        if constexpr (KB == 4)
          asm volatile("" : "+v"(bb[nb & 1][0][0]), "+v"(bb[nb & 1][0][1]), "+v"(bb[nb & 1][1][0]), "+v"(bb[nb & 1][1][1]),
                            "+v"(bb[nb & 1][2][0]), "+v"(bb[nb & 1][2][1]), "+v"(bb[nb & 1][3][0]), "+v"(bb[nb & 1][3][1])
                       :: "memory");
        else if constexpr (KB == 2)
          asm volatile("" : "+v"(bb[nb & 1][0][0]), "+v"(bb[nb & 1][0][1]), "+v"(bb[nb & 1][1][0]), "+v"(bb[nb & 1][1][1])
                       :: "memory");
        else
          asm volatile("" : "+v"(bb[nb & 1][0][0]), "+v"(bb[nb & 1][0][1]) :: "memory");
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const int ks = q * KPS + nb * KB + i;
          acc[0][0] = mfma16(wreg[0][ks], bb[nb & 1][i][0], acc[0][0]);
          acc[0][1] = mfma16(wreg[0][ks], bb[nb & 1][i][1], acc[0][1]);
          acc[1][0] = mfma16(wreg[1][ks], bb[nb & 1][i][0], acc[1][0]);
          acc[1][1] = mfma16(wreg[1][ks], bb[nb & 1][i][1], acc[1][1]);
        }
      }
    }
    // cell update, lane-local: acc register q = gate q of (unit kq of the row tile, batch row r of the column tile)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int b = ct * 16 + r, ul = rt * 4 + kq;
        const float pi = static_cast<float>(gcur[rt][ct][0]) + acc[rt][ct][0];
        const float pf = static_cast<float>(gcur[rt][ct][1]) + acc[rt][ct][1];
        const float pg = static_cast<float>(gcur[rt][ct][2]) + acc[rt][ct][2];
        const float po = static_cast<float>(gcur[rt][ct][3]) + acc[rt][ct][3];
        const float ig = FastAct<HARD>::sigm(pi), fg = FastAct<HARD>::sigm(pf);
        const float gg = FastAct<HARD>::tanhv(pg), og = FastAct<HARD>::sigm(po);
        const float c = ig * gg + fg * creg[rt][ct];
        const T cv = static_cast<T>(c);
        const T yv = static_cast<T>(og * FastAct<HARD>::tanhv(c));
        creg[rt][ct] = static_cast<float>(cv);
        trh[b * 8 + ul] = yv;
        trc[b * 8 + ul] = cv;
        if (b < B) {
          g4 v;
          v[0] = static_cast<T>(ig); v[1] = static_cast<T>(fg); v[2] = static_cast<T>(gg); v[3] = static_cast<T>(og);
          *reinterpret_cast<g4*>(g + ((int64_t)b * H + u0 + ul) * 4) = v;
        }
        gcur[rt][ct] = gnext[rt][ct];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      // lane -> (batch row lane>>1, 4 units): one 8-byte piece of the h row (write-through) and of the c row
      const int b = lane >> 1, half = lane & 1;
      if (b < B) {
        const g4 hv = *reinterpret_cast<const g4*>(trh + b * 8 + half * 4);
        const g4 cv = *reinterpret_cast<const g4*>(trc + b * 8 + half * 4);
        const int64_t e = (int64_t)b * H + u0 + half * 4;
        unsigned long long hbits;
        __builtin_memcpy(&hbits, &hv, 8);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(w.y[slot] + so * (s + 1) + e), hbits, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        *reinterpret_cast<g4*>(w.c[slot] + so * (s + 1) + e) = cv;
        if (w.ymask[slot]) {
          g4 mv;
          float ks[4];
          drop_scale4(w.seed, w.drop_base[slot] + (uint64_t)s * (uint64_t)so + (uint64_t)e, pd, inv_keep, ks);
#pragma unroll
          for (int q = 0; q < 4; ++q) mv[q] = static_cast<T>(static_cast<float>(hv[q]) * ks[q]);
          *reinterpret_cast<g4*>(w.ymask[slot] + so * s + e) = mv;
        }
        if (s == nsteps - 1) {   // leave the ring as the step kernels expect it
          const int64_t hsz = (int64_t)((Bs + 31) / 32 * 32) * H;
          T* h_out = w.hring[slot] + ((w.parity[slot] + nsteps) & 1) * hsz;
          *reinterpret_cast<g4*>(h_out + tiled_index(b, u0 + half * 4, NKS)) = hv;
        }
      }
    }
    CAIMAN_PROF_MARK(2)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // every storing wave drains before the workgroup signals
    __syncthreads();
    if (tid == 0 && s + 1 < nsteps) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    CAIMAN_PROF_MARK(3)
  }
  if (prof) {
    for (int i = 0; i < 4; ++i)
      __hip_atomic_fetch_add(fail_host + kResProfFwd + i, (unsigned)tp[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_add(fail_host + kResProfFwd + 4, (unsigned)nsteps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Backward counterpart: dh[t] = delta[t] + dG[t+1]·R with K = 4H.  A workgroup owns 32 hidden units and keeps the
// matching 32 columns of R (as rows of Rᵀ, all 4H of K: again 256 KB at H = 1024) in registers: wave (rt, kh) holds
// the 16-unit row tile rt for half of the k-steps.  The operand every workgroup needs is the whole dG row of the
// step before (32 x 4H, 256 KB): it is streamed through LDS in stages, each wave multiplying its half of a stage
// while later stages are in flight; the two K-halves meet in LDS and 256 threads finish 4 units of one batch row
// each (32-byte dG pieces, written through for the next step's readers).  dC and the bias-gradient sums stay in
// registers across the launch.
// DMA (H >= 256): the stages are 512 columns wide (1 KB per batch row = one `global_load_lds_dwordx4` wave
// instruction) and go straight from L2 to a ring of NB LDS buffers, no staging registers: up to three stages
// (96 KB per CU) are in flight.  Narrower layers (H = 64, 128) use four stages through two sets of staging
// registers.  Either way the gather arrives at about 50 GB/s per CU with every CU pulling.
template <int NKS>
struct BwdResGeom {
  static constexpr bool DMA = NKS >= 8;
  static constexpr int H = NKS * 32;
  static constexpr int NST = DMA ? NKS / 4 : 4;                 // stages per timestep
  static constexpr int SW = 4 * H / NST;                        // columns per stage (512 with DMA)
  static constexpr int LDW = SW + 8;
  static constexpr int NB = DMA ? (NST < 4 ? NST : 4) : 2;      // LDS stage buffers
  static constexpr int HK = 2 * NKS / NST;                      // k-steps per wave and stage (the two K-halves of a stage)
};

template <typename T, bool HARD, int NKS, bool PROF>
__global__ __launch_bounds__(256, 1) void lstm_bwd_resident(BwdSlots<T> w, int B, unsigned* sync, unsigned* fail_host, unsigned* scrub) {
  using frag = typename frag8<T>::type;
  using g4 = __attribute__((ext_vector_type(4))) T;
  using G = BwdResGeom<NKS>;
  // Without DMA the dG row is streamed in NST stages of SW columns through two register sets (two stages in flight).
  constexpr bool DMA = G::DMA;
  constexpr int H = G::H, NST = G::NST, SW = G::SW, LDW = G::LDW, NB = G::NB, HK = G::HK;
  constexpr int PER = DMA ? 1 : 32 * SW / 8 / 256;     // 16-byte pieces per thread and stage (register path)
  static_assert(PER >= 1 && HK >= 1 && (!DMA || SW == 512), "stage geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // DMA: the ring buffers are four separate LDS objects so that the compiler's waitcnt pass can tell a read of one
  // from a DMA still in flight to another (with one object it drains vmcnt to 0 before every LDS read); their total
  // is a multiple of 16 bytes, so the dynamic region behind them (reduction scratch + flag) keeps its alignment.
  __shared__ __attribute__((aligned(16))) T ring0[DMA ? 32 * LDW : 8], ring1[DMA ? 32 * LDW : 8], ring2[DMA ? 32 * LDW : 8],
      ring3[DMA ? 32 * LDW : 8];
  auto ring = [&](int k) -> T* { return k == 0 ? ring0 : (k == 1 ? ring1 : (k == 2 ? ring2 : ring3)); };
  T* buf = reinterpret_cast<T*>(smem);                                  // register path: [2][32][LDW]
  float* red = reinterpret_cast<float*>(buf + (DMA ? 0 : NB * 32 * LDW));   // [kh 2][rt 2][ct 2][16 units][17]
  int* flag = reinterpret_cast<int*>(red + 8 * 16 * 17);

  int slot, j;
  res_role<NKS>(slot, j);
  const int nsteps = w.nsteps[slot];
  res_scrub(scrub, (int)(kResSyncBytes / sizeof(unsigned)));
  if (nsteps <= 0) return;
  const int nwg = NKS;
  unsigned* cnt = sync + slot * kResCounterStride;
  unsigned* fail_dev = sync + kMaxSlots * kResCounterStride;
  const int Bs = w.batch_stride ? w.batch_stride : B;   // rows between timesteps (a slice of a larger batch: res_batch_slice)
  const int64_t go = (int64_t)Bs * 4 * H, so = (int64_t)Bs * H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int rt = wave & 1, kh = wave >> 1;

  // Workgroups j, j+8, j+16, j+24 of a slot usually share an XCD (blocks are dealt round-robin): each walks the
  // stages from a different quarter of the row, so that what one has pulled into the XCD's L2 the other three find
  // there.  Placement only changes the speed; the sum over K is in a per-workgroup, fixed order either way.
  const int rot = ((j >> 3) & 3) * (NST / 4);
  frag wreg[NST][HK];
  {
    const T* Rt = w.Rttile[slot];
    const int64_t blk = (int64_t)j * 2 + rt;
#pragma unroll
    for (int q = 0; q < NST; ++q)
#pragma unroll
      for (int i = 0; i < HK; ++i) {
        const int sidx = ((q + rot) % NST) * (2 * HK) + kh * HK + i;
        wreg[q][i] = *reinterpret_cast<const frag*>(Rt + ((blk * (4 * NKS) + sidx) * 16 + r) * 32 + 8 * kq);
      }
  }
  // epilogue role: batch row eb, units u .. u+3
  const int eb = tid >> 3, ul4 = (tid & 7) * 4, u = j * 32 + ul4;
  const bool ep = eb < B;
  const int64_t eoff = (int64_t)eb * H + u;
  float dcs[4] = {0.f, 0.f, 0.f, 0.f};
  float bsum[4][4];   // bias gradient: this thread's batch row, units u..u+3 x gates, summed over the launch's timesteps
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) bsum[q][e] = 0.f;
  if (ep) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(w.dC[slot] + eoff);
#pragma unroll
    for (int q = 0; q < 4; ++q) dcs[q] = v[q];
  }
  if (tid == 0) *flag = 0;
  const float pd = w.drop_p[slot];
  const float inv_keep = 1.f / (1.f - pd);
  const int64_t d_st = w.d_st[slot], d_sb = w.d_sb[slot];

  const bool prof = PROF && tid == 0 && slot == 0 && j == 0;
  long long tp[5] = {0, 0, 0, 0, 0}, pt = 0;
  if (prof) pt = wall_clock64();
  for (int s = 0; s < nsteps; ++s) {
    const T* g = w.g[slot] - go * s;
    const T* c_prev = w.c[slot] - so * s;
    const T* delta = w.delta[slot] - d_st * s;
    T* dG = w.dG[slot] - go * s;
    const bool has_in = s > 0 || w.has_in0[slot];
    // epilogue inputs: none of them depends on the recurrence, so they travel while the workgroup waits
    frag gv0, gv1;
    g4 cpv, ccv, dlv;
    if (ep) {
      gv0 = *reinterpret_cast<const frag*>(g + eoff * 4);
      gv1 = *reinterpret_cast<const frag*>(g + eoff * 4 + 8);
      cpv = *reinterpret_cast<const g4*>(c_prev + eoff);
      ccv = *reinterpret_cast<const g4*>(c_prev + so + eoff);
      dlv = *reinterpret_cast<const g4*>(delta + (int64_t)eb * d_sb + u);
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (has_in) {
      if (s > 0 && tid == 0) {
        if (!res_wait(cnt, (unsigned)nwg * (unsigned)s, fail_dev, fail_host)) *flag = 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __syncthreads();
      if (*flag) break;
      CAIMAN_PROF_MARK(0)
      if constexpr (DMA) {
        // wave w brings rows w, w+4, .. of a stage: lane l the 16 bytes at column 8*l; a stage is 8 instructions per
        // wave.  Written-through rows + sc1 on the load: the same hand-off as the register path, minus the registers.
        const T* src = dG + go;
        auto issue = [&](int q) {
          const int pq = (q + rot) % NST;
          T* bq = ring(q % NB);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            // always 8 instructions per wave and stage (the vmcnt arithmetic below counts them); batch rows past B
            // re-read the last valid row: their columns of the product are never stored
            const int b = wave + 4 * i, bs = b < B ? b : B - 1;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + (int64_t)bs * 4 * H + pq * SW + lane * 8),
                (__attribute__((address_space(3))) void*)(bq + b * LDW), 16, 0, 16);
          }
        };
#pragma unroll
        for (int q = 0; q < NB; ++q) issue(q);
        static_assert(HK == 8, "DMA stages are 16 k-steps wide");
#pragma unroll
        for (int q = 0; q < NST; ++q) {
          // loads issued after stage q's: stages q+1 .. last; each is exactly 8 instructions of this wave
          constexpr int kNB = NB;
          const int later = q == 0 ? kNB - 1 : ((NST - 1 - q) < (kNB - 2) ? (NST - 1 - q) : (kNB - 2));
          // A bare s_barrier: __syncthreads() carries a workgroup release fence, which drains every DMA in flight.
          // This wave's rows of stage q have landed (vmcnt) and its LDS reads of stage q-1 have returned (lgkmcnt)
          // before it arrives; behind the barrier that holds for every wave.
          if (later >= 3) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          else if (later == 2) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          else if (later == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          if (q >= 1 && q - 1 + NB < NST) issue(q - 1 + NB);
          const T* bq = ring(q % NB);
          // all of the stage's fragments in flight before the first MFMA; the MFMAs take their operands from the asm,
          // so they cannot be hoisted back between the reads (see the forward kernel).  Pipelining the reads of stage q
          // under the MFMAs of stage q-1 as well changed nothing: the stage time is the DMA stream's.
          frag bb[HK][2];
#pragma unroll
          for (int i = 0; i < HK; ++i) {
            const int kk = (kh * HK + i) * 32 + kq * 8;
            bb[i][0] = *reinterpret_cast<const frag*>(bq + r * LDW + kk);
            bb[i][1] = *reinterpret_cast<const frag*>(bq + (16 + r) * LDW + kk);
          }
          asm volatile("" : "+v"(bb[0][0]), "+v"(bb[0][1]), "+v"(bb[1][0]), "+v"(bb[1][1]), "+v"(bb[2][0]), "+v"(bb[2][1]),
                            "+v"(bb[3][0]), "+v"(bb[3][1]), "+v"(bb[4][0]), "+v"(bb[4][1]), "+v"(bb[5][0]), "+v"(bb[5][1]),
                            "+v"(bb[6][0]), "+v"(bb[6][1]), "+v"(bb[7][0]), "+v"(bb[7][1])
                       :: "memory");
#pragma unroll
          for (int i = 0; i < HK; ++i) {
            acc[0] = mfma16(wreg[q][i], bb[i][0], acc[0]);
            acc[1] = mfma16(wreg[q][i], bb[i][1], acc[1]);
          }
        }
      } else {
        const __amdgpu_buffer_rsrc_t rs = res_rsrc(dG + go);   // dG of the step before: row t+1, [B][4H]
        frag v[2][PER];
  #pragma unroll
        for (int q0 = 0; q0 < 2; ++q0)
  #pragma unroll
          for (int i = 0; i < PER; ++i) {
            const int idx = tid + 256 * i, b = idx / (SW / 8), k8 = idx % (SW / 8);
            if (b < B) v[q0][i] = res_load16<T>(rs, (b * 4 * H + ((q0 + rot) % NST) * SW + k8 * 8) * (int)sizeof(T));
            else {
  #pragma unroll
              for (int e = 0; e < 8; ++e) v[q0][i][e] = static_cast<T>(0.f);
            }
          }
  #pragma unroll
        for (int q = 0; q < NST; ++q) {
          T* bq = buf + (q & 1) * (32 * LDW);
  #pragma unroll
          for (int i = 0; i < PER; ++i) {
            const int idx = tid + 256 * i, b = idx / (SW / 8), k8 = idx % (SW / 8);
            *reinterpret_cast<frag*>(bq + b * LDW + k8 * 8) = v[q & 1][i];
          }
          if (q + 2 < NST) {
  #pragma unroll
            for (int i = 0; i < PER; ++i) {
              const int idx = tid + 256 * i, b = idx / (SW / 8), k8 = idx % (SW / 8);
              if (b < B) v[q & 1][i] = res_load16<T>(rs, (b * 4 * H + ((q + 2 + rot) % NST) * SW + k8 * 8) * (int)sizeof(T));
            }
          }
          __syncthreads();
  #pragma unroll
          for (int i = 0; i < HK; ++i) {
            const int kk = (kh * HK + i) * 32 + kq * 8;
            const frag b0 = *reinterpret_cast<const frag*>(bq + r * LDW + kk);
            const frag b1 = *reinterpret_cast<const frag*>(bq + (16 + r) * LDW + kk);
            acc[0] = mfma16(wreg[q][i], b0, acc[0]);
            acc[1] = mfma16(wreg[q][i], b1, acc[1]);
          }
        }
      }
    }
    CAIMAN_PROF_MARK(1)
    // C layout: column lane&15 = batch row of the column tile, row kq*4 + reg = unit of the row tile
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[(((kh * 2 + rt) * 2 + ct) * 16 + kq * 4 + q) * 17 + r] = acc[ct][q];
    __syncthreads();
    if (ep) {
      const int ert = ul4 >> 4, ect = eb >> 4, ebl = eb & 15, eul = ul4 & 15;
      g4 vI, vF, vG, vO;
      float ks[4] = {1.f, 1.f, 1.f, 1.f};
      if (pd > 0.f) drop_scale4(w.seed, w.drop_base[slot] - (uint64_t)s * (uint64_t)so + (uint64_t)eoff, pd, inv_keep, ks);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float dy = static_cast<float>(dlv[q]) * ks[q];
        if (has_in)
          dy += red[(((0 * 2 + ert) * 2 + ect) * 16 + eul + q) * 17 + ebl] + red[(((1 * 2 + ert) * 2 + ect) * 16 + eul + q) * 17 + ebl];
        const frag& gv = q < 2 ? gv0 : gv1;
        const float gi = static_cast<float>(gv[(q & 1) * 4 + 0]), gf = static_cast<float>(gv[(q & 1) * 4 + 1]);
        const float gg = static_cast<float>(gv[(q & 1) * 4 + 2]), go_ = static_cast<float>(gv[(q & 1) * 4 + 3]);
        const float cp = static_cast<float>(cpv[q]), cc = static_cast<float>(ccv[q]);
        const float ct = FastAct<HARD>::tanhv(cc);
        const float dc = dy * go_ * FastAct<HARD>::tanh_prime(ct) + dcs[q];
        vI[q] = static_cast<T>(dc * gg * FastAct<HARD>::sigm_prime(gi));
        vF[q] = static_cast<T>(dc * cp * FastAct<HARD>::sigm_prime(gf));
        vG[q] = static_cast<T>(dc * gi * FastAct<HARD>::tanh_prime(gg));
        vO[q] = static_cast<T>(dy * ct * FastAct<HARD>::sigm_prime(go_));
        dcs[q] = dc * gf;
        bsum[q][0] += static_cast<float>(vI[q]); bsum[q][1] += static_cast<float>(vF[q]);
        bsum[q][2] += static_cast<float>(vG[q]); bsum[q][3] += static_cast<float>(vO[q]);
      }
      frag o0, o1;   // [unit][gate] interleaved: units u, u+1 | u+2, u+3
      o0[0] = vI[0]; o0[1] = vF[0]; o0[2] = vG[0]; o0[3] = vO[0]; o0[4] = vI[1]; o0[5] = vF[1]; o0[6] = vG[1]; o0[7] = vO[1];
      o1[0] = vI[2]; o1[1] = vF[2]; o1[2] = vG[2]; o1[3] = vO[2]; o1[4] = vI[3]; o1[5] = vF[3]; o1[6] = vG[3]; o1[7] = vO[3];
      const __amdgpu_buffer_rsrc_t ro = res_rsrc(dG);
      res_store16(o0, ro, (int)(eoff * 4) * (int)sizeof(T));
      res_store16(o1, ro, (int)(eoff * 4 + 8) * (int)sizeof(T));
      if (s == nsteps - 1) {   // leave the ring and dC as the step kernels expect them
        const int64_t dsz = (int64_t)((Bs + 31) / 32 * 32) * 4 * H;
        T* dG_out = w.dring[slot] + ((w.parity[slot] + s) & 1) * dsz;
        *reinterpret_cast<frag*>(dG_out + tiled_index(eb, u * 4, 4 * NKS)) = o0;
        *reinterpret_cast<frag*>(dG_out + tiled_index(eb, u * 4 + 8, 4 * NKS)) = o1;
        f32x4 dv;
#pragma unroll
        for (int q = 0; q < 4; ++q) dv[q] = dcs[q];
        *reinterpret_cast<f32x4*>(w.dC[slot] + eoff) = dv;
      }
    }
    CAIMAN_PROF_MARK(2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && s + 1 < nsteps) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    CAIMAN_PROF_MARK(3)
  }
  if (w.dbias[slot] && !*flag) {
    // sum over the 32 batch rows: lanes 8 apart hold the same units (rows wave*8 + lane/8), then the four waves meet
    // in LDS; this workgroup is the only writer of its 128 entries in this launch, launches are stream-ordered
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = bsum[q][e];
        v += __shfl_xor(v, 8, kWave);
        v += __shfl_xor(v, 16, kWave);
        v += __shfl_xor(v, 32, kWave);
        bsum[q][e] = v;
      }
    __syncthreads();
    if (lane < 8) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(wave * 8 + lane) * 16 + q * 4 + e] = bsum[q][e];
    }
    __syncthreads();
    if (tid < 128) {   // entry tid = (unit within the 32, gate): column group tid / 16, element tid % 16
      const int cg = tid >> 4, el = tid & 15;
      const float v = red[(0 * 8 + cg) * 16 + el] + red[(1 * 8 + cg) * 16 + el] + red[(2 * 8 + cg) * 16 + el] +
                      red[(3 * 8 + cg) * 16 + el];
      w.dbias[slot][(int64_t)(j * 32) * 4 + tid] += v;
    }
  }
  if (prof) {
    for (int i = 0; i < 4; ++i)
      __hip_atomic_fetch_add(fail_host + kResProfBwd + i, (unsigned)tp[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_add(fail_host + kResProfBwd + 4, (unsigned)nsteps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
#undef CAIMAN_PROF_MARK

// ===========================================================================
// Backward resident kernel, 2-D split (H = 512, 1024): the round-1 kernel above gives a workgroup 32 columns of R
// for ALL of K = 4H, so each of a layer's 32 workgroups pulls the whole dG row of the step before (256 KB at
// H = 1024) out of L2 every timestep: 4.9 of its 8.0 us (profiles/r01_bench_v13_summary.md).  Same registers, other
// shape: workgroup bx = (jq = bx / 4, kq = bx % 4) keeps the 128 columns [128 jq, +128) of R for ONE QUARTER of K
// (the gate rows of units [kq H/4, +H/4): 128 x H elements = 256 KB at H = 1024, 2 x H/32 fragments per wave as
// before) and therefore gathers a quarter of the row: 64 KB, in H/512 LDS-DMA stages.  The price is a second
// hand-off per timestep: the four workgroups of a group jq hold the four K-quarter partial sums of the same 128
// dh columns.  Wave w of every member has the partials of columns [128 jq + 32 w, +32), which is exactly what the
// member with kq = w finalises (it owns units [32 bx, +32) in the epilogue, as in the round-1 kernel), so a wave sends
// its 32 x 32 fp32 block (4 KB, write-through) to ONE peer, or keeps it when that peer is its own workgroup; a
// member receives 3 x 4 KB.  Counters, one 128-byte line each, per slot:
//   qc[q]  += 1 by each of the NKS/4 workgroups that finalise units of quarter q, once their dG piece of a timestep is
//            written (write-through, every wave drained): the consumers of quarter q (every workgroup with kq == q)
//            wait for (NKS/4) * s before gathering at timestep s;
//   gc[jq] += 1 by each member once its partial blocks are written: members wait for 4 * (round + 1).
// Both are the hand-off of the visibility table's first row (sc1 stores, drain, barrier, one lane adds; one lane polls,
// barrier, sc1 loads).  The partial buffers alternate by round parity: a member cannot finish round r + 1 before every
// peer has produced its round r + 1 block, which it does only after reading its round r blocks, so a block written in
// round r + 2 never overtakes a reader of round r.  With kq = bx % 4 the four workgroups that share an XCD (blocks are
// dealt round-robin: bx, bx + 8, .. ) want the SAME quarter: each handed-off line is fetched into an XCD's L2 once and
// only by the XCDs that need it (2.1x -> ~1x HBM-side traffic).  Placement only changes the speed.
// Sums are fp32 throughout: four K-quarter partials added in a fixed order (deterministic run to run).
// ===========================================================================
constexpr size_t kRes2PartialFloatsPerSlot = (size_t)2 * kRes2MaxGroups * 16 * 1024;   // [parity][jq][dst 4][src 4][32 x 32]
constexpr int kResProfBwd2 = 16;                        // fail_host words [16, 24): six phase sums, unused, timesteps

template <typename T, bool HARD, int NKS, bool PROF, int KBW = 1>
__global__ __launch_bounds__(256, 1) void lstm_bwd_resident2(BwdSlots<T> w, int B, unsigned* sync, unsigned* fail_host,
                                                             float* pws, unsigned* scrub) {
  using frag = typename frag8<T>::type;
  using g4 = __attribute__((ext_vector_type(4))) T;
  constexpr int H = NKS * 32;
  constexpr int NST = H / 512;                           // LDS-DMA stages of 512 columns (1 KB per batch row)
  constexpr int LDW = 512 + 8;
  constexpr int KPS = 16;                                // k-steps per stage
  constexpr int PPQ = NKS / 4;                           // workgroups that finalise units of one K quarter
  static_assert(H % 512 == 0 && NST >= 1 && NST <= 3 && NKS / 4 <= kRes2MaxGroups, "2-D split kernel: H = 512, 1024 or 1536");
  __shared__ __attribute__((aligned(16))) T ring0[32 * LDW], ring1[NST > 1 ? 32 * LDW : 8], ring2[NST > 2 ? 32 * LDW : 8];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ownp = reinterpret_cast<float*>(smem);          // [32 batch][32 units + 4]: this workgroup's own partial block
  int* flag = reinterpret_cast<int*>(ownp + 32 * 36);
  // H = 1536: 384 of a wave's 512 registers hold weights; the 16 running bias-gradient sums of a thread live in LDS
  // ([sum][thread]: conflict-free), not in registers (res_bwd2_lds sizes the dynamic part)
  constexpr bool BSUM_LDS = NKS > 32;
  float* bs_l = reinterpret_cast<float*>(flag + 4) + threadIdx.x;
  auto ring = [&](int k) -> T* { return k == 0 ? ring0 : (k == 1 ? ring1 : ring2); };

  int slot, bx;
  res_role<NKS>(slot, bx);
  const int nsteps = w.nsteps[slot];
  res_scrub(scrub, (int)(kResSyncBytes / sizeof(unsigned)));
  if (nsteps <= 0) return;
  const int kq = bx & 3, jq = bx >> 2;
  unsigned* qc_wait = sync + (slot * kRes2CtrPerSlot + kq) * kResCounterStride;
  unsigned* qc_mine = sync + (slot * kRes2CtrPerSlot + (4 * bx) / NKS) * kResCounterStride;
  unsigned* gc = sync + (slot * kRes2CtrPerSlot + 4 + jq) * kResCounterStride;
  unsigned* fail_dev = sync + kMaxSlots * kRes2CtrPerSlot * kResCounterStride;
  float* pslot = pws + (size_t)slot * kRes2PartialFloatsPerSlot;
  const int Bs = w.batch_stride ? w.batch_stride : B;   // rows between timesteps (a slice of a larger batch: res_batch_slice)
  const int64_t go = (int64_t)Bs * 4 * H, so = (int64_t)Bs * H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kg = lane >> 4;

  // resident fragments: row tiles (16 units) 8 jq + 2 wave + {0, 1}, k-steps kq NKS + [0, NKS)
  frag wreg[2][NKS];
  {
    const T* Rt = w.Rttile[slot];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int64_t blk = (int64_t)jq * 8 + wave * 2 + rt;
#pragma unroll
      for (int i = 0; i < NKS; ++i)
        wreg[rt][i] = *reinterpret_cast<const frag*>(Rt + ((blk * (4 * NKS) + kq * NKS + i) * 16 + r) * 32 + 8 * kg);
    }
  }
  // epilogue role (as in lstm_bwd_resident with j = bx): batch row eb, units u .. u+3
  const int eb = tid >> 3, ul4 = (tid & 7) * 4, u = bx * 32 + ul4;
  const bool ep = eb < B;
  const unsigned eoff = (unsigned)(eb * H + u);   // 32-bit lane offsets against wave-uniform row pointers
  float dcs[4] = {0.f, 0.f, 0.f, 0.f};
  float bsum[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bsum[q][e] = 0.f;
      if constexpr (BSUM_LDS) bs_l[(q * 4 + e) * 256] = 0.f;
    }
  if (ep) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(w.dC[slot] + eoff);
#pragma unroll
    for (int q = 0; q < 4; ++q) dcs[q] = v[q];
  }
  if (tid == 0) *flag = 0;
  const float pd = w.drop_p[slot];
  const float inv_keep = 1.f / (1.f - pd);
  const int64_t d_st = w.d_st[slot], d_sb = w.d_sb[slot];
  const int has_in0 = w.has_in0[slot];

  const bool prof = PROF && tid == 0 && slot == 0 && bx == 0;
  long long tp[6] = {0, 0, 0, 0, 0, 0}, pt = 0;
  if (prof) pt = wall_clock64();
#define CAIMAN_PROF2(i)                      \
  if (prof) {                                \
    const long long now_ = wall_clock64();   \
    tp[i] += now_ - pt;                      \
    pt = now_;                               \
  }
  for (int s = 0; s < nsteps; ++s) {
    const T* g = w.g[slot] - go * s;
    const T* c_prev = w.c[slot] - so * s;
    const T* delta = w.delta[slot] - d_st * s;
    T* dG = w.dG[slot] - go * s;
    const bool has_in = s > 0 || has_in0;
    // The epilogue's operands (none depends on the recurrence) are loaded BEHIND the DMAs of the gather, not in front of the
    // wait as in round 2: prefetched early they sat in accumulator registers that the MFMA phase wanted, and the compiler
    // protected them with a vmcnt(0) in the middle of the DMA issue -- 12 of the 16 DMAs had to land before the last four
    // were issued (seen in the ISA).  Now they travel under the MFMAs.  EXACTLY five unconditional load instructions (rows
    // past B read the last row; the results are only used under `ep`): the stage waits below count them.
    frag gv0, gv1;
    g4 cpv, ccv, dlv;
    const unsigned eoffc = (unsigned)((eb < B ? eb : B - 1) * H + u);
    auto load_epilogue_operands = [&]() {
      gv0 = *reinterpret_cast<const frag*>(g + eoffc * 4u);
      gv1 = *reinterpret_cast<const frag*>(g + eoffc * 4u + 8u);
      cpv = *reinterpret_cast<const g4*>(c_prev + eoffc);
      ccv = *reinterpret_cast<const g4*>(c_prev + so + eoffc);
      dlv = *reinterpret_cast<const g4*>(delta + (int64_t)(eb < B ? eb : B - 1) * d_sb + u);
    };
    float psum[4] = {0.f, 0.f, 0.f, 0.f};   // (dG[t+1] R) for this thread's 4 units, all of K
    if (!has_in) load_epilogue_operands();
    if (has_in) {
      if (s > 0 && tid == 0) {
        if (!res_wait(qc_wait, (unsigned)PPQ * (unsigned)s, fail_dev, fail_host)) *flag = 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __syncthreads();
      if (*flag) break;
      CAIMAN_PROF2(0)
      // ---- gather this workgroup's K quarter of dG[t+1] (32 rows x H columns) by LDS-DMA and multiply -------------
      {
        const T* src = dG + go + (int64_t)kq * H;
#pragma unroll
        for (int q = 0; q < NST; ++q) {
          T* bq = ring(q);
#pragma unroll
          for (int i = 0; i < 8; ++i) {   // always 8 instructions per wave and stage (the vmcnt arithmetic counts them)
            const int b = wave + 4 * i, bs = b < B ? b : B - 1;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + (int64_t)bs * 4 * H + q * 512 + lane * 8),
                (__attribute__((address_space(3))) void*)(bq + b * LDW), 16, 0, 16);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // issue order pinned: [8 NST DMAs][5 loads] -- the wait counts below depend on it
      load_epilogue_operands();
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[2][2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NST; ++q) {
        // this wave's rows of stage q have landed (vmcnt: the later stages' 8 instructions each and the 5 loads may still
        // fly) and its LDS reads of the stage before have returned; behind the bare barrier that holds for every wave.
        // simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 0 << 8 | vmcnt[5:4] << 14: vmcnt 21 / 13 / 5.
        if (q + 2 < NST) { __builtin_amdgcn_s_waitcnt(0x4075); __builtin_amdgcn_s_barrier(); }
        else if (q + 1 < NST) { __builtin_amdgcn_s_waitcnt(0x007D); __builtin_amdgcn_s_barrier(); }
        else { __builtin_amdgcn_s_waitcnt(0x0075); __builtin_amdgcn_s_barrier(); }
        const T* bq = ring(q);
        constexpr int KB = NKS > 32 ? KBW : 4, NB_ = KPS / KB;   // H = 1536: 384 weight registers leave room for one or two k-steps ahead
        frag bb[2][KB][2];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          bb[0][i][0] = *reinterpret_cast<const frag*>(bq + r * LDW + i * 32 + kg * 8);
          bb[0][i][1] = *reinterpret_cast<const frag*>(bq + (16 + r) * LDW + i * 32 + kg * 8);
        }
#pragma unroll
        for (int nb = 0; nb < NB_; ++nb) {
          if (nb + 1 < NB_) {
#pragma unroll
            for (int i = 0; i < KB; ++i) {
              const int ks = (nb + 1) * KB + i;
              bb[(nb + 1) & 1][i][0] = *reinterpret_cast<const frag*>(bq + r * LDW + ks * 32 + kg * 8);
              bb[(nb + 1) & 1][i][1] = *reinterpret_cast<const frag*>(bq + (16 + r) * LDW + ks * 32 + kg * 8);
            }
          }
          // the MFMAs take their operands from this asm: they cannot be hoisted above the reads just issued
          if constexpr (KB == 4)
            asm volatile("" : "+v"(bb[nb & 1][0][0]), "+v"(bb[nb & 1][0][1]), "+v"(bb[nb & 1][1][0]), "+v"(bb[nb & 1][1][1]),
                              "+v"(bb[nb & 1][2][0]), "+v"(bb[nb & 1][2][1]), "+v"(bb[nb & 1][3][0]), "+v"(bb[nb & 1][3][1])
                         :: "memory");
          else if constexpr (KB == 2)
            asm volatile("" : "+v"(bb[nb & 1][0][0]), "+v"(bb[nb & 1][0][1]), "+v"(bb[nb & 1][1][0]), "+v"(bb[nb & 1][1][1])
                         :: "memory");
          else
            asm volatile("" : "+v"(bb[nb & 1][0][0]), "+v"(bb[nb & 1][0][1]) :: "memory");
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            const int ks = q * KPS + nb * KB + i;
            acc[0][0] = mfma16(wreg[0][ks], bb[nb & 1][i][0], acc[0][0]);
            acc[0][1] = mfma16(wreg[0][ks], bb[nb & 1][i][1], acc[0][1]);
            acc[1][0] = mfma16(wreg[1][ks], bb[nb & 1][i][0], acc[1][0]);
            acc[1][1] = mfma16(wreg[1][ks], bb[nb & 1][i][1], acc[1][1]);
          }
        }
      }
      CAIMAN_PROF2(1)
      // ---- hand the partial block of columns [128 jq + 32 wave, +32) to the member that finalises them ----------------
      // C layout: column lane & 15 = batch row of the column tile, row kg * 4 + reg = unit of the row tile
      const int round = s - (has_in0 ? 0 : 1);
      float* pround = pslot + ((size_t)(round & 1) * kRes2MaxGroups + jq) * (16 * 1024);
      if (wave == kq) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            *reinterpret_cast<f32x4*>(ownp + (ct * 16 + r) * 36 + rt * 16 + kg * 4) = acc[rt][ct];
      } else {
        const __amdgpu_buffer_rsrc_t rp = res_rsrc(pround + (size_t)(wave * 4 + kq) * 1024);   // [dst = wave][src = kq]
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            res_store16(acc[rt][ct], rp, ((ct * 16 + r) * 32 + rt * 16 + kg * 4) * (int)sizeof(float));
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // every storing wave drains before the workgroup signals
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CAIMAN_PROF2(2)
        if (!res_wait(gc, 4u * (unsigned)(round + 1), fail_dev, fail_host)) *flag = 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __syncthreads();
      if (*flag) break;
      CAIMAN_PROF2(3)
      if (ep) {   // the four K-quarter partials of this thread's (batch row, 4 units), added in the order of kq
        // (one load per `src != kq` branch, each with its own wait.  Three back-to-back loads from sources kq + 1 .. kq + 3 put
        // in order by selects looked better in the ISA -- one wait instead of three -- and measured WORSE: backward recurrence
        // 4.45 -> 4.78 ms per training step in an A/B of four builds on one box, gpurun_out/r3j)
        f32x4 part[4];
#pragma unroll
        for (int src = 0; src < 4; ++src) {
          if (src == kq) {
            part[src] = *reinterpret_cast<const f32x4*>(ownp + eb * 36 + ul4);
          } else {
            const __amdgpu_buffer_rsrc_t rp = res_rsrc(pround + (size_t)(kq * 4 + src) * 1024);
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rp, (eb * 32 + ul4) * (int)sizeof(float), 0, 16);
            __builtin_memcpy(&part[src], &raw, 16);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) psum[q] = ((part[0][q] + part[1][q]) + part[2][q]) + part[3][q];
      }
    }
    if (ep) {
      g4 vI, vF, vG, vO;
      float ks[4] = {1.f, 1.f, 1.f, 1.f};
      if (pd > 0.f) drop_scale4(w.seed, w.drop_base[slot] - (uint64_t)s * (uint64_t)so + (uint64_t)eoff, pd, inv_keep, ks);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float dy = static_cast<float>(dlv[q]) * ks[q];
        dy += psum[q];
        const frag& gv = q < 2 ? gv0 : gv1;
        const float gi = static_cast<float>(gv[(q & 1) * 4 + 0]), gf = static_cast<float>(gv[(q & 1) * 4 + 1]);
        const float gg = static_cast<float>(gv[(q & 1) * 4 + 2]), go_ = static_cast<float>(gv[(q & 1) * 4 + 3]);
        const float cp = static_cast<float>(cpv[q]), cc = static_cast<float>(ccv[q]);
        const float ct = FastAct<HARD>::tanhv(cc);
        const float dc = dy * go_ * FastAct<HARD>::tanh_prime(ct) + dcs[q];
        vI[q] = static_cast<T>(dc * gg * FastAct<HARD>::sigm_prime(gi));
        vF[q] = static_cast<T>(dc * cp * FastAct<HARD>::sigm_prime(gf));
        vG[q] = static_cast<T>(dc * gi * FastAct<HARD>::tanh_prime(gg));
        vO[q] = static_cast<T>(dy * ct * FastAct<HARD>::sigm_prime(go_));
        dcs[q] = dc * gf;
        if constexpr (BSUM_LDS) {
          bs_l[(q * 4 + 0) * 256] += static_cast<float>(vI[q]); bs_l[(q * 4 + 1) * 256] += static_cast<float>(vF[q]);
          bs_l[(q * 4 + 2) * 256] += static_cast<float>(vG[q]); bs_l[(q * 4 + 3) * 256] += static_cast<float>(vO[q]);
        } else {
          bsum[q][0] += static_cast<float>(vI[q]); bsum[q][1] += static_cast<float>(vF[q]);
          bsum[q][2] += static_cast<float>(vG[q]); bsum[q][3] += static_cast<float>(vO[q]);
        }
      }
      frag o0, o1;   // [unit][gate] interleaved: units u, u+1 | u+2, u+3
      o0[0] = vI[0]; o0[1] = vF[0]; o0[2] = vG[0]; o0[3] = vO[0]; o0[4] = vI[1]; o0[5] = vF[1]; o0[6] = vG[1]; o0[7] = vO[1];
      o1[0] = vI[2]; o1[1] = vF[2]; o1[2] = vG[2]; o1[3] = vO[2]; o1[4] = vI[3]; o1[5] = vF[3]; o1[6] = vG[3]; o1[7] = vO[3];
      const __amdgpu_buffer_rsrc_t ro = res_rsrc(dG);
      res_store16(o0, ro, (int)(eoff * 4u) * (int)sizeof(T));
      res_store16(o1, ro, (int)(eoff * 4u + 8u) * (int)sizeof(T));
      if (s == nsteps - 1) {   // leave the ring and dC as the step kernels expect them
        const int64_t dsz = (int64_t)((Bs + 31) / 32 * 32) * 4 * H;
        T* dG_out = w.dring[slot] + ((w.parity[slot] + s) & 1) * dsz;
        *reinterpret_cast<frag*>(dG_out + tiled_index(eb, u * 4, 4 * NKS)) = o0;
        *reinterpret_cast<frag*>(dG_out + tiled_index(eb, u * 4 + 8, 4 * NKS)) = o1;
        f32x4 dv;
#pragma unroll
        for (int q = 0; q < 4; ++q) dv[q] = dcs[q];
        *reinterpret_cast<f32x4*>(w.dC[slot] + eoff) = dv;
      }
    }
    CAIMAN_PROF2(4)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (tid == 0 && s + 1 < nsteps) __hip_atomic_fetch_add(qc_mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    CAIMAN_PROF2(5)
  }
  if (w.dbias[slot] && !*flag) {   // as in lstm_bwd_resident: rows wave * 8 + lane / 8 hold the same units
    float* red = ownp;
    if constexpr (BSUM_LDS) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) bsum[q][e] = bs_l[(q * 4 + e) * 256];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = bsum[q][e];
        v += __shfl_xor(v, 8, kWave);
        v += __shfl_xor(v, 16, kWave);
        v += __shfl_xor(v, 32, kWave);
        bsum[q][e] = v;
      }
    __syncthreads();
    if (lane < 8) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(wave * 8 + lane) * 16 + q * 4 + e] = bsum[q][e];
    }
    __syncthreads();
    if (tid < 128) {
      const int cg = tid >> 4, el = tid & 15;
      const float v = red[(0 * 8 + cg) * 16 + el] + red[(1 * 8 + cg) * 16 + el] + red[(2 * 8 + cg) * 16 + el] +
                      red[(3 * 8 + cg) * 16 + el];
      w.dbias[slot][(int64_t)(bx * 32) * 4 + tid] += v;
    }
  }
  if (prof) {
    for (int i = 0; i < 6; ++i)
      __hip_atomic_fetch_add(fail_host + kResProfBwd2 + i, (unsigned)tp[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_add(fail_host + kResProfBwd2 + 7, (unsigned)nsteps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
#undef CAIMAN_PROF2

// Batch tiles (B > 32) of the 2-D split kernel: an inner loop over tiles of 32 batch rows, each an independent
// recurrence with its own twelve counters and partial-sum buffers (see lstm_fwd_resident_bt).  dC of a tile lives in
// its fp32 buffer between a tile's timesteps (the thread that wrote it reads it back); the bias-gradient sums run over
// all tiles in registers.
template <typename T, bool HARD, int NKS>
__global__ __launch_bounds__(256, 1) void lstm_bwd_resident2_bt(BwdSlots<T> w, int B, unsigned* sync, unsigned* fail_host,
                                                                float* pws, unsigned* scrub) {
  constexpr bool PROF = false;
  using frag = typename frag8<T>::type;
  using g4 = __attribute__((ext_vector_type(4))) T;
  constexpr int H = NKS * 32;
  constexpr int NST = H / 512;                           // LDS-DMA stages of 512 columns (1 KB per batch row)
  constexpr int LDW = 512 + 8;
  constexpr int KPS = 16;                                // k-steps per stage
  constexpr int PPQ = NKS / 4;                           // workgroups that finalise units of one K quarter
  static_assert(H % 512 == 0 && NST >= 1 && NST <= 2, "2-D split kernel: H = 512 or 1024");
  __shared__ __attribute__((aligned(16))) T ring0[32 * LDW], ring1[NST > 1 ? 32 * LDW : 8];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ownp = reinterpret_cast<float*>(smem);          // [32 batch][32 units + 4]: this workgroup's own partial block
  int* flag = reinterpret_cast<int*>(ownp + 32 * 36);
  auto ring = [&](int k) -> T* { return k == 0 ? ring0 : ring1; };

  int slot, bx;
  res_role<NKS>(slot, bx);
  const int nsteps = w.nsteps[slot];
  res_scrub(scrub, (int)(kResSyncBytesBT / sizeof(unsigned)));
  if (nsteps <= 0) return;
  const int kq = bx & 3, jq = bx >> 2;
  const int ntiles = (B + 31) / 32;
  unsigned* fail_dev = sync + kMaxSlots * kResMaxTiles * 12 * kResCounterStride;
  auto counters = [&](int bt) -> unsigned* { return sync + ((slot * kResMaxTiles + bt) * 12) * kResCounterStride; };
  const int64_t go = (int64_t)B * 4 * H, so = (int64_t)B * H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kg = lane >> 4;

  // resident fragments: row tiles (16 units) 8 jq + 2 wave + {0, 1}, k-steps kq NKS + [0, NKS)
  frag wreg[2][NKS];
  {
    const T* Rt = w.Rttile[slot];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int64_t blk = (int64_t)jq * 8 + wave * 2 + rt;
#pragma unroll
      for (int i = 0; i < NKS; ++i)
        wreg[rt][i] = *reinterpret_cast<const frag*>(Rt + ((blk * (4 * NKS) + kq * NKS + i) * 16 + r) * 32 + 8 * kg);
    }
  }
  // epilogue role (as in lstm_bwd_resident with j = bx): batch row eb, units u .. u+3
  const int ebl = tid >> 3, ul4 = (tid & 7) * 4, u = bx * 32 + ul4;   // row within the tile
  float dcs[4] = {0.f, 0.f, 0.f, 0.f};
  float bsum[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) bsum[q][e] = 0.f;
  if (tid == 0) *flag = 0;
  const float pd = w.drop_p[slot];
  const float inv_keep = 1.f / (1.f - pd);
  const int64_t d_st = w.d_st[slot], d_sb = w.d_sb[slot];
  const int has_in0 = w.has_in0[slot];

  bool aborted = false;
  unsigned pre_q = 0;   // (thread 0) the next tile-step's quarter counter, polled a tile ahead
  for (int s = 0; s < nsteps && !aborted; ++s) {
   for (int bt = 0; bt < ntiles; ++bt) {
    unsigned* ctr = counters(bt);
    unsigned* qc_wait = ctr + kq * kResCounterStride;
    unsigned* qc_mine = ctr + ((4 * bx) / NKS) * kResCounterStride;
    unsigned* gc = ctr + (4 + jq) * kResCounterStride;
    float* pslot = pws + (size_t)(slot * kResMaxTiles + bt) * kRes2PartialFloatsPerSlot;
    const int row0 = bt * 32, eb = row0 + ebl;
    const bool ep = eb < B;
    const int64_t eoff = (int64_t)eb * H + u;
    const int bt_n = bt + 1 < ntiles ? bt + 1 : 0, s_n = bt + 1 < ntiles ? s : s + 1;
    const T* g = w.g[slot] - go * s;
    const T* c_prev = w.c[slot] - so * s;
    const T* delta = w.delta[slot] - d_st * s;
    T* dG = w.dG[slot] - go * s;
    const bool has_in = s > 0 || has_in0;
    // the epilogue's operands go BEHIND the DMAs of the gather (see lstm_bwd_resident2): exactly six unconditional loads
    frag gv0, gv1;
    g4 cpv, ccv, dlv;
    f32x4 dv;
    const int ebc = eb < B ? eb : B - 1;
    const int64_t eoffc = (int64_t)ebc * H + u;
    auto load_epilogue_operands = [&]() {
      gv0 = *reinterpret_cast<const frag*>(g + eoffc * 4);
      gv1 = *reinterpret_cast<const frag*>(g + eoffc * 4 + 8);
      cpv = *reinterpret_cast<const g4*>(c_prev + eoffc);
      ccv = *reinterpret_cast<const g4*>(c_prev + so + eoffc);
      dlv = *reinterpret_cast<const g4*>(delta + (int64_t)ebc * d_sb + u);
      dv = *reinterpret_cast<const f32x4*>(w.dC[slot] + eoffc);   // written by this thread a timestep ago
    };
    float psum[4] = {0.f, 0.f, 0.f, 0.f};   // (dG[t+1] R) for this thread's 4 units, all of K
    if (!has_in) load_epilogue_operands();
    if (has_in) {
      if (s > 0 && tid == 0) {
        const unsigned target = (unsigned)PPQ * (unsigned)s;
        if (pre_q < target && !res_wait(qc_wait, target, fail_dev, fail_host)) *flag = 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __syncthreads();
      if (*flag) { aborted = true; break; }
      // ---- gather this workgroup's K quarter of dG[t+1] (32 rows x H columns) by LDS-DMA and multiply -------------
      {
        const T* src = dG + go + (int64_t)kq * H;
#pragma unroll
        for (int q = 0; q < NST; ++q) {
          T* bq = ring(q);
#pragma unroll
          for (int i = 0; i < 8; ++i) {   // always 8 instructions per wave and stage (the vmcnt arithmetic counts them)
            const int b = wave + 4 * i, bg = row0 + b, bs = bg < B ? bg : B - 1;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + (int64_t)bs * 4 * H + q * 512 + lane * 8),
                (__attribute__((address_space(3))) void*)(bq + b * LDW), 16, 0, 16);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // issue order pinned: [8 NST DMAs][6 loads] -- the wait counts below depend on it
      load_epilogue_operands();
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[2][2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NST; ++q) {
        // this wave's rows of stage q have landed (vmcnt: the later stage's 8 instructions and the 6 loads may still fly) and
        // its LDS reads of the stage before have returned; behind the bare barrier that holds for every wave (vmcnt 14 / 6)
        if (q + 1 < NST) { __builtin_amdgcn_s_waitcnt(0x007E); __builtin_amdgcn_s_barrier(); }
        else { __builtin_amdgcn_s_waitcnt(0x0076); __builtin_amdgcn_s_barrier(); }
        const T* bq = ring(q);
        constexpr int KB = 4, NB_ = KPS / KB;
        frag bb[2][KB][2];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          bb[0][i][0] = *reinterpret_cast<const frag*>(bq + r * LDW + i * 32 + kg * 8);
          bb[0][i][1] = *reinterpret_cast<const frag*>(bq + (16 + r) * LDW + i * 32 + kg * 8);
        }
#pragma unroll
        for (int nb = 0; nb < NB_; ++nb) {
          if (nb + 1 < NB_) {
#pragma unroll
            for (int i = 0; i < KB; ++i) {
              const int ks = (nb + 1) * KB + i;
              bb[(nb + 1) & 1][i][0] = *reinterpret_cast<const frag*>(bq + r * LDW + ks * 32 + kg * 8);
              bb[(nb + 1) & 1][i][1] = *reinterpret_cast<const frag*>(bq + (16 + r) * LDW + ks * 32 + kg * 8);
            }
          }
          // the MFMAs take their operands from this asm: they cannot be hoisted above the reads just issued
          asm volatile("" : "+v"(bb[nb & 1][0][0]), "+v"(bb[nb & 1][0][1]), "+v"(bb[nb & 1][1][0]), "+v"(bb[nb & 1][1][1]),
                            "+v"(bb[nb & 1][2][0]), "+v"(bb[nb & 1][2][1]), "+v"(bb[nb & 1][3][0]), "+v"(bb[nb & 1][3][1])
                       :: "memory");
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            const int ks = q * KPS + nb * KB + i;
            acc[0][0] = mfma16(wreg[0][ks], bb[nb & 1][i][0], acc[0][0]);
            acc[0][1] = mfma16(wreg[0][ks], bb[nb & 1][i][1], acc[0][1]);
            acc[1][0] = mfma16(wreg[1][ks], bb[nb & 1][i][0], acc[1][0]);
            acc[1][1] = mfma16(wreg[1][ks], bb[nb & 1][i][1], acc[1][1]);
          }
        }
      }
      if (tid == 0 && s_n > 0 && s_n < nsteps)   // poll the next tile-step's quarter counter under the hand-off below
        pre_q = __hip_atomic_load(counters(bt_n) + kq * kResCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // ---- hand the partial block of columns [128 jq + 32 wave, +32) to the member that finalises them ----------------
      // C layout: column lane & 15 = batch row of the column tile, row kg * 4 + reg = unit of the row tile
      const int round = s - (has_in0 ? 0 : 1);
      float* pround = pslot + ((size_t)(round & 1) * kRes2MaxGroups + jq) * (16 * 1024);
      if (wave == kq) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            *reinterpret_cast<f32x4*>(ownp + (ct * 16 + r) * 36 + rt * 16 + kg * 4) = acc[rt][ct];
      } else {
        const __amdgpu_buffer_rsrc_t rp = res_rsrc(pround + (size_t)(wave * 4 + kq) * 1024);   // [dst = wave][src = kq]
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            res_store16(acc[rt][ct], rp, ((ct * 16 + r) * 32 + rt * 16 + kg * 4) * (int)sizeof(float));
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // every storing wave drains before the workgroup signals
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!res_wait(gc, 4u * (unsigned)(round + 1), fail_dev, fail_host)) *flag = 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __syncthreads();
      if (*flag) { aborted = true; break; }
      if (ep) {   // the four K-quarter partials of this thread's (batch row, 4 units), added in the order of kq
        f32x4 part[4];
#pragma unroll
        for (int src = 0; src < 4; ++src) {
          if (src == kq) {
            part[src] = *reinterpret_cast<const f32x4*>(ownp + ebl * 36 + ul4);
          } else {
            const __amdgpu_buffer_rsrc_t rp = res_rsrc(pround + (size_t)(kq * 4 + src) * 1024);
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rp, (ebl * 32 + ul4) * (int)sizeof(float), 0, 16);
            __builtin_memcpy(&part[src], &raw, 16);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) psum[q] = ((part[0][q] + part[1][q]) + part[2][q]) + part[3][q];
      }
    }
    if (ep) {
#pragma unroll
      for (int q = 0; q < 4; ++q) dcs[q] = dv[q];
      g4 vI, vF, vG, vO;
      float ks[4] = {1.f, 1.f, 1.f, 1.f};
      if (pd > 0.f) drop_scale4(w.seed, w.drop_base[slot] - (uint64_t)s * (uint64_t)so + (uint64_t)eoff, pd, inv_keep, ks);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float dy = static_cast<float>(dlv[q]) * ks[q];
        dy += psum[q];
        const frag& gv = q < 2 ? gv0 : gv1;
        const float gi = static_cast<float>(gv[(q & 1) * 4 + 0]), gf = static_cast<float>(gv[(q & 1) * 4 + 1]);
        const float gg = static_cast<float>(gv[(q & 1) * 4 + 2]), go_ = static_cast<float>(gv[(q & 1) * 4 + 3]);
        const float cp = static_cast<float>(cpv[q]), cc = static_cast<float>(ccv[q]);
        const float ct = FastAct<HARD>::tanhv(cc);
        const float dc = dy * go_ * FastAct<HARD>::tanh_prime(ct) + dcs[q];
        vI[q] = static_cast<T>(dc * gg * FastAct<HARD>::sigm_prime(gi));
        vF[q] = static_cast<T>(dc * cp * FastAct<HARD>::sigm_prime(gf));
        vG[q] = static_cast<T>(dc * gi * FastAct<HARD>::tanh_prime(gg));
        vO[q] = static_cast<T>(dy * ct * FastAct<HARD>::sigm_prime(go_));
        dcs[q] = dc * gf;
        bsum[q][0] += static_cast<float>(vI[q]); bsum[q][1] += static_cast<float>(vF[q]);
        bsum[q][2] += static_cast<float>(vG[q]); bsum[q][3] += static_cast<float>(vO[q]);
      }
      frag o0, o1;   // [unit][gate] interleaved: units u, u+1 | u+2, u+3
      o0[0] = vI[0]; o0[1] = vF[0]; o0[2] = vG[0]; o0[3] = vO[0]; o0[4] = vI[1]; o0[5] = vF[1]; o0[6] = vG[1]; o0[7] = vO[1];
      o1[0] = vI[2]; o1[1] = vF[2]; o1[2] = vG[2]; o1[3] = vO[2]; o1[4] = vI[3]; o1[5] = vF[3]; o1[6] = vG[3]; o1[7] = vO[3];
      const __amdgpu_buffer_rsrc_t ro = res_rsrc(dG);
      res_store16(o0, ro, (int)(eoff * 4) * (int)sizeof(T));
      res_store16(o1, ro, (int)(eoff * 4 + 8) * (int)sizeof(T));
      if (s == nsteps - 1) {   // leave the ring and dC as the step kernels expect them
        const int64_t dsz = (int64_t)((B + 31) / 32 * 32) * 4 * H;
        T* dG_out = w.dring[slot] + ((w.parity[slot] + s) & 1) * dsz;
        *reinterpret_cast<frag*>(dG_out + tiled_index(eb, u * 4, 4 * NKS)) = o0;
        *reinterpret_cast<frag*>(dG_out + tiled_index(eb, u * 4 + 8, 4 * NKS)) = o1;
      }
      f32x4 dv;
#pragma unroll
      for (int q = 0; q < 4; ++q) dv[q] = dcs[q];
      *reinterpret_cast<f32x4*>(w.dC[slot] + eoff) = dv;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (tid == 0 && s + 1 < nsteps) __hip_atomic_fetch_add(qc_mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
  }
  if (w.dbias[slot] && !*flag) {   // as in lstm_bwd_resident: rows wave * 8 + lane / 8 hold the same units
    float* red = ownp;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = bsum[q][e];
        v += __shfl_xor(v, 8, kWave);
        v += __shfl_xor(v, 16, kWave);
        v += __shfl_xor(v, 32, kWave);
        bsum[q][e] = v;
      }
    __syncthreads();
    if (lane < 8) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(wave * 8 + lane) * 16 + q * 4 + e] = bsum[q][e];
    }
    __syncthreads();
    if (tid < 128) {
      const int cg = tid >> 4, el = tid & 15;
      const float v = red[(0 * 8 + cg) * 16 + el] + red[(1 * 8 + cg) * 16 + el] + red[(2 * 8 + cg) * 16 + el] +
                      red[(3 * 8 + cg) * 16 + el];
      w.dbias[slot][(int64_t)(bx * 32) * 4 + tid] += v;
    }
  }
}

// Backward batch tiles with the K quarter of tile-step i + 1 arriving under the MFMAs and the partial exchange of tile-step
// i (H = 512, 1024; 32 < B <= 128): lstm_bwd_resident2_bt with two buffer sets, the prefetch decision of
// lstm_fwd_resident_bt_dma (hand-off counter already at its target -> gather now, else the next tile-step waits and gathers
// as before), a fixed DMA count per tile-step and every wait through the builtin.  A tile-step whose quarter was prefetched
// starts multiplying at once: no wait for the quarter's producers, no gather in front of the first MFMA.
template <typename T, bool HARD, int NKS>
__global__ __launch_bounds__(256, 1) void lstm_bwd_resident2_bt_dma(BwdSlots<T> w, int B, unsigned* sync, unsigned* fail_host,
                                                                float* pws, unsigned* scrub) {
  constexpr bool PROF = false;
  using frag = typename frag8<T>::type;
  using g4 = __attribute__((ext_vector_type(4))) T;
  constexpr int H = NKS * 32;
  constexpr int NST = H / 512;                           // LDS-DMA stages of 512 columns (1 KB per batch row)
  constexpr int LDW = 512 + 8;
  constexpr int KPS = 16;                                // k-steps per stage
  constexpr int PPQ = NKS / 4;                           // workgroups that finalise units of one K quarter
  static_assert(H % 512 == 0 && NST >= 1 && NST <= 2, "2-D split kernel: H = 512 or 1024");
  __shared__ __attribute__((aligned(16))) T bufE[NST * 32 * LDW], bufO[NST * 32 * LDW];   // two buffer sets, one LDS object each
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ownp = reinterpret_cast<float*>(smem);          // [32 batch][32 units + 4]: this workgroup's own partial block
  int* flag = reinterpret_cast<int*>(ownp + 32 * 36);    // [0] abort, [1] the next tile-step's K quarter is complete

  int slot, bx;
  res_role<NKS>(slot, bx);
  const int nsteps = w.nsteps[slot];
  res_scrub(scrub, (int)(kResSyncBytesBT / sizeof(unsigned)));
  if (nsteps <= 0) return;
  const int kq = bx & 3, jq = bx >> 2;
  const int ntiles = (B + 31) / 32;
  unsigned* fail_dev = sync + kMaxSlots * kResMaxTiles * 12 * kResCounterStride;
  auto counters = [&](int bt) -> unsigned* { return sync + ((slot * kResMaxTiles + bt) * 12) * kResCounterStride; };
  const int64_t go = (int64_t)B * 4 * H, so = (int64_t)B * H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kg = lane >> 4;

  // resident fragments: row tiles (16 units) 8 jq + 2 wave + {0, 1}, k-steps kq NKS + [0, NKS)
  frag wreg[2][NKS];
  {
    const T* Rt = w.Rttile[slot];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int64_t blk = (int64_t)jq * 8 + wave * 2 + rt;
#pragma unroll
      for (int i = 0; i < NKS; ++i)
        wreg[rt][i] = *reinterpret_cast<const frag*>(Rt + ((blk * (4 * NKS) + kq * NKS + i) * 16 + r) * 32 + 8 * kg);
    }
  }
  // epilogue role (as in lstm_bwd_resident with j = bx): batch row eb, units u .. u+3
  const int ebl = tid >> 3, ul4 = (tid & 7) * 4, u = bx * 32 + ul4;   // row within the tile
  float dcs[4] = {0.f, 0.f, 0.f, 0.f};
  float bsum[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) bsum[q][e] = 0.f;
  if (tid == 0) { flag[0] = 0; flag[1] = 0; }
  const float pd = w.drop_p[slot];
  const float inv_keep = 1.f / (1.f - pd);
  const int64_t d_st = w.d_st[slot], d_sb = w.d_sb[slot];
  const int has_in0 = w.has_in0[slot];

  // the K quarter of a dG row block (32 rows x H columns from `src`, rows row_first ..) -> one buffer set
  auto gather = [&](T* b0, const T* src, int row_first) {
#pragma unroll
    for (int q = 0; q < NST; ++q) {
      T* bq = b0 + q * 32 * LDW;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int b = wave + 4 * i, bg = row_first + b, bs = bg < B ? bg : B - 1;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + (int64_t)bs * 4 * H + q * 512 + lane * 8),
            (__attribute__((address_space(3))) void*)(bq + b * LDW), 16, 0, 16);
      }
    }
  };
  int s = 0, bt = 0;
  bool have = false, aborted = false;   // have: this tile-step's quarter was gathered during the previous tile-step
  auto body = [&](T* c0, T* n0) {
    unsigned* ctr = counters(bt);
    unsigned* qc_wait = ctr + kq * kResCounterStride;
    unsigned* qc_mine = ctr + ((4 * bx) / NKS) * kResCounterStride;
    unsigned* gc = ctr + (4 + jq) * kResCounterStride;
    float* pslot = pws + (size_t)(slot * kResMaxTiles + bt) * kRes2PartialFloatsPerSlot;
    const int row0 = bt * 32, eb = row0 + ebl;
    const bool ep = eb < B;
    const int64_t eoff = (int64_t)eb * H + u;
    const int bt_n = bt + 1 < ntiles ? bt + 1 : 0, s_n = bt + 1 < ntiles ? s : s + 1;
    const T* g = w.g[slot] - go * s;
    const T* c_prev = w.c[slot] - so * s;
    const T* delta = w.delta[slot] - d_st * s;
    T* dG = w.dG[slot] - go * s;
    const bool has_in = s > 0 || has_in0;
    // the epilogue's operands go BEHIND the DMAs of the gather (see lstm_bwd_resident2): exactly six unconditional loads
    frag gv0, gv1;
    g4 cpv, ccv, dlv;
    f32x4 dv;
    const int ebc = eb < B ? eb : B - 1;
    const int64_t eoffc = (int64_t)ebc * H + u;
    auto load_epilogue_operands = [&]() {
      gv0 = *reinterpret_cast<const frag*>(g + eoffc * 4);
      gv1 = *reinterpret_cast<const frag*>(g + eoffc * 4 + 8);
      cpv = *reinterpret_cast<const g4*>(c_prev + eoffc);
      ccv = *reinterpret_cast<const g4*>(c_prev + so + eoffc);
      dlv = *reinterpret_cast<const g4*>(delta + (int64_t)ebc * d_sb + u);
      dv = *reinterpret_cast<const f32x4*>(w.dC[slot] + eoffc);   // written by this thread a timestep ago
    };
    float psum[4] = {0.f, 0.f, 0.f, 0.f};   // (dG[t+1] R) for this thread's 4 units, all of K
    // (1) this tile-step's K quarter: gathered during the previous tile-step (`have`), or now
    if (has_in && !have) {
      if (s > 0 && tid == 0) {
        if (!res_wait(qc_wait, (unsigned)PPQ * (unsigned)s, fail_dev, fail_host)) flag[0] = 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __syncthreads();
      if (flag[0]) { aborted = true; return; }
      gather(c0, dG + go + (int64_t)kq * H, row0);
      __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this wave's share has landed
    }
    // (2) is the NEXT tile-step's quarter complete already?  One relaxed load by one lane, broadcast behind the barrier.
    const bool next_in = s_n < nsteps && (s_n > 0 || has_in0);
    if (tid == 0) {
      int rdy = 0;
      if (next_in)
        rdy = s_n == 0 || __hip_atomic_load(counters(bt_n) + kq * kResCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >=
                              (unsigned)PPQ * (unsigned)s_n;
      flag[1] = rdy;
    }
    __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) lgkmcnt(0): visible to the compiler's bookkeeping
    __builtin_amdgcn_s_barrier();
    const bool ready = flag[1] != 0;
    // (3) ALWAYS 8 NST DMA instructions (a gather under a condition makes the wait counts fall back to vmcnt(0) at the join):
    // the next tile-step's quarter when it is complete, else this step's own dG rows (valid memory, contents irrelevant: the
    // next tile-step gathers over them after its wait).  They fly under the MFMAs and the partial exchange below.
    {
      const T* dG_n = w.dG[slot] - go * s_n;
      gather(n0, ready ? dG_n + go + (int64_t)kq * H : dG, ready ? bt_n * 32 : row0);
    }
    __builtin_amdgcn_sched_barrier(0);   // the epilogue's operands behind the DMAs (see lstm_bwd_resident2)
    load_epilogue_operands();
    __builtin_amdgcn_sched_barrier(0);
    if (has_in) {
      f32x4 acc[2][2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NST; ++q) {
        const T* bq = c0 + q * 32 * LDW;
        constexpr int KB = 4, NB_ = KPS / KB;
        frag bb[2][KB][2];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          bb[0][i][0] = *reinterpret_cast<const frag*>(bq + r * LDW + i * 32 + kg * 8);
          bb[0][i][1] = *reinterpret_cast<const frag*>(bq + (16 + r) * LDW + i * 32 + kg * 8);
        }
#pragma unroll
        for (int nb = 0; nb < NB_; ++nb) {
          if (nb + 1 < NB_) {
#pragma unroll
            for (int i = 0; i < KB; ++i) {
              const int ks = (nb + 1) * KB + i;
              bb[(nb + 1) & 1][i][0] = *reinterpret_cast<const frag*>(bq + r * LDW + ks * 32 + kg * 8);
              bb[(nb + 1) & 1][i][1] = *reinterpret_cast<const frag*>(bq + (16 + r) * LDW + ks * 32 + kg * 8);
            }
          }
          // the MFMAs take their operands from this asm: they cannot be hoisted above the reads just issued
          asm volatile("" : "+v"(bb[nb & 1][0][0]), "+v"(bb[nb & 1][0][1]), "+v"(bb[nb & 1][1][0]), "+v"(bb[nb & 1][1][1]),
                            "+v"(bb[nb & 1][2][0]), "+v"(bb[nb & 1][2][1]), "+v"(bb[nb & 1][3][0]), "+v"(bb[nb & 1][3][1])
                       :: "memory");
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            const int ks = q * KPS + nb * KB + i;
            acc[0][0] = mfma16(wreg[0][ks], bb[nb & 1][i][0], acc[0][0]);
            acc[0][1] = mfma16(wreg[0][ks], bb[nb & 1][i][1], acc[0][1]);
            acc[1][0] = mfma16(wreg[1][ks], bb[nb & 1][i][0], acc[1][0]);
            acc[1][1] = mfma16(wreg[1][ks], bb[nb & 1][i][1], acc[1][1]);
          }
        }
      }
      // ---- hand the partial block of columns [128 jq + 32 wave, +32) to the member that finalises them ----------------
      // C layout: column lane & 15 = batch row of the column tile, row kg * 4 + reg = unit of the row tile
      const int round = s - (has_in0 ? 0 : 1);
      float* pround = pslot + ((size_t)(round & 1) * kRes2MaxGroups + jq) * (16 * 1024);
      if (wave == kq) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            *reinterpret_cast<f32x4*>(ownp + (ct * 16 + r) * 36 + rt * 16 + kg * 4) = acc[rt][ct];
      } else {
        const __amdgpu_buffer_rsrc_t rp = res_rsrc(pround + (size_t)(wave * 4 + kq) * 1024);   // [dst = wave][src = kq]
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            res_store16(acc[rt][ct], rp, ((ct * 16 + r) * 32 + rt * 16 + kg * 4) * (int)sizeof(float));
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // every storing wave drains before the workgroup signals
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!res_wait(gc, 4u * (unsigned)(round + 1), fail_dev, fail_host)) flag[0] = 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __syncthreads();
      if (flag[0]) { aborted = true; return; }
      if (ep) {   // the four K-quarter partials of this thread's (batch row, 4 units), added in the order of kq
        f32x4 part[4];
#pragma unroll
        for (int src = 0; src < 4; ++src) {
          if (src == kq) {
            part[src] = *reinterpret_cast<const f32x4*>(ownp + ebl * 36 + ul4);
          } else {
            const __amdgpu_buffer_rsrc_t rp = res_rsrc(pround + (size_t)(kq * 4 + src) * 1024);
            const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rp, (ebl * 32 + ul4) * (int)sizeof(float), 0, 16);
            __builtin_memcpy(&part[src], &raw, 16);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) psum[q] = ((part[0][q] + part[1][q]) + part[2][q]) + part[3][q];
      }
    }
    if (ep) {
#pragma unroll
      for (int q = 0; q < 4; ++q) dcs[q] = dv[q];
      g4 vI, vF, vG, vO;
      float ks[4] = {1.f, 1.f, 1.f, 1.f};
      if (pd > 0.f) drop_scale4(w.seed, w.drop_base[slot] - (uint64_t)s * (uint64_t)so + (uint64_t)eoff, pd, inv_keep, ks);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float dy = static_cast<float>(dlv[q]) * ks[q];
        dy += psum[q];
        const frag& gv = q < 2 ? gv0 : gv1;
        const float gi = static_cast<float>(gv[(q & 1) * 4 + 0]), gf = static_cast<float>(gv[(q & 1) * 4 + 1]);
        const float gg = static_cast<float>(gv[(q & 1) * 4 + 2]), go_ = static_cast<float>(gv[(q & 1) * 4 + 3]);
        const float cp = static_cast<float>(cpv[q]), cc = static_cast<float>(ccv[q]);
        const float ct = FastAct<HARD>::tanhv(cc);
        const float dc = dy * go_ * FastAct<HARD>::tanh_prime(ct) + dcs[q];
        vI[q] = static_cast<T>(dc * gg * FastAct<HARD>::sigm_prime(gi));
        vF[q] = static_cast<T>(dc * cp * FastAct<HARD>::sigm_prime(gf));
        vG[q] = static_cast<T>(dc * gi * FastAct<HARD>::tanh_prime(gg));
        vO[q] = static_cast<T>(dy * ct * FastAct<HARD>::sigm_prime(go_));
        dcs[q] = dc * gf;
        bsum[q][0] += static_cast<float>(vI[q]); bsum[q][1] += static_cast<float>(vF[q]);
        bsum[q][2] += static_cast<float>(vG[q]); bsum[q][3] += static_cast<float>(vO[q]);
      }
      frag o0, o1;   // [unit][gate] interleaved: units u, u+1 | u+2, u+3
      o0[0] = vI[0]; o0[1] = vF[0]; o0[2] = vG[0]; o0[3] = vO[0]; o0[4] = vI[1]; o0[5] = vF[1]; o0[6] = vG[1]; o0[7] = vO[1];
      o1[0] = vI[2]; o1[1] = vF[2]; o1[2] = vG[2]; o1[3] = vO[2]; o1[4] = vI[3]; o1[5] = vF[3]; o1[6] = vG[3]; o1[7] = vO[3];
      const __amdgpu_buffer_rsrc_t ro = res_rsrc(dG);
      res_store16(o0, ro, (int)(eoff * 4) * (int)sizeof(T));
      res_store16(o1, ro, (int)(eoff * 4 + 8) * (int)sizeof(T));
      if (s == nsteps - 1) {   // leave the ring and dC as the step kernels expect them
        const int64_t dsz = (int64_t)((B + 31) / 32 * 32) * 4 * H;
        T* dG_out = w.dring[slot] + ((w.parity[slot] + s) & 1) * dsz;
        *reinterpret_cast<frag*>(dG_out + tiled_index(eb, u * 4, 4 * NKS)) = o0;
        *reinterpret_cast<frag*>(dG_out + tiled_index(eb, u * 4 + 8, 4 * NKS)) = o1;
      }
      f32x4 dv;
#pragma unroll
      for (int q = 0; q < 4; ++q) dv[q] = dcs[q];
      *reinterpret_cast<f32x4*>(w.dC[slot] + eoff) = dv;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (tid == 0 && s + 1 < nsteps) __hip_atomic_fetch_add(qc_mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    have = ready;
    bt = bt_n;
    s = s_n;
  };
  const int total = nsteps * ntiles;
  for (int i = 0; i < total && !aborted; i += 2) {
    body(bufE, bufO);
    if (i + 1 < total && !aborted) body(bufO, bufE);
  }
  if (w.dbias[slot] && !flag[0]) {   // as in lstm_bwd_resident: rows wave * 8 + lane / 8 hold the same units
    float* red = ownp;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = bsum[q][e];
        v += __shfl_xor(v, 8, kWave);
        v += __shfl_xor(v, 16, kWave);
        v += __shfl_xor(v, 32, kWave);
        bsum[q][e] = v;
      }
    __syncthreads();
    if (lane < 8) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(wave * 8 + lane) * 16 + q * 4 + e] = bsum[q][e];
    }
    __syncthreads();
    if (tid < 128) {
      const int cg = tid >> 4, el = tid & 15;
      const float v = red[(0 * 8 + cg) * 16 + el] + red[(1 * 8 + cg) * 16 + el] + red[(2 * 8 + cg) * 16 + el] +
                      red[(3 * 8 + cg) * 16 + el];
      w.dbias[slot][(int64_t)(bx * 32) * 4 + tid] += v;
    }
  }
}




template <typename T>
constexpr bool kHasMfma = std::is_same<T, bf16_t>::value || std::is_same<T, f16_t>::value;

inline int64_t pad32(int64_t b) { return (b + 31) / 32 * 32; }

// operand fragments kept in flight per wave: the whole per-wave K range when it is small (base: 8),
// otherwise the largest batch that divides it evenly (large config, H = 1536: 12 k-steps -> 2 x 6)
inline int pick_batch(int nkw) {
  if (nkw <= 8) return (nkw == 1 || nkw == 2 || nkw == 4 || nkw == 6 || nkw == 8) ? nkw : (nkw == 3 ? 4 : 8);
  for (int b : {8, 6, 4}) if (nkw % b == 0) return b;
  return 8;  // uneven tail: the clamp-and-zero path covers it
}

template <typename T, bool HARD, bool IL>
int launch_fwd_waves(const FwdSlots<T>& w, int n_slots, int n_launches, int64_t B, int64_t H, hipStream_t s) {
  const int nkw = (int)(((H >> 5) + 3) / 4);  // k-steps per wave
  const dim3 grid((unsigned)(H / 4), (unsigned)((B + 31) / 32), (unsigned)n_slots);
  for (int i = 0; i < n_launches; ++i) {
#define CAIMAN_FWD(NKV) hipLaunchKernelGGL((lstm_fwd_step_mfma<T, HARD, NKV, IL>), grid, dim3(256), 0, s, w, i, (int)B, (int)H)
    switch (pick_batch(nkw)) {
      case 1: CAIMAN_FWD(1); break;
      case 2: CAIMAN_FWD(2); break;
      case 4: CAIMAN_FWD(4); break;
      case 6: CAIMAN_FWD(6); break;
      case 8: CAIMAN_FWD(8); break;
      default: CAIMAN_FWD(0); break;
    }
#undef CAIMAN_FWD
  }
  return check_launch("lstm forward wave");
}

// dbias[c] += sum over the call's n*B dG rows (per-timestep path: the step kernels do not carry the sums).  The rows
// of a backward call run DOWN from dG_hi.  A workgroup owns 64 columns; its four waves take every fourth row (eight
// loads in flight each) and meet in LDS, so the result has a fixed summation order.
template <typename T>
__global__ __launch_bounds__(256) void dbias_rows_kernel(const T* __restrict__ dG_hi, int64_t row_elems, int n_rows_t, int B,
                                                        int cols, float* __restrict__ dbias) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int total = n_rows_t * B;
  float s = 0.f;
  if (c < cols) {
    int r = grp;
    for (; r + 28 < total; r += 32) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = r + 4 * i, t = rr / B, b = rr - t * B;
        v[i] = static_cast<float>(dG_hi[-(int64_t)t * row_elems + (int64_t)b * cols + c]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; r < total; r += 4) {
      const int t = r / B, b = r - t * B;
      s += static_cast<float>(dG_hi[-(int64_t)t * row_elems + (int64_t)b * cols + c]);
    }
  }
  part[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && c < cols) dbias[c] += part[0][cl] + part[1][cl] + part[2][cl] + part[3][cl];
}

template <typename T, bool HARD, bool IL>
int launch_bwd_waves(const BwdSlots<T>& w, int n_slots, int n_launches, int64_t B, int64_t H, hipStream_t s) {
  const int nkw = (int)(((4 * H >> 5) + 15) / 16);
  const dim3 grid((unsigned)(H / 16), (unsigned)((B + 31) / 32), (unsigned)n_slots);
  for (int i = 0; i < n_launches; ++i) {
#define CAIMAN_BWD(NKV) hipLaunchKernelGGL((lstm_bwd_step_mfma<T, HARD, NKV, IL>), grid, dim3(1024), 0, s, w, i, (int)B, (int)H)
    switch (pick_batch(nkw)) {
      case 1: CAIMAN_BWD(1); break;
      case 2: CAIMAN_BWD(2); break;
      case 4: CAIMAN_BWD(4); break;
      case 6: CAIMAN_BWD(6); break;
      case 8: CAIMAN_BWD(8); break;
      default: CAIMAN_BWD(0); break;
    }
#undef CAIMAN_BWD
  }
  for (int i = 0; i < n_slots; ++i) {
    if (!w.dbias[i] || w.nsteps[i] <= 0) continue;
    const int Hs = w.hidden[i] ? w.hidden[i] : (int)H;
    hipLaunchKernelGGL((dbias_rows_kernel<T>), dim3((unsigned)((4 * Hs + 63) / 64)), dim3(256), 0, s, w.dG[i],
                       (int64_t)B * 4 * Hs, w.nsteps[i], (int)B, 4 * Hs, w.dbias[i]);
  }
  return check_launch("lstm backward wave");
}

// ---- resident launch state: per-device pool of zeroed counter blocks + a host-visible failure word ----
constexpr int kResPool = 32;
struct ResState {
  unsigned* sync[kResPool] = {};
  unsigned* fail_host = nullptr;
  float* partials = nullptr;   // 2-D split backward kernel: K-quarter partial sums in flight between workgroups
  float* partials_bt = nullptr;         // batch-tile kernels (B > 32): allocated on first use
  unsigned* sync_bt[kResPool] = {};
  int next_bt = 0;
  int next = 0;
  int cus = 0;
  int dev = 0;
  bool ok = false;
  // Two resident grids on different streams could each hold part of the chip and wait for the rest: a resident
  // launch on another stream than the previous one first waits for that one's completion event.
  hipEvent_t done = nullptr;
  hipStream_t last_stream = nullptr;
  bool has_last = false;
};
std::mutex g_res_mu;
ResState g_res[16];
std::atomic<int> g_res_mode{1};
std::atomic<int> g_res_xcd_roles{0};   // flat grid with slot = block % 8 (res_role): a slot's workgroups share an XCD (measured: no gain)
inline dim3 res_grid(int nks, int n_slots, int cus) {
  if (g_res_xcd_roles.load(std::memory_order_relaxed) && nks * 8 <= cus) return dim3((unsigned)(nks * 8));
  return dim3((unsigned)nks, (unsigned)n_slots);
}
std::atomic<int> g_res_bwd_split{1};   // 2-D split backward kernel where the shape allows it (H = 512, 1024)
std::atomic<long long> g_res_launches{0};

// counter block for one launch; orders the launch behind a resident launch still running on another stream
unsigned* res_begin(ResState* st, hipStream_t s, unsigned** scrub) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  unsigned* sync = st->sync[st->next];
  *scrub = st->sync[(st->next + kResPool / 2) % kResPool];   // res_scrub: cleared by this launch for the one half a pool later
  st->next = (st->next + 1) % kResPool;
  // measurement aid: the round-1 behaviour (a fill command in front of every launch) for an A/B on one box
  static const bool memset_too = std::getenv("CAIMAN_LSTM_RESIDENT_MEMSET") != nullptr;
  if (memset_too) (void)hipMemsetAsync(sync, 0, kResSyncBytes, s);
  if (st->has_last && st->last_stream != s) (void)hipStreamWaitEvent(s, st->done, 0);
  return sync;
}
void res_end(ResState* st, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  (void)hipEventRecord(st->done, s);
  st->last_stream = s;
  st->has_last = true;
  g_res_launches.fetch_add(1, std::memory_order_relaxed);
}

ResState* res_state() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(g_res_mu);
  ResState& st = g_res[dev];
  if (!st.ok) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return nullptr;
    st.cus = prop.multiProcessorCount;
    st.dev = dev;
    if (hipHostMalloc(reinterpret_cast<void**>(&st.fail_host), 128, hipHostMallocMapped) != hipSuccess) return nullptr;
    for (int i = 0; i < 32; ++i) st.fail_host[i] = 0;
    if (hipMalloc(reinterpret_cast<void**>(&st.partials), kMaxSlots * kRes2PartialFloatsPerSlot * sizeof(float)) != hipSuccess)
      return nullptr;
    for (int i = 0; i < kResPool; ++i)
      if (hipMalloc(reinterpret_cast<void**>(&st.sync[i]), kResSyncBytes) != hipSuccess ||
          hipMemset(st.sync[i], 0, kResSyncBytes) != hipSuccess)
        return nullptr;
    if (hipEventCreateWithFlags(&st.done, hipEventDisableTiming) != hipSuccess) return nullptr;
    st.ok = true;
  }
  return &st;
}

template <typename T>
inline size_t res_fwd_lds(int H) { return (size_t)(32 * (H + 8) + 4 * 2 * 32 * 8) * sizeof(T) + 16; }

// A tick's slots are independent recurrences (each layer works on its own chunk): when they do not fit the chip together
// (H = 1536: 48 workgroups per layer, five layers per launch) they go out as consecutive resident launches.
template <typename S>
inline S res_slot_group(const S& w, int first, int count) {
  S g{};
  g.seed = w.seed;
  g.batch_stride = w.batch_stride;
  auto cp = [&](auto S::*field) {
    for (int i = 0; i < count; ++i) (g.*field)[i] = (w.*field)[first + i];
  };
  if constexpr (std::is_same<S, FwdSlots<bf16_t>>::value || std::is_same<S, FwdSlots<f16_t>>::value) {
    cp(&S::Rtile); cp(&S::g); cp(&S::c); cp(&S::y); cp(&S::hring); cp(&S::parity); cp(&S::nsteps); cp(&S::ymask);
    cp(&S::drop_base); cp(&S::drop_p); cp(&S::hidden);
  } else {
    cp(&S::Rttile); cp(&S::g); cp(&S::c); cp(&S::delta); cp(&S::d_st); cp(&S::d_sb); cp(&S::dG); cp(&S::dring); cp(&S::dC);
    cp(&S::parity); cp(&S::nsteps); cp(&S::has_in0); cp(&S::drop_base); cp(&S::drop_p); cp(&S::hidden); cp(&S::dbias);
  }
  return g;
}

// the DMA-gather forward kernel: the only one for H = 1536, and the default for H = 512 / 1024 as well (forward recurrence
// of the base encoder 3.65 -> 3.54 ms per training step in an A/B on one box; CAIMAN_LSTM_FWD_DMA=0 restores the
// register-staged gather of lstm_fwd_resident)
inline bool res_fwd_use_dma(int nks) {
  static const bool off = std::getenv("CAIMAN_LSTM_FWD_DMA") != nullptr && std::atoi(std::getenv("CAIMAN_LSTM_FWD_DMA")) == 0;
  return nks == 48 || (!off && (nks == 16 || nks == 32));
}
// LDS reads issued ahead of the MFMAs in the H = 1536 kernels, in k-steps (CAIMAN_LSTM_WIDE_KB=1|2 forces both).  Measured
// on large-196M, B = 32: forward 5.50 (1) / 5.39 ms (2: 5 registers spill, still faster), backward 6.73 (1) / 6.93 ms (2).
inline int res_wide_kb(bool backward) {
  static const int forced = std::getenv("CAIMAN_LSTM_WIDE_KB") ? std::atoi(std::getenv("CAIMAN_LSTM_WIDE_KB")) : 0;
  if (forced == 1 || forced == 2) return forced;
  return backward ? 1 : 2;
}

// Batches the batch-tile kernels do not take (B > 32 with H = 768 or 1536, or more tiles than they hold): the B <= 32
// kernels run once per slice of 32 batch rows.  A slice is an independent recurrence, so its launch gets the slot
// pointers moved to the slice's first row and `batch_stride` = the whole batch; the ring halves are tiled by 32 rows
// (tiled_index), so slice i is tile i of either half.  The weights are loaded once per slice instead of once per call
// (H = 1536: 18.9 MB per layer from L2/HBM, ~3 % of a 24-step call).
template <typename T>
inline FwdSlots<T> res_batch_slice(const FwdSlots<T>& w, int n_slots, int64_t B, int64_t H, int64_t b0) {
  FwdSlots<T> g = w;
  g.batch_stride = (int)B;
  for (int i = 0; i < n_slots; ++i) {
    g.g[i] += b0 * 4 * H; g.c[i] += b0 * H; g.y[i] += b0 * H; g.hring[i] += b0 * H;
    if (g.ymask[i]) g.ymask[i] += b0 * H;
    g.drop_base[i] += (uint64_t)(b0 * H);
  }
  return g;
}
template <typename T>
inline BwdSlots<T> res_batch_slice(const BwdSlots<T>& w, int n_slots, int64_t B, int64_t H, int64_t b0) {
  BwdSlots<T> g = w;
  g.batch_stride = (int)B;
  for (int i = 0; i < n_slots; ++i) {
    g.g[i] += b0 * 4 * H; g.c[i] += b0 * H; g.delta[i] += b0 * g.d_sb[i]; g.dG[i] += b0 * 4 * H;
    g.dring[i] += b0 * 4 * H; g.dC[i] += b0 * H;
    g.drop_base[i] += (uint64_t)(b0 * H);
  }
  return g;
}
// shapes served slice by slice: what the B <= 32 kernels take and the batch-tile kernels (H = 512, 1024 up to
// 32 * kResMaxTiles rows) do not
inline bool res_sliced_shape(int64_t B, int64_t H) {
  static const bool off = std::getenv("CAIMAN_LSTM_BATCH_SLICES") != nullptr && std::atoi(std::getenv("CAIMAN_LSTM_BATCH_SLICES")) == 0;
  if (off || B <= 32) return false;
  if ((H == 512 || H == 1024) && B <= 32 * kResMaxTiles) return false;
  return true;
}

// true when the launch was taken by the resident kernel
template <typename T, bool HARD>
bool try_fwd_resident(const FwdSlots<T>& w_all, int n_slots_all, int n_launches, int64_t B, int64_t H, hipStream_t s, int* err) {
  *err = CAIMAN_OK;
  if (!g_res_mode.load(std::memory_order_relaxed) || B > 32 || n_launches < 2) return false;
  for (int i = 0; i < n_slots_all; ++i)
    if ((w_all.hidden[i] ? w_all.hidden[i] : (int)H) != (int)H) return false;   // one width per launch
  const int nks = (int)(H / 32);
  if (!(nks == 2 || nks == 4 || nks == 8 || nks == 16 || nks == 24 || nks == 32 || nks == 48)) return false;
  ResState* st = res_state();
  if (!st || nks > st->cus) return false;
  if (*reinterpret_cast<volatile unsigned*>(st->fail_host) != 0u) return false;   // a hand-off has timed out in this process: stay on the per-timestep kernels
  const int per_launch = st->cus / nks;
  const bool dma = res_fwd_use_dma(nks);
  const int mode = g_res_mode.load(std::memory_order_relaxed);
  for (int first = 0; first < n_slots_all; first += per_launch) {
    const int n_slots = std::min(per_launch, n_slots_all - first);
    const FwdSlots<T> w = (first == 0 && n_slots == n_slots_all) ? w_all : res_slot_group(w_all, first, n_slots);
    unsigned* scrub = nullptr;
    unsigned* sync = res_begin(st, s, &scrub);
    const dim3 grid = res_grid(nks, n_slots, st->cus);
    if (dma) {
      const size_t lds = (size_t)(4 * 2 * 32 * 8) * sizeof(T) + 16;   // the row buffers are static LDS objects of the kernel
#define CAIMAN_RESD(NKV, KBV)                                                                                        \
  do {                                                                                                               \
    if (mode == 2) hipLaunchKernelGGL((lstm_fwd_resident_dma<T, HARD, NKV, true, KBV>), grid, dim3(256), lds, s, w, (int)B, \
                                      sync, st->fail_host, scrub);                                                   \
    else hipLaunchKernelGGL((lstm_fwd_resident_dma<T, HARD, NKV, false, KBV>), grid, dim3(256), lds, s, w, (int)B, sync,    \
                            st->fail_host, scrub);                                                                   \
  } while (0)
      if (nks == 16) CAIMAN_RESD(16, 1);
      else if (nks == 32) CAIMAN_RESD(32, 1);
      else if (res_wide_kb(false) == 2) CAIMAN_RESD(48, 2);
      else CAIMAN_RESD(48, 1);
#undef CAIMAN_RESD
    } else {
      const size_t lds = res_fwd_lds<T>((int)H);
#define CAIMAN_RES(NKV)                                                                                              \
  do {                                                                                                               \
    auto kern = mode == 2 ? lstm_fwd_resident<T, HARD, NKV, true> : lstm_fwd_resident<T, HARD, NKV, false>;           \
    static bool attr_set[2][16] = {};   /* per device: the attribute belongs to the device's code object */         \
    bool& attr_done = attr_set[mode == 2][st->dev];                                                                  \
    if (!attr_done) {                                                                                                 \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                              (int)res_fwd_lds<T>(NKV * 32)) != hipSuccess) {                                        \
        *err = check_launch("lstm resident attribute");                                                              \
        return true;                                                                                                 \
      }                                                                                                              \
      attr_done = true;                                                                                              \
    }                                                                                                                \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, w, (int)B, sync, st->fail_host, scrub);                                \
  } while (0)
      switch (nks) {
        case 2: CAIMAN_RES(2); break;
        case 4: CAIMAN_RES(4); break;
        case 8: CAIMAN_RES(8); break;
        case 16: CAIMAN_RES(16); break;
        case 24: CAIMAN_RES(24); break;
        default: CAIMAN_RES(32); break;
      }
#undef CAIMAN_RES
    }
    res_end(st, s);
    *err = check_launch("lstm resident forward");
    if (*err != CAIMAN_OK) return true;
  }
  return true;
}

// B > 32 outside the batch-tile kernels' shapes: one B <= 32 launch (group) per 32-row slice, in stream order
template <typename T, bool HARD>
bool try_fwd_resident_slices(const FwdSlots<T>& w, int n_slots, int n_launches, int64_t B, int64_t H, hipStream_t s, int* err) {
  *err = CAIMAN_OK;
  if (!res_sliced_shape(B, H)) return false;
  for (int64_t b0 = 0; b0 < B; b0 += 32) {
    const FwdSlots<T> ws = res_batch_slice(w, n_slots, B, H, b0);
    if (!try_fwd_resident<T, HARD>(ws, n_slots, n_launches, std::min<int64_t>(32, B - b0), H, s, err)) {
      if (b0 == 0) return false;   // not admitted: nothing has been launched
      // only a hand-off failure between two slices ends up here; the earlier slices have overwritten their gates
      set_error("lstm resident forward: slice at row %lld refused after earlier slices ran", (long long)b0);
      *err = CAIMAN_ERR_LAUNCH;
      return true;
    }
    if (*err != CAIMAN_OK) return true;
  }
  return true;
}

template <typename T>
inline size_t res_bwd_lds(int H) {   // BwdResGeom
  const int nks = H / 32;
  const bool dma = nks >= 8;
  const int nst = dma ? nks / 4 : 4, nb = dma ? (nst < 4 ? nst : 4) : 2;
  // dynamic part only: with DMA the ring buffers are static LDS objects of the kernel
  return (dma ? 0 : (size_t)(nb * 32 * (4 * H / nst + 8)) * sizeof(T)) + (size_t)8 * 16 * 17 * sizeof(float) + 16;
}

template <typename T, bool HARD>
bool try_bwd_resident(const BwdSlots<T>& w, int n_slots, int n_launches, int64_t B, int64_t H, hipStream_t s, int* err) {
  *err = CAIMAN_OK;
  if (!g_res_mode.load(std::memory_order_relaxed) || B > 32 || n_launches < 2) return false;
  for (int i = 0; i < n_slots; ++i) {
    if ((w.hidden[i] ? w.hidden[i] : (int)H) != (int)H) return false;
    // the epilogue moves delta in 8-byte and dC in 16-byte pieces
    if ((reinterpret_cast<uintptr_t>(w.delta[i]) & 7u) || (w.d_sb[i] & 3) || (w.d_st[i] & 3) ||
        (reinterpret_cast<uintptr_t>(w.dC[i]) & 15u))
      return false;
  }
  const int nks = (int)(H / 32);
  if (!(nks == 2 || nks == 4 || nks == 8 || nks == 16 || nks == 24 || nks == 32)) return false;
  ResState* st = res_state();
  if (!st || (int64_t)n_slots * nks > st->cus) return false;
  if (*reinterpret_cast<volatile unsigned*>(st->fail_host) != 0u) return false;   // a hand-off has timed out in this process: stay on the per-timestep kernels
  unsigned* scrub = nullptr;
  unsigned* sync = res_begin(st, s, &scrub);
  const dim3 grid = res_grid(nks, n_slots, st->cus);
  const size_t lds = res_bwd_lds<T>((int)H);
#define CAIMAN_RES(NKV)                                                                                              \
  do {                                                                                                               \
    auto kern = g_res_mode.load(std::memory_order_relaxed) == 2 ? lstm_bwd_resident<T, HARD, NKV, true>               \
                                                                : lstm_bwd_resident<T, HARD, NKV, false>;             \
    static bool attr_set[2][16] = {};   /* per device: the attribute belongs to the device's code object */         \
    bool& attr_done = attr_set[g_res_mode.load(std::memory_order_relaxed) == 2][st->dev];                            \
    if (!attr_done) {                                                                                                 \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                              (int)res_bwd_lds<T>(NKV * 32)) != hipSuccess) {                                        \
        *err = check_launch("lstm resident attribute");                                                              \
        return true;                                                                                                 \
      }                                                                                                              \
      attr_done = true;                                                                                              \
    }                                                                                                                \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, w, (int)B, sync, st->fail_host, scrub);                                \
  } while (0)
  switch (nks) {
    case 2: CAIMAN_RES(2); break;
    case 4: CAIMAN_RES(4); break;
    case 8: CAIMAN_RES(8); break;
    case 16: CAIMAN_RES(16); break;
    case 24: CAIMAN_RES(24); break;
    default: CAIMAN_RES(32); break;
  }
#undef CAIMAN_RES
  res_end(st, s);
  *err = check_launch("lstm resident backward");
  return true;
}

// 2-D split variant (lstm_bwd_resident2): same admission rules, H = 512, 1024 or 1536 (the latter in slot groups that
// fit the chip, as in try_fwd_resident)
template <typename T, bool HARD>
bool try_bwd_resident2(const BwdSlots<T>& w_all, int n_slots_all, int n_launches, int64_t B, int64_t H, hipStream_t s, int* err) {
  *err = CAIMAN_OK;
  if (!g_res_mode.load(std::memory_order_relaxed) || B > 32 || n_launches < 2) return false;
  if (H != 1536 && !g_res_bwd_split.load(std::memory_order_relaxed)) return false;   // H = 1536 has no whole-row kernel
  if (H != 512 && H != 1024 && H != 1536) return false;
  for (int i = 0; i < n_slots_all; ++i) {
    if ((w_all.hidden[i] ? w_all.hidden[i] : (int)H) != (int)H) return false;
    if ((reinterpret_cast<uintptr_t>(w_all.delta[i]) & 7u) || (w_all.d_sb[i] & 3) || (w_all.d_st[i] & 3) ||
        (reinterpret_cast<uintptr_t>(w_all.dC[i]) & 15u))
      return false;
  }
  const int nks = (int)(H / 32);
  ResState* st = res_state();
  if (!st || nks > st->cus) return false;
  if (*reinterpret_cast<volatile unsigned*>(st->fail_host) != 0u) return false;
  const int per_launch = st->cus / nks;
  const bool prof = g_res_mode.load(std::memory_order_relaxed) == 2;
  for (int first = 0; first < n_slots_all; first += per_launch) {
    const int n_slots = std::min(per_launch, n_slots_all - first);
    const BwdSlots<T> w = (first == 0 && n_slots == n_slots_all) ? w_all : res_slot_group(w_all, first, n_slots);
    unsigned* scrub = nullptr;
    unsigned* sync = res_begin(st, s, &scrub);
    const dim3 grid = res_grid(nks, n_slots, st->cus);
    const size_t lds = (size_t)(32 * 36) * sizeof(float) + 16 + (nks > 32 ? (size_t)16 * 256 * sizeof(float) : 0);
#define CAIMAN_RES2(NKV, KBV)                                                                                        \
  do {                                                                                                               \
    if (prof) hipLaunchKernelGGL((lstm_bwd_resident2<T, HARD, NKV, true, KBV>), grid, dim3(256), lds, s, w, (int)B, sync, \
                                 st->fail_host, st->partials, scrub);                                                \
    else hipLaunchKernelGGL((lstm_bwd_resident2<T, HARD, NKV, false, KBV>), grid, dim3(256), lds, s, w, (int)B, sync,    \
                            st->fail_host, st->partials, scrub);                                                     \
  } while (0)
    if (nks == 16) CAIMAN_RES2(16, 1);
    else if (nks == 32) CAIMAN_RES2(32, 1);
    else if (res_wide_kb(true) == 2) CAIMAN_RES2(48, 2);
    else CAIMAN_RES2(48, 1);
#undef CAIMAN_RES2
    res_end(st, s);
    *err = check_launch("lstm resident backward (2-D split)");
    if (*err != CAIMAN_OK) return true;
  }
  return true;
}

template <typename T, bool HARD>
bool try_bwd_resident_slices(const BwdSlots<T>& w, int n_slots, int n_launches, int64_t B, int64_t H, hipStream_t s, int* err) {
  *err = CAIMAN_OK;
  if (!res_sliced_shape(B, H)) return false;
  for (int64_t b0 = 0; b0 < B; b0 += 32) {
    const BwdSlots<T> ws = res_batch_slice(w, n_slots, B, H, b0);
    const int64_t rows = std::min<int64_t>(32, B - b0);
    if (!try_bwd_resident2<T, HARD>(ws, n_slots, n_launches, rows, H, s, err) &&
        !try_bwd_resident<T, HARD>(ws, n_slots, n_launches, rows, H, s, err)) {
      if (b0 == 0) return false;
      set_error("lstm resident backward: slice at row %lld refused after earlier slices ran", (long long)b0);
      *err = CAIMAN_ERR_LAUNCH;
      return true;
    }
    if (*err != CAIMAN_OK) return true;
  }
  return true;
}

// ---- batch-tile launches (32 < B <= 128): lstm_fwd_resident_bt / lstm_bwd_resident2_bt ----------------------------
bool res_bt_ready(ResState* st) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  if (st->partials_bt) return true;
  for (int i = 0; i < kResPool; ++i)
    if (hipMalloc(reinterpret_cast<void**>(&st->sync_bt[i]), kResSyncBytesBT) != hipSuccess ||
        hipMemset(st->sync_bt[i], 0, kResSyncBytesBT) != hipSuccess)
      return false;
  return hipMalloc(reinterpret_cast<void**>(&st->partials_bt),
                   (size_t)kMaxSlots * kResMaxTiles * kRes2PartialFloatsPerSlot * sizeof(float)) == hipSuccess;
}
unsigned* res_begin_bt(ResState* st, hipStream_t s, unsigned** scrub) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  unsigned* sync = st->sync_bt[st->next_bt];
  *scrub = st->sync_bt[(st->next_bt + kResPool / 2) % kResPool];
  st->next_bt = (st->next_bt + 1) % kResPool;
  if (st->has_last && st->last_stream != s) (void)hipStreamWaitEvent(s, st->done, 0);
  return sync;
}

std::atomic<int> g_res_bt_dma{1};   // batch-tile forward kernel with double-buffered DMA operands (lstm_fwd_resident_bt_dma)

template <typename T, bool HARD>
bool try_fwd_resident_bt(const FwdSlots<T>& w, int n_slots, int n_launches, int64_t B, int64_t H, hipStream_t s, int* err) {
  *err = CAIMAN_OK;
  if (!g_res_mode.load(std::memory_order_relaxed) || B <= 32 || B > 32 * kResMaxTiles || n_launches < 2) return false;
  if (H != 256 && H != 512 && H != 1024) return false;
  for (int i = 0; i < n_slots; ++i)
    if ((w.hidden[i] ? w.hidden[i] : (int)H) != (int)H) return false;
  const int nks = (int)(H / 32);
  ResState* st = res_state();
  if (!st || (int64_t)n_slots * nks > st->cus || !res_bt_ready(st)) return false;
  if (*reinterpret_cast<volatile unsigned*>(st->fail_host) != 0u) return false;
  unsigned* scrub = nullptr;
  unsigned* sync = res_begin_bt(st, s, &scrub);
  const dim3 grid = res_grid(nks, n_slots, st->cus);
  if (g_res_bt_dma.load(std::memory_order_relaxed) && (nks == 16 || nks == 32)) {
    const size_t lds = (size_t)(4 * 2 * 32 * 8) * sizeof(T) + 16;   // the operand buffers are static LDS objects of the kernel
    if (nks == 16) hipLaunchKernelGGL((lstm_fwd_resident_bt_dma<T, HARD, 16>), grid, dim3(256), lds, s, w, (int)B, sync, st->fail_host, scrub);
    else hipLaunchKernelGGL((lstm_fwd_resident_bt_dma<T, HARD, 32>), grid, dim3(256), lds, s, w, (int)B, sync, st->fail_host, scrub);
    res_end(st, s);
    *err = check_launch("lstm resident forward (batch tiles, double-buffered)");
    return true;
  }
  const size_t lds = res_fwd_lds<T>((int)H);
#define CAIMAN_RESBT(NKV)                                                                                            \
  do {                                                                                                               \
    auto kern = lstm_fwd_resident_bt<T, HARD, NKV>;                                                                  \
    static bool attr_set[16] = {};                                                                                   \
    if (!attr_set[st->dev]) {                                                                                        \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                              (int)res_fwd_lds<T>(NKV * 32)) != hipSuccess) {                                        \
        *err = check_launch("lstm resident attribute");                                                              \
        return true;                                                                                                 \
      }                                                                                                              \
      attr_set[st->dev] = true;                                                                                      \
    }                                                                                                                \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, w, (int)B, sync, st->fail_host, scrub);                                \
  } while (0)
  if (nks == 8) CAIMAN_RESBT(8); else if (nks == 16) CAIMAN_RESBT(16); else CAIMAN_RESBT(32);
#undef CAIMAN_RESBT
  res_end(st, s);
  *err = check_launch("lstm resident forward (batch tiles)");
  return true;
}

template <typename T, bool HARD>
bool try_bwd_resident2_bt(const BwdSlots<T>& w, int n_slots, int n_launches, int64_t B, int64_t H, hipStream_t s, int* err) {
  *err = CAIMAN_OK;
  if (!g_res_mode.load(std::memory_order_relaxed) || B <= 32 || B > 32 * kResMaxTiles || n_launches < 2) return false;
  if (H != 512 && H != 1024) return false;
  for (int i = 0; i < n_slots; ++i) {
    if ((w.hidden[i] ? w.hidden[i] : (int)H) != (int)H) return false;
    if ((reinterpret_cast<uintptr_t>(w.delta[i]) & 7u) || (w.d_sb[i] & 3) || (w.d_st[i] & 3) ||
        (reinterpret_cast<uintptr_t>(w.dC[i]) & 15u))
      return false;
  }
  const int nks = (int)(H / 32);
  ResState* st = res_state();
  if (!st || (int64_t)n_slots * nks > st->cus || !res_bt_ready(st)) return false;
  if (*reinterpret_cast<volatile unsigned*>(st->fail_host) != 0u) return false;
  unsigned* scrub = nullptr;
  unsigned* sync = res_begin_bt(st, s, &scrub);
  const dim3 grid = res_grid(nks, n_slots, st->cus);
  const size_t lds = (size_t)(32 * 36) * sizeof(float) + 16;
  const bool dma = g_res_bt_dma.load(std::memory_order_relaxed) != 0;   // double-buffered gather (default) | round-2 kernel
  if (nks == 16) {
    if (dma) hipLaunchKernelGGL((lstm_bwd_resident2_bt_dma<T, HARD, 16>), grid, dim3(256), lds, s, w, (int)B, sync, st->fail_host, st->partials_bt, scrub);
    else hipLaunchKernelGGL((lstm_bwd_resident2_bt<T, HARD, 16>), grid, dim3(256), lds, s, w, (int)B, sync, st->fail_host, st->partials_bt, scrub);
  } else {
    if (dma) hipLaunchKernelGGL((lstm_bwd_resident2_bt_dma<T, HARD, 32>), grid, dim3(256), lds, s, w, (int)B, sync, st->fail_host, st->partials_bt, scrub);
    else hipLaunchKernelGGL((lstm_bwd_resident2_bt<T, HARD, 32>), grid, dim3(256), lds, s, w, (int)B, sync, st->fail_host, st->partials_bt, scrub);
  }
  res_end(st, s);
  *err = check_launch("lstm resident backward (2-D split, batch tiles)");
  return true;
}

template <typename T>
int prepare_fwd(const T* R, const T* h0, T* Rtile, T* hring, int64_t B, int64_t H, hipStream_t s, bool zeroed = false) {
  if (!zeroed && hipMemsetAsync(hring, 0, sizeof(T) * (size_t)(2 * pad32(B) * H), s) != hipSuccess) return check_launch("lstm prepare memset");
  if (R)   // R == nullptr: Rtile already holds the image (caiman_lstm_weight_images)
    hipLaunchKernelGGL((tile_R_fwd_kernel<T>), dim3((unsigned)((4 * H * H + 255) / 256)), dim3(256), 0, s, R, Rtile, (int)H);
  hipLaunchKernelGGL((tile_rows_kernel<T>), dim3((unsigned)((B * H + 255) / 256)), dim3(256), 0, s, h0, hring, (int)B, (int)H);
  return check_launch("lstm prepare forward");
}

template <typename T>
int prepare_bwd(const T* R, T* Rttile, T* dring, float* dC, int64_t B, int64_t H, hipStream_t s, bool il = false, bool zeroed = false) {
  if (!zeroed) {
    if (hipMemsetAsync(dring, 0, sizeof(T) * (size_t)(2 * pad32(B) * 4 * H), s) != hipSuccess) return check_launch("lstm prepare memset");
    if (hipMemsetAsync(dC, 0, sizeof(float) * (size_t)(B * H), s) != hipSuccess) return check_launch("lstm prepare memset");
  }
  if (!R) return check_launch("lstm prepare backward");   // the image is already there (caiman_lstm_weight_images)
  if (il) hipLaunchKernelGGL((tile_Rt_bwd_kernel<T, true>), dim3((unsigned)((H / 16) * (4 * H / 32))), dim3(256), 0, s, R, Rttile, (int)H);
  else hipLaunchKernelGGL((tile_Rt_bwd_kernel<T, false>), dim3((unsigned)((H / 16) * (4 * H / 32))), dim3(256), 0, s, R, Rttile, (int)H);
  return check_launch("lstm prepare backward");
}

template <typename T, bool HARD>
int run_fwd(const T* R, T* gates, T* c, T* y, T* work, int64_t Tn, int64_t B, int64_t H, hipStream_t s) {
  const int64_t go = B * 4 * H, so = B * H;
  if constexpr (kHasMfma<T>) {
    if ((H % 32 == 0) && work != nullptr) {
      T* Rtile = work;
      T* hring = work + 4 * H * H;
      if (int e = prepare_fwd<T>(R, y, Rtile, hring, B, H, s)) return e;
      FwdSlots<T> w{};
      w.Rtile[0] = Rtile; w.g[0] = gates; w.c[0] = c; w.y[0] = y; w.hring[0] = hring; w.parity[0] = 0;
      // the launch index is an int: feed the sequence in pieces so `nsteps` stays small
      for (int64_t t0 = 0; t0 < Tn; t0 += 4096) {
        const int n = (int)std::min<int64_t>(4096, Tn - t0);
        w.g[0] = gates + go * t0; w.c[0] = c + so * t0; w.y[0] = y + so * t0;
        w.parity[0] = (int)(t0 & 1); w.nsteps[0] = n;
        if (int e = launch_fwd_waves<T, HARD, false>(w, 1, n, B, H, s)) return e;
      }
      return CAIMAN_OK;
    }
  }
  for (int64_t t = 0; t < Tn; ++t)
    hipLaunchKernelGGL((lstm_fwd_step_generic<T, HARD>), dim3((unsigned)((H + 63) / 64), (unsigned)B), dim3(64),
                       0, s, R, gates + go * t, c + so * t, c + so * (t + 1), y + so * t, y + so * (t + 1), (int)B,
                       (int)H);
  return check_launch("caiman_lstm_fused_fwd");
}

template <typename T, bool HARD>
int run_bwd(const T* R, const T* gates, const T* c, const T* delta, int64_t d_st, int64_t d_sb, T* dG,
            acc_t<T>* dC, T* work, int64_t Tn, int64_t B, int64_t H, hipStream_t s) {
  const int64_t go = B * 4 * H, so = B * H;
  if constexpr (kHasMfma<T>) {
    if ((H % 32 == 0) && work != nullptr) {
      T* Rttile = work;
      T* dring = work + 4 * H * H;
      if (int e = prepare_bwd<T>(R, Rttile, dring, dC, B, H, s)) return e;
      BwdSlots<T> w{};
      w.Rttile[0] = Rttile; w.dring[0] = dring; w.dC[0] = dC; w.d_st[0] = d_st; w.d_sb[0] = d_sb;
      for (int64_t thi = Tn - 1; thi >= 0; thi -= 4096) {
        const int n = (int)std::min<int64_t>(4096, thi + 1);
        w.g[0] = gates + go * thi; w.c[0] = c + so * thi; w.delta[0] = delta + d_st * thi; w.dG[0] = dG + go * thi;
        w.parity[0] = (int)(thi & 1); w.nsteps[0] = n; w.has_in0[0] = thi < Tn - 1;
        if (int e = launch_bwd_waves<T, HARD, false>(w, 1, n, B, H, s)) return e;
      }
      return CAIMAN_OK;
    }
  }
  if (hipMemsetAsync(dC, 0, sizeof(acc_t<T>) * (size_t)so, s) != hipSuccess) return check_launch("lstm_bwd memset");
  for (int64_t t = Tn - 1; t >= 0; --t) {
    const T* dgn = (t < Tn - 1) ? dG + go * (t + 1) : nullptr;
    hipLaunchKernelGGL((lstm_bwd_step_generic<T, HARD>), dim3((unsigned)((H + 63) / 64), (unsigned)B), dim3(64),
                       0, s, R, gates + go * t, c + so * t, c + so * (t + 1), delta + d_st * t, d_sb, dgn,
                       dG + go * t, dC, (int)B, (int)H);
  }
  return check_launch("caiman_lstm_fused_bwd");
}

}  // namespace
}  // namespace caiman

namespace caiman {
const unsigned* resident_fail_word() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(g_res_mu);
  return g_res[dev].ok ? g_res[dev].fail_host : nullptr;
}
namespace {
__global__ void resident_poison_kernel(float* grad, const unsigned* fail_word, const unsigned* seen) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && *fail_word > *seen) *grad = __builtin_nanf("");
}
}  // namespace
}  // namespace caiman

// Data-parallel runs: a hand-off timeout on ONE rank must make EVERY rank drop the optimiser step.  Queued in front of
// the all-reduce of the gradient slice that holds `grad_elem`, this writes a NaN there when the failure count has
// moved past `*seen` (the count caiman_lamb_step recorded at its last call: work[5] of its scratch, as uint32); the
// sum carries the NaN to every rank and each rank's caiman_lamb_step skips the update (non-finite gradient norm).
extern "C" int caiman_lstm_resident_poison(float* grad_elem, const uint32_t* seen, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(grad_elem && seen, "lstm_resident_poison: null pointer");
  const unsigned* fw = resident_fail_word();
  if (!fw) return CAIMAN_OK;   // no resident launch was ever attempted on this device
  hipLaunchKernelGGL(resident_poison_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), grad_elem, fw, seen);
  return check_launch("caiman_lstm_resident_poison");
}

namespace caiman { namespace {
// `workgroups` workgroups that each fill a CU's LDS (so that one lands per CU) and stay for `microseconds` of wall-clock time
// (s_memrealtime: 100 MHz), then leave.  Bounded by construction: no workgroup waits for another.
__global__ __launch_bounds__(256) void occupy_cus_kernel(unsigned ticks, unsigned* sink) {
  extern __shared__ unsigned occupy_lds[];
  occupy_lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  unsigned acc = occupy_lds[(threadIdx.x * 7) & 255];
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    acc = acc * 1664525u + 1013904223u;
    __builtin_amdgcn_s_sleep(32);
  }
  if (acc == 0x9e3779b9u && sink) sink[0] = acc;     // keeps the loop alive; practically never true
}
} }  // namespace

// What a collective's kernel does to the chip, without a second rank: `workgroups` CUs are held for `microseconds` on
// `stream` (RCCL's kernels hold a fixed set of CUs for as long as the slowest rank takes).  The single-GPU rehearsal of
// "weight-resident LSTM grids and collectives in one job" (tests/test_gpu_distributed.py) queues it on the reducer's
// communication stream; train_utils/overlap.py::fence_collectives is what keeps the resident grids clear of it.
extern "C" int caiman_debug_occupy_cus(int workgroups, int microseconds, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(workgroups >= 1 && workgroups <= 256 && microseconds >= 1 && microseconds <= 2000000,
               "debug_occupy_cus: 1..256 workgroups for 1 us .. 2 s (got %d, %d)", workgroups, microseconds);
  static bool attr_set = false;
  const int lds = 96 * 1024;   // more than half of a CU's 160 KB: one such workgroup per CU
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_cus_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return check_launch("debug_occupy_cus attribute");
    attr_set = true;
  }
  hipLaunchKernelGGL(occupy_cus_kernel, dim3((unsigned)workgroups), dim3(256), lds, static_cast<hipStream_t>(stream),
                     (unsigned)microseconds * 100u, (unsigned*)nullptr);
  return check_launch("caiman_debug_occupy_cus");
}

extern "C" int64_t caiman_lstm_workspace_elems(int64_t B, int64_t H, int backward) {
  if (B < 1 || H < 1) return 0;
  const int64_t bp = (B + 31) / 32 * 32;
  return 4 * H * H + 2 * bp * (backward ? 4 * H : H);
}

namespace caiman { namespace {
template <typename T>
__global__ __launch_bounds__(256) void dropout_mask_kernel(T* __restrict__ out, int64_t n, uint64_t seed, uint64_t base, float p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = static_cast<T>(drop_scale(seed, base + (uint64_t)i, p, 1.f / (1.f - p)));
}
} }  // namespace

// keep/scale factors the LSTM kernels apply for element counters base .. base+n-1 (tests, debugging)
extern "C" int caiman_lstm_dropout_mask(void* out, int64_t n, uint64_t seed, uint64_t base, float p, int dtype,
                                        caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(out && n >= 0 && p >= 0.f && p < 1.f, "lstm_dropout_mask: bad arguments");
  CAIMAN_CHECK(dtype == CAIMAN_F16 || dtype == CAIMAN_BF16 || dtype == CAIMAN_F32, "lstm_dropout_mask: f16 / bf16 / f32");
  if (n == 0) return CAIMAN_OK;
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == CAIMAN_BF16) hipLaunchKernelGGL((dropout_mask_kernel<bf16_t>), grid, dim3(256), 0, s, (bf16_t*)out, n, seed, base, p);
  else if (dtype == CAIMAN_F16) hipLaunchKernelGGL((dropout_mask_kernel<f16_t>), grid, dim3(256), 0, s, (f16_t*)out, n, seed, base, p);
  else hipLaunchKernelGGL((dropout_mask_kernel<float>), grid, dim3(256), 0, s, (float*)out, n, seed, base, p);
  return check_launch("caiman_lstm_dropout_mask");
}

// Weight-resident chunk kernels (above): 0 = per-timestep launches only, 1 (default) = one launch per call where the shapes
// allow it (interleaved gates, B <= 32, one hidden size per call, slots x H/32 workgroups <= CUs).  Returns the
// previous mode.
extern "C" int caiman_lstm_resident_mode(int mode) {
  return caiman::g_res_mode.exchange(mode == 2 ? 2 : (mode ? 1 : 0));
}

// 1 (default): backward wave calls with H = 512 / 1024 use the 2-D split resident kernel (a workgroup gathers a
// quarter of the dG row, the four K-quarter partials meet in a second hand-off); 0: the round-1 kernel (whole row per
// workgroup).  Returns the previous setting.  For A/B measurements and tests.
// Resident launches as a flat grid whose workgroup -> slot mapping puts a slot on one XCD (1) or as grid (NKS, slots) (0, default).
// Placement only changes the speed.  Returns the previous setting.
extern "C" int caiman_lstm_resident_xcd_roles(int on) { return caiman::g_res_xcd_roles.exchange(on ? 1 : 0); }

extern "C" int caiman_lstm_resident_bwd_split(int on) { return caiman::g_res_bwd_split.exchange(on ? 1 : 0); }

// Batch-tile kernels (forward AND backward): 1 = operands of the next tile-step by LDS-DMA under the MFMAs of the current one (default),
// 0 = the round-2 kernels.  Returns the previous setting.  Results are bit-identical.
extern "C" int caiman_lstm_resident_bt_dma(int on) { return caiman::g_res_bt_dma.exchange(on ? 1 : 0); }

// Mode 2 phase timers of the 2-D split backward kernel (workgroup 0 of slot 0), 10 ns ticks summed over timesteps:
// out8[0..5] = {wait for the quarter's producers, gather + MFMA, partial blocks out + drain, wait for the group,
// partial blocks in + epilogue, drain + barrier}, out8[6] unused, out8[7] = timesteps; out8[8..14] split "gather + MFMA" into
// {DMA issue, stage 0 wait, stage 0 MFMA, stage 1 wait, stage 1 MFMA, later stages' waits, later stages' MFMAs}.
// `out8` has room for 16 values.  Synchronises; clears the counters.
extern "C" int caiman_lstm_resident_profile_bwd2(uint32_t* out8) {
  using namespace caiman;
  int dev = 0;
  if (!out8 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return CAIMAN_ERR_INVALID;
  if (hipDeviceSynchronize() != hipSuccess) return check_launch("lstm resident profile");
  std::lock_guard<std::mutex> lk(g_res_mu);
  for (int i = 0; i < 16; ++i) out8[i] = 0;
  if (!g_res[dev].ok) return CAIMAN_OK;
  volatile unsigned* f = g_res[dev].fail_host;
  for (int i = 0; i < 16; ++i) { out8[i] = f[kResProfBwd2 + i]; f[kResProfBwd2 + i] = 0; }
  return CAIMAN_OK;
}

// Overwrites the failure count (0: re-admit the resident kernels after an incident has been dealt with; tests use a
// non-zero value to check that the process then keeps to the per-timestep kernels).  Returns the previous count.
extern "C" int caiman_lstm_resident_set_failures(int count) {
  using namespace caiman;
  ResState* st = res_state();
  if (!st) return 0;
  std::lock_guard<std::mutex> lk(g_res_mu);
  volatile unsigned* f = st->fail_host;
  const int prev = (int)f[0];
  if (count <= 0 && prev != 0) {
    // an aborted launch leaves its counter block (and the block it was scrubbing) in an unknown state
    (void)hipDeviceSynchronize();
    for (int i = 0; i < kResPool; ++i) {
      (void)hipMemset(st->sync[i], 0, kResSyncBytes);
      if (st->sync_bt[i]) (void)hipMemset(st->sync_bt[i], 0, kResSyncBytesBT);
    }
  }
  f[0] = (unsigned)(count < 0 ? 0 : count);
  return prev;
}

// Mode 2 = mode 1 with phase timers in workgroup 0 of slot 0: out[0..4] forward, out[5..9] backward, each
// {wait for peers, operand row into LDS (+ MFMA stages, backward), MFMA + cell update / epilogue, drain + barrier}
// in 10 ns ticks summed over timesteps, then the number of timesteps.  Clears the counters.
extern "C" int caiman_lstm_resident_profile(uint32_t* out10) {
  using namespace caiman;
  int dev = 0;
  if (!out10 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return CAIMAN_ERR_INVALID;
  if (hipDeviceSynchronize() != hipSuccess) return check_launch("lstm resident profile");
  std::lock_guard<std::mutex> lk(g_res_mu);
  for (int i = 0; i < 10; ++i) out10[i] = 0;
  if (!g_res[dev].ok) return CAIMAN_OK;
  volatile unsigned* f = g_res[dev].fail_host;
  for (int i = 0; i < 5; ++i) { out10[i] = f[kResProfFwd + i]; out10[5 + i] = f[kResProfBwd + i]; f[kResProfFwd + i] = 0; f[kResProfBwd + i] = 0; }
  return CAIMAN_OK;
}

// 1 when a wave call of this shape (interleaved gates, n_slots slots of hidden size H, several timesteps) would be
// served by a resident launch on the current device.  Callers use it to decide whether to ask for the fused bias
// gradients (`dbias`): the per-timestep path fills them too, but with an extra reduction launch per call.
extern "C" int caiman_lstm_resident_would_run(int64_t B, int64_t H, int n_slots) {
  using namespace caiman;
  if (!g_res_mode.load(std::memory_order_relaxed) || B < 1 || H % 32 != 0 || n_slots < 1 || n_slots > kMaxSlots) return 0;
  const bool sliced = res_sliced_shape(B, H);   // B > 32 outside the batch-tile kernels' shapes: 32-row slices of the B <= 32 kernels
  if (B > 32 && !sliced && (B > 32 * kResMaxTiles || (H != 512 && H != 1024))) return 0;
  const int nks = (int)(H / 32);
  if (!(nks == 2 || nks == 4 || nks == 8 || nks == 16 || nks == 24 || nks == 32 || nks == 48)) return 0;
  ResState* st = res_state();
  if (!st || *reinterpret_cast<volatile unsigned*>(st->fail_host) != 0u) return 0;
  // B <= 32: slots that do not fit the chip together go out as consecutive launches; the batch-tile kernels do not split
  return (B <= 32 || sliced ? nks <= st->cus : (int64_t)n_slots * nks <= st->cus) ? 1 : 0;
}

// Wave calls served by a resident launch since the library was loaded (callers that account launches and bytes
// compare the value before and after a call).
extern "C" int64_t caiman_lstm_resident_launches(void) {
  return (int64_t)caiman::g_res_launches.load(std::memory_order_relaxed);
}

// Number of resident-kernel workgroups that gave up waiting for their peers since the library was loaded (0 in a
// healthy run; results of such a launch are invalid).  Reads host memory: no device synchronisation.
extern "C" int caiman_lstm_resident_failures(void) {
  using namespace caiman;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  std::lock_guard<std::mutex> lk(g_res_mu);
  return g_res[dev].ok ? (int)*reinterpret_cast<volatile unsigned*>(g_res[dev].fail_host) : 0;
}

// ---- multi-layer ("wave") interface -----------------------------------------------------------------
extern "C" int caiman_lstm_prepare(const void* R, const void* h0, void* weights_tiled, void* ring, void* dC,
                                   int64_t B, int64_t H, int dtype, int backward, int gate_layout,
                                   caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(B >= 1 && H >= 32 && H % 32 == 0, "lstm_prepare: H must be a positive multiple of 32 (got %lld)", (long long)H);
  CAIMAN_CHECK(dtype == CAIMAN_F16 || dtype == CAIMAN_BF16, "lstm_prepare: the wave interface is f16 / bf16 only");
  CAIMAN_CHECK(weights_tiled && ring && (backward ? dC != nullptr : h0 != nullptr), "lstm_prepare: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool il = (gate_layout & 1) != 0, zeroed = (gate_layout & 2) != 0;   // bit 1: the caller has already zeroed ring (and dC)
  if (dtype == CAIMAN_BF16)
    return backward ? prepare_bwd<bf16_t>((const bf16_t*)R, (bf16_t*)weights_tiled, (bf16_t*)ring, (float*)dC, B, H, s, il, zeroed)
                    : prepare_fwd<bf16_t>((const bf16_t*)R, (const bf16_t*)h0, (bf16_t*)weights_tiled, (bf16_t*)ring, B, H, s, zeroed);
  return backward ? prepare_bwd<f16_t>((const f16_t*)R, (f16_t*)weights_tiled, (f16_t*)ring, (float*)dC, B, H, s, il, zeroed)
                  : prepare_fwd<f16_t>((const f16_t*)R, (const f16_t*)h0, (f16_t*)weights_tiled, (f16_t*)ring, B, H, s, zeroed);
}

extern "C" int caiman_lstm_wave_fwd(const caiman_lstm_fwd_slot_t* slots, int n_slots, int n_launches, int64_t B,
                                    int64_t H, int dtype, int hard, int gate_layout, uint64_t seed,
                                    caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(slots && n_slots >= 1 && n_slots <= kMaxSlots, "lstm_wave_fwd: 1..%d slots", kMaxSlots);
  CAIMAN_CHECK(B >= 1 && B <= 32 * 65535 && H >= 32 && H % 32 == 0, "lstm_wave_fwd: bad extents");
  CAIMAN_CHECK(dtype == CAIMAN_F16 || dtype == CAIMAN_BF16, "lstm_wave_fwd: f16 / bf16 only");
  CAIMAN_CHECK(n_launches >= 0, "lstm_wave_fwd: negative launch count");
  if (n_launches == 0) return CAIMAN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto go_ = [&](auto tag) -> int {
    using T = decltype(tag);
    FwdSlots<T> w{};
    for (int i = 0; i < n_slots; ++i) {
      CAIMAN_CHECK(slots[i].weights_tiled && slots[i].gates && slots[i].c && slots[i].y && slots[i].ring,
                   "lstm_wave_fwd: null pointer in slot %d", i);
      CAIMAN_CHECK(slots[i].nsteps >= 0 && slots[i].nsteps <= n_launches, "lstm_wave_fwd: slot %d nsteps out of range", i);
      w.Rtile[i] = (const T*)slots[i].weights_tiled; w.g[i] = (T*)slots[i].gates; w.c[i] = (T*)slots[i].c;
      w.y[i] = (T*)slots[i].y; w.hring[i] = (T*)slots[i].ring; w.parity[i] = slots[i].parity & 1;
      w.nsteps[i] = slots[i].nsteps;
      w.ymask[i] = (T*)slots[i].y_masked; w.drop_base[i] = slots[i].drop_counter; w.drop_p[i] = slots[i].drop_p;
      CAIMAN_CHECK(slots[i].hidden == 0 || (slots[i].hidden >= 32 && slots[i].hidden % 32 == 0 && slots[i].hidden <= H),
                   "lstm_wave_fwd: slot %d hidden size must be 0 or a multiple of 32 not above H", i);
      w.hidden[i] = slots[i].hidden;
      CAIMAN_CHECK(slots[i].drop_p >= 0.f && slots[i].drop_p < 1.f, "lstm_wave_fwd: dropout p must be in [0,1)");
    }
    w.seed = seed;
    if (gate_layout) {
      int err = CAIMAN_OK;
      if (hard ? try_fwd_resident_bt<T, true>(w, n_slots, n_launches, B, H, s, &err)
               : try_fwd_resident_bt<T, false>(w, n_slots, n_launches, B, H, s, &err))
        return err;
      if (hard ? try_fwd_resident<T, true>(w, n_slots, n_launches, B, H, s, &err)
               : try_fwd_resident<T, false>(w, n_slots, n_launches, B, H, s, &err))
        return err;
      if (hard ? try_fwd_resident_slices<T, true>(w, n_slots, n_launches, B, H, s, &err)
               : try_fwd_resident_slices<T, false>(w, n_slots, n_launches, B, H, s, &err))
        return err;
    }
    if (gate_layout)
      return hard ? launch_fwd_waves<T, true, true>(w, n_slots, n_launches, B, H, s)
                  : launch_fwd_waves<T, false, true>(w, n_slots, n_launches, B, H, s);
    return hard ? launch_fwd_waves<T, true, false>(w, n_slots, n_launches, B, H, s)
                : launch_fwd_waves<T, false, false>(w, n_slots, n_launches, B, H, s);
  };
  return dtype == CAIMAN_BF16 ? go_(bf16_t{}) : go_(f16_t{});
}

extern "C" int caiman_lstm_wave_bwd(const caiman_lstm_bwd_slot_t* slots, int n_slots, int n_launches, int64_t B,
                                    int64_t H, int dtype, int hard, int gate_layout, uint64_t seed,
                                    caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(slots && n_slots >= 1 && n_slots <= kMaxSlots, "lstm_wave_bwd: 1..%d slots", kMaxSlots);
  CAIMAN_CHECK(B >= 1 && B <= 32 * 65535 && H >= 32 && H % 32 == 0, "lstm_wave_bwd: bad extents");
  CAIMAN_CHECK(dtype == CAIMAN_F16 || dtype == CAIMAN_BF16, "lstm_wave_bwd: f16 / bf16 only");
  CAIMAN_CHECK(n_launches >= 0, "lstm_wave_bwd: negative launch count");
  if (n_launches == 0) return CAIMAN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto go_ = [&](auto tag) -> int {
    using T = decltype(tag);
    BwdSlots<T> w{};
    for (int i = 0; i < n_slots; ++i) {
      CAIMAN_CHECK(slots[i].weights_tiled && slots[i].gates && slots[i].c && slots[i].delta && slots[i].dG &&
                       slots[i].ring && slots[i].dC, "lstm_wave_bwd: null pointer in slot %d", i);
      CAIMAN_CHECK(slots[i].nsteps >= 0 && slots[i].nsteps <= n_launches, "lstm_wave_bwd: slot %d nsteps out of range", i);
      w.Rttile[i] = (const T*)slots[i].weights_tiled; w.g[i] = (const T*)slots[i].gates; w.c[i] = (const T*)slots[i].c;
      w.delta[i] = (const T*)slots[i].delta; w.d_st[i] = slots[i].delta_stride_t; w.d_sb[i] = slots[i].delta_stride_b;
      w.dG[i] = (T*)slots[i].dG; w.dring[i] = (T*)slots[i].ring; w.dC[i] = (float*)slots[i].dC;
      w.parity[i] = slots[i].parity & 1; w.nsteps[i] = slots[i].nsteps; w.has_in0[i] = slots[i].has_next ? 1 : 0;
      w.drop_base[i] = slots[i].drop_counter; w.drop_p[i] = slots[i].drop_p;
      CAIMAN_CHECK(slots[i].drop_p >= 0.f && slots[i].drop_p < 1.f, "lstm_wave_bwd: dropout p must be in [0,1)");
      CAIMAN_CHECK(slots[i].hidden == 0 || (slots[i].hidden >= 32 && slots[i].hidden % 32 == 0 && slots[i].hidden <= H),
                   "lstm_wave_bwd: slot %d hidden size must be 0 or a multiple of 32 not above H", i);
      w.hidden[i] = slots[i].hidden;
      CAIMAN_CHECK(slots[i].dbias == nullptr || gate_layout != 0, "lstm_wave_bwd: dbias needs the interleaved gate layout");
      w.dbias[i] = slots[i].dbias;
    }
    w.seed = seed;
    if (gate_layout) {
      int err = CAIMAN_OK;
      if (hard ? try_bwd_resident2_bt<T, true>(w, n_slots, n_launches, B, H, s, &err)
               : try_bwd_resident2_bt<T, false>(w, n_slots, n_launches, B, H, s, &err))
        return err;
      if (hard ? try_bwd_resident2<T, true>(w, n_slots, n_launches, B, H, s, &err)
               : try_bwd_resident2<T, false>(w, n_slots, n_launches, B, H, s, &err))
        return err;
      if (hard ? try_bwd_resident<T, true>(w, n_slots, n_launches, B, H, s, &err)
               : try_bwd_resident<T, false>(w, n_slots, n_launches, B, H, s, &err))
        return err;
      if (hard ? try_bwd_resident_slices<T, true>(w, n_slots, n_launches, B, H, s, &err)
               : try_bwd_resident_slices<T, false>(w, n_slots, n_launches, B, H, s, &err))
        return err;
    }
    if (gate_layout)
      return hard ? launch_bwd_waves<T, true, true>(w, n_slots, n_launches, B, H, s)
                  : launch_bwd_waves<T, false, true>(w, n_slots, n_launches, B, H, s);
    return hard ? launch_bwd_waves<T, true, false>(w, n_slots, n_launches, B, H, s)
                : launch_bwd_waves<T, false, false>(w, n_slots, n_launches, B, H, s);
  };
  return dtype == CAIMAN_BF16 ? go_(bf16_t{}) : go_(f16_t{});
}

extern "C" int caiman_lstm_fused_fwd(const void* R, void* gates, void* c, void* y, void* work, int64_t T,
                                     int64_t B, int64_t H, int dtype, int hard, caiman_stream_t stream) {
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
    auto wp = static_cast<scalar_t*>(work);
    return hard ? run_fwd<scalar_t, true>(Rp, gp, cp, yp, wp, T, B, H, s)
                : run_fwd<scalar_t, false>(Rp, gp, cp, yp, wp, T, B, H, s);
  });
}

extern "C" int caiman_lstm_fused_bwd(const void* R, const void* gates, const void* c, const void* delta,
                                     int64_t delta_stride_t, int64_t delta_stride_b, void* dG, void* dC,
                                     void* work, int64_t T, int64_t B, int64_t H, int dtype, int hard,
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
    auto Rtp = static_cast<scalar_t*>(work);
    return hard ? run_bwd<scalar_t, true>(Rp, gp, cp, dp, delta_stride_t, delta_stride_b, dGp, dCp, Rtp, T, B, H, s)
                : run_bwd<scalar_t, false>(Rp, gp, cp, dp, delta_stride_t, delta_stride_b, dGp, dCp, Rtp, T, B, H, s);
  });
}
