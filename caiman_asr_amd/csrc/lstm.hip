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

// Rttile[((blk*nk4 + s)*16 + n)*32 + kk] = R[(s*32 + kk)*H + blk*16 + n]   (nk4 = 4H/32)
template <typename T>
__global__ __launch_bounds__(256) void tile_Rt_bwd_kernel(const T* __restrict__ R, T* __restrict__ out, int H) {
  __shared__ T t[32][17];
  const int nk4 = (4 * H) >> 5;
  const int s = blockIdx.x % nk4, blk = blockIdx.x / nk4;
  // read R[s*32 + kk][blk*16 + n] : 32 rows x 16 cols
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int kk = e >> 4, n = e & 15;
    t[kk][n] = R[(int64_t)(s * 32 + kk) * H + blk * 16 + n];
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

// ---- forward step: grid (H/4, ceil(B/32)), 256 threads = 4 waves --------------------------
//   output tile: 32 batch rows x 16 gate columns (4 gates x 4 hidden units); the H/32 k-steps
//   are dealt round-robin to the 4 waves; NK = k-steps per wave (0 = runtime loop).
template <typename T, bool HARD, int NK>
__global__ __launch_bounds__(256) void lstm_fwd_step_mfma(const T* __restrict__ Rtile, T* __restrict__ g,
                                                          const T* __restrict__ c0, T* __restrict__ c1,
                                                          const T* __restrict__ h_in, T* __restrict__ y1,
                                                          T* __restrict__ h_out, int B, int H) {
  using frag = typename frag8<T>::type;
  __shared__ float tile[4][2][16][17];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int j0 = blockIdx.x * 4, mt = blockIdx.y, m0 = mt * 32;
  const int nk = H >> 5;

  // epilogue inputs first: their latency hides under the operand loads
  const int eb = tid >> 2, eu = tid & 3;
  const int be = m0 + eb, ne = j0 + eu;
  const bool ep = (tid < 128) && (be < B);
  const int64_t gb = (int64_t)be * 4 * H + ne;
  float pre[4] = {0.f, 0.f, 0.f, 0.f};
  float cprev = 0.f;
  if (ep) {
#pragma unroll
    for (int q = 0; q < 4; ++q) pre[q] = static_cast<float>(g[gb + (int64_t)q * H]);
    cprev = static_cast<float>(c0[(int64_t)be * H + ne]);
  }

  const T* Bbase = Rtile + ((int64_t)blockIdx.x * nk * 16 + r) * 32 + 8 * kq;  // + s*512
  const T* Abase = h_in + ((int64_t)mt * nk * 32 + r) * 32 + 8 * kq;           // + s*1024 (+512: rows 16..31)
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (NK > 0) {
    frag bf[NK], a0[NK], a1[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int s = wave + 4 * i;
      const int sc = s < nk ? s : 0;  // NK*4 may exceed nk: clamp the address, zero the product below
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
  } else {
#pragma unroll 4
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
  c1[(int64_t)be * H + ne] = static_cast<T>(c);
  g[gb] = static_cast<T>(i);
  g[gb + H] = static_cast<T>(f);
  g[gb + 2 * (int64_t)H] = static_cast<T>(gg);
  g[gb + 3 * (int64_t)H] = static_cast<T>(o);
}

// ---- backward step: grid (H/16, ceil(B/32)), 1024 threads = 16 waves ------------------------
//   dh tile: 32 batch rows x 16 hidden units, K = 4H dealt round-robin to the 16 waves.
template <typename T, bool HARD, int NK>
__global__ __launch_bounds__(1024) void lstm_bwd_step_mfma(const T* __restrict__ Rttile, const T* __restrict__ g,
                                                           const T* __restrict__ c_prev,
                                                           const T* __restrict__ c_cur,
                                                           const T* __restrict__ delta, int64_t d_sb,
                                                           const T* __restrict__ dG_in, T* __restrict__ dG,
                                                           T* __restrict__ dG_out, float* __restrict__ dC, int B,
                                                           int H) {
  using frag = typename frag8<T>::type;
  __shared__ float tile[16][2][16][17];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n0 = blockIdx.x * 16, mt = blockIdx.y, m0 = mt * 32;
  const int nk4 = (4 * H) >> 5;

  const int eb = tid >> 4, eu = tid & 15;  // 32 rows x 16 units on the first 512 threads
  const int be = m0 + eb, ne = n0 + eu;
  const bool ep = (tid < 512) && (be < B);
  const int64_t gb = (int64_t)be * 4 * H + ne;
  float dy = 0.f, gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, cp = 0.f, cc = 0.f, dcf = 0.f;
  if (ep) {
    dy = static_cast<float>(delta[(int64_t)be * d_sb + ne]);
    gi = static_cast<float>(g[gb]);
    gf = static_cast<float>(g[gb + H]);
    gg = static_cast<float>(g[gb + 2 * (int64_t)H]);
    go = static_cast<float>(g[gb + 3 * (int64_t)H]);
    cp = static_cast<float>(c_prev[(int64_t)be * H + ne]);
    cc = static_cast<float>(c_cur[(int64_t)be * H + ne]);
    dcf = dC[(int64_t)be * H + ne];
  }

  if (dG_in) {
    const int r = lane & 15, kq = lane >> 4;
    const T* Bbase = Rttile + ((int64_t)blockIdx.x * nk4 * 16 + r) * 32 + 8 * kq;  // + s*512
    const T* Abase = dG_in + ((int64_t)mt * nk4 * 32 + r) * 32 + 8 * kq;           // + s*1024 (+512)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NK > 0) {
      frag bf[NK], a0[NK], a1[NK];
#pragma unroll
      for (int i = 0; i < NK; ++i) {
        const int s = wave + 16 * i;
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
    } else {
#pragma unroll 2
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
  if (dG_in) {
    const int half = eb >> 4, rr = eb & 15;
#pragma unroll
    for (int w = 0; w < 16; ++w) dy += tile[w][half][rr][eu];
  }
  const float ct = Act<float, HARD>::tanhv(cc);
  const float dc = dy * go * Act<float, HARD>::tanh_prime(ct) + dcf;
  const T vI = static_cast<T>(dc * gg * Act<float, HARD>::sigm_prime(gi));
  const T vF = static_cast<T>(dc * cp * Act<float, HARD>::sigm_prime(gf));
  const T vG = static_cast<T>(dc * gi * Act<float, HARD>::tanh_prime(gg));
  const T vO = static_cast<T>(dy * ct * Act<float, HARD>::sigm_prime(go));
  if (dG_out) {  // tiled copy for the next (earlier-in-time) step
    dG_out[tiled_index(be, ne, nk4)] = vI;
    dG_out[tiled_index(be, H + ne, nk4)] = vF;
    dG_out[tiled_index(be, 2 * H + ne, nk4)] = vG;
    dG_out[tiled_index(be, 3 * H + ne, nk4)] = vO;
  }
  dG[gb] = vI;
  dG[gb + H] = vF;
  dG[gb + 2 * (int64_t)H] = vG;
  dG[gb + 3 * (int64_t)H] = vO;
  dC[(int64_t)be * H + ne] = dc * gf;
}

template <typename T>
constexpr bool kHasMfma = std::is_same<T, bf16_t>::value || std::is_same<T, f16_t>::value;

inline int64_t pad32(int64_t b) { return (b + 31) / 32 * 32; }

template <typename T, bool HARD>
int run_fwd(const T* R, T* gates, T* c, T* y, T* work, int64_t Tn, int64_t B, int64_t H, hipStream_t s) {
  const int64_t go = B * 4 * H, so = B * H;
  bool mfma = false;
  if constexpr (kHasMfma<T>) mfma = (H % 32 == 0) && work != nullptr;
  if constexpr (kHasMfma<T>) {
    if (mfma) {
      T* Rtile = work;
      T* hring = work + 4 * H * H;
      const int64_t hsz = pad32(B) * H;
      if (hipMemsetAsync(hring, 0, sizeof(T) * (size_t)(2 * hsz), s) != hipSuccess) return check_launch("lstm_fwd memset");
      hipLaunchKernelGGL((tile_R_fwd_kernel<T>), dim3((unsigned)((4 * H * H + 255) / 256)), dim3(256), 0, s, R, Rtile, (int)H);
      hipLaunchKernelGGL((tile_rows_kernel<T>), dim3((unsigned)((B * H + 255) / 256)), dim3(256), 0, s, y, hring, (int)B, (int)H);
      const int nkw = (int)(((H >> 5) + 3) / 4);  // k-steps per wave
      const dim3 grid((unsigned)(H / 4), (unsigned)((B + 31) / 32));
      for (int64_t t = 0; t < Tn; ++t) {
        T* hin = hring + (t & 1) * hsz;
        T* hout = hring + ((t + 1) & 1) * hsz;
#define CAIMAN_FWD(NKV)                                                                                       \
  hipLaunchKernelGGL((lstm_fwd_step_mfma<T, HARD, NKV>), grid, dim3(256), 0, s, Rtile, gates + go * t,        \
                     c + so * t, c + so * (t + 1), hin, y + so * (t + 1), hout, (int)B, (int)H)
        switch (nkw) {
          case 1: CAIMAN_FWD(1); break;
          case 2: CAIMAN_FWD(2); break;
          case 4: CAIMAN_FWD(4); break;
          case 6: CAIMAN_FWD(6); break;
          case 8: CAIMAN_FWD(8); break;
          case 12: CAIMAN_FWD(12); break;
          default: CAIMAN_FWD(0); break;
        }
#undef CAIMAN_FWD
      }
      return check_launch("caiman_lstm_fused_fwd");
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
  if (hipMemsetAsync(dC, 0, sizeof(acc_t<T>) * (size_t)so, s) != hipSuccess) return check_launch("lstm_bwd memset");
  bool mfma = false;
  if constexpr (kHasMfma<T>) mfma = (H % 32 == 0) && work != nullptr;
  if constexpr (kHasMfma<T>) {
    if (mfma) {
      T* Rttile = work;
      T* dring = work + 4 * H * H;
      const int64_t dsz = pad32(B) * 4 * H;
      if (Tn > 1) {
        if (hipMemsetAsync(dring, 0, sizeof(T) * (size_t)(2 * dsz), s) != hipSuccess) return check_launch("lstm_bwd memset");
        hipLaunchKernelGGL((tile_Rt_bwd_kernel<T>), dim3((unsigned)((H / 16) * (4 * H / 32))), dim3(256), 0, s, R, Rttile, (int)H);
      }
      const int nkw = (int)(((4 * H >> 5) + 15) / 16);
      const dim3 grid((unsigned)(H / 16), (unsigned)((B + 31) / 32));
      for (int64_t t = Tn - 1; t >= 0; --t) {
        const T* din = (t < Tn - 1) ? dring + ((t + 1) & 1) * dsz : nullptr;
        T* dout = (t > 0) ? dring + (t & 1) * dsz : nullptr;
#define CAIMAN_BWD(NKV)                                                                                       \
  hipLaunchKernelGGL((lstm_bwd_step_mfma<T, HARD, NKV>), grid, dim3(1024), 0, s, Rttile, gates + go * t,      \
                     c + so * t, c + so * (t + 1), delta + d_st * t, d_sb, din, dG + go * t, dout, dC, (int)B, (int)H)
        switch (nkw) {
          case 1: CAIMAN_BWD(1); break;
          case 2: CAIMAN_BWD(2); break;
          case 4: CAIMAN_BWD(4); break;
          case 6: CAIMAN_BWD(6); break;
          case 8: CAIMAN_BWD(8); break;
          default: CAIMAN_BWD(0); break;
        }
#undef CAIMAN_BWD
      }
      return check_launch("caiman_lstm_fused_bwd");
    }
  }
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

extern "C" int64_t caiman_lstm_workspace_elems(int64_t B, int64_t H, int backward) {
  if (B < 1 || H < 1) return 0;
  const int64_t bp = (B + 31) / 32 * 32;
  return 4 * H * H + 2 * bp * (backward ? 4 * H : H);
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
