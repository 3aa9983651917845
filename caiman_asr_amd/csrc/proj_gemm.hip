// Grouped input-projection GEMM of the layer-pipelined LSTM stacks.
//
// The reference computes a layer's gate pre-activations for the whole sequence with one library call,
// `gates = torch.addmm(bias, x, W_ih.t())` (training/lib/src/rnnt_ext/custom_lstm/lstm.py:51-55), and the input gradient
// of the layer below with `dG @ W_ih` (autograd of the same call).  In the layer pipeline (custom_lstm/encoder_pipe.py)
// those products are cut into one chunk per layer and tick: 5-7 small problems (M = 512 ... 1024 rows, N = 4H, K = H or
// 2H) between two weight-resident recurrence launches, which the library serves with three launches at 220 - 440 TFLOP/s.
// This kernel takes all problems of a tick in ONE launch:
//
//   C[m][n] = sum_k A[m][k] * W[n][k] (+ bias[n]),   W row-major [N][K] (both operands K-contiguous)
//
// * workgroup = 4 waves (2 x 2) on a BM x BN tile, BK = 64 per step; the wave tile is (BM/2) x (BN/2).  The MFMA is
//   issued transposed (A operand = 16 rows of W, B operand = 16 rows of the activations), so a lane ends up with FOUR
//   CONSECUTIVE COLUMNS of one output row: 8-byte stores, no transpose through LDS.
// * both operand tiles arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers) into two LDS buffers per operand
//   -- separate LDS objects, so that the compiler's waitcnt pass can tell the buffer being read from the one being filled
//   -- with the XOR swizzle applied on the SOURCE address (the DMA writes lane-linear): the 16-byte piece q of row r sits
//   at position q ^ ((r >> 1) & 7), which spreads the 16 rows of a fragment read over all 64 banks.
// * rows of A and C are addressed as (outer, inner, segment): element (m, k) of A lives at
//   (m / inner) * stride_outer + (m % inner) * stride_inner + (k / kseg) * stride_seg + k % kseg.  That is what lets the
//   kernel read the StackTime view of the layer below ([T, B, H] -> rows (t, b) of f*H features: segment = frame within
//   the stack) and scatter the input gradient back through it without the copy kernels the library path needs.
// * tiles are enumerated problem by problem, the caller putting the problems with the longest K first: the hardware
//   dispatches workgroups in order, so the long tiles start first and the short ones fill in behind them.
#include "common.h"

namespace caiman {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
template <typename T>
struct pfrag {
  using type = __attribute__((ext_vector_type(8))) T;
};
__device__ __forceinline__ f32x4 pmfma(pfrag<bf16_t>::type a, pfrag<bf16_t>::type b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 pmfma(pfrag<f16_t>::type a, pfrag<f16_t>::type b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

constexpr int kProjMax = CAIMAN_PROJ_MAX_PROBLEMS;
struct ProjBatch {
  caiman_proj_problem_t p[kProjMax];
  int tile_begin[kProjMax + 1];
  int tiles_n[kProjMax];
  int n;
};

// KS = 2: two groups of NW waves share the tile and split its K range in halves (the K loop of a workgroup is a serial
// chain of ~1 us steps; for the backward problems, K = 4H with one tile per CU, the chain IS the launch time); the halves
// meet in LDS -- group 1 parks its accumulators in its own stage buffers, group 0 adds them and runs the epilogue.
// CELL: the product is one timestep of an LSTM layer for M rows (live streams / pending hypotheses of the streaming
// decoders): A = [x_t | h_{t-1}] rows, W = [W_ih | W_hh] with its 4H rows ordered [unit][gate] -- the lane that ends up
// with four consecutive columns of a row holds the i, f, g, o pre-activations of ONE unit, so the cell update, the scatter
// of (c, h) into the state pools by slot index and the next layer's input row are the epilogue (training/lib/csrc/lstm.cu:
// 85-135 pointwise after :259-271's GEMM; training/caiman_asr_train/rnnt/beam.py:564-612): the gate matrix never exists.
template <typename T>
struct CellArgs {
  float* c_pool;            // this layer's cell pool [1 + slots][H], f32
  T* h_pool;                // this layer's hidden pool [1 + slots][H]
  const T* h_pool_next;     // next layer's hidden pool or NULL (last layer)
  const int32_t* slot_in;   // [M]: pool row - 1 the old state is read from
  const int32_t* slot_out;  // [M]: pool row - 1 the new state is written to
  T* x_next;                // [M][ldx]: [ h | h_pool_next[slot_in] ]
  int64_t ldx;
  int H;
};

template <typename T, int BM, int BN, int NS, int NW, int KS, bool CELL = false>
__global__ __launch_bounds__(64 * NW * KS, 1) void proj_gemm_kernel(ProjBatch pb, CellArgs<T> cell) {
  using frag = typename pfrag<T>::type;
  constexpr int BK = 64;
  constexpr int WGM = NW / 2;                     // waves along M; two along N
  constexpr int WM = BM / WGM, WN = BN / 2, TM = WM / 16, TN = WN / 16;
  constexpr int ABLK = BM / 8 / NW, WBLK = BN / 8 / NW;   // 8-row blocks (1 KB, one DMA instruction) per wave and operand
  static_assert(KS == 1 || (KS == 2 && NS == 2 && BM == 128 && BN == 128 && NW == 8), "K split: 128 x 128 tiles, two stages");
  __shared__ __attribute__((aligned(1024))) T gA0[BM * BK], gA1[BM * BK], gW0[BN * BK], gW1[BN * BK];
  __shared__ __attribute__((aligned(1024))) T lA2[NS >= 3 ? BM * BK : 8], lW2[NS >= 3 ? BN * BK : 8];   // third stage
  __shared__ __attribute__((aligned(1024))) T lA3[NS == 4 ? BM * BK : 8], lW3[NS == 4 ? BN * BK : 8];   // fourth stage
  __shared__ __attribute__((aligned(1024))) T hA0[KS == 2 ? BM * BK : 8], hA1[KS == 2 ? BM * BK : 8], hW0[KS == 2 ? BN * BK : 8],
      hW1[KS == 2 ? BN * BK : 8];                                                                        // second K group

  const int tid = threadIdx.x, lane = tid & 63;
  const int grp = KS == 2 ? (tid >> 6) / NW : 0, wave = (tid >> 6) % NW;
  T* const lA0 = grp ? hA0 : gA0;
  T* const lA1 = grp ? hA1 : gA1;
  T* const lW0 = grp ? hW0 : gW0;
  T* const lW1 = grp ? hW1 : gW1;
  const int bid = blockIdx.x;
  int g = 0;
#pragma unroll
  for (int i = 1; i < kProjMax; ++i)
    if (i < pb.n && bid >= pb.tile_begin[i]) g = i;
  const caiman_proj_problem_t& P = pb.p[g];
  const int t = bid - pb.tile_begin[g];
  const int tiles_n = pb.tiles_n[g];
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  const int M = P.M, K = P.K;
  const T* __restrict__ Ap = static_cast<const T*>(P.a);
  const T* __restrict__ Wp = static_cast<const T*>(P.w);

  // per-lane DMA sources: lane l of a block brings the 16 bytes that belong at (row l >> 3, position l & 7).  Kept as
  // 32-bit byte offsets from a wave-uniform base (scalar base + vector offset addressing: half the address registers)
  unsigned a_off[ABLK], w_off[WBLK];
#pragma unroll
  for (int i = 0; i < ABLK; ++i) {
    const int rt = (wave + NW * i) * 8 + (lane >> 3);
    const int m = m0 + rt < M ? m0 + rt : M - 1;          // rows past M re-read the last row; never stored
    const int q = (lane & 7) ^ ((rt >> 1) & 7);
    a_off[i] = (unsigned)(((int64_t)(m / P.a_inner - m0 / P.a_inner) * P.a_stride_outer +
                           (int64_t)(m % P.a_inner) * P.a_stride_inner + q * 8) * (int64_t)sizeof(T));
  }
  const char* a_base = reinterpret_cast<const char*>(Ap + (int64_t)(m0 / P.a_inner) * P.a_stride_outer);
#pragma unroll
  for (int i = 0; i < WBLK; ++i) {
    const int rt = (wave + NW * i) * 8 + (lane >> 3);
    const int q = (lane & 7) ^ ((rt >> 1) & 7);
    w_off[i] = (unsigned)((rt * K + q * 8) * (int)sizeof(T));
  }
  const char* w_base = reinterpret_cast<const char*>(Wp + (int64_t)n0 * K);
  const int a_kseg = P.a_kseg;
  const int64_t a_stride_seg = P.a_stride_seg;

  auto issue = [&](T* lA, T* lW, int k0) {
    const int seg = k0 / a_kseg;
    const char* ab = a_base + ((int64_t)seg * a_stride_seg + (k0 - seg * a_kseg)) * (int64_t)sizeof(T);
    const char* wb = w_base + (int64_t)k0 * (int64_t)sizeof(T);
#pragma unroll
    for (int i = 0; i < ABLK; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ab + a_off[i]),
                                       (__attribute__((address_space(3))) void*)(lA + (wave + NW * i) * 8 * BK), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < WBLK; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + w_off[i]),
                                       (__attribute__((address_space(3))) void*)(lW + (wave + NW * i) * 8 * BK), 16, 0, 0);
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, kq = lane >> 4;
  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const T* lA, const T* lW) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int q = kk * 4 + kq;
      frag wf[TN], af[TM];
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        const int row = wn * WN + a * 16 + r16;
        wf[a] = *reinterpret_cast<const frag*>(lW + row * BK + ((q ^ ((row >> 1) & 7)) * 8));
      }
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int row = wm * WM + b * 16 + r16;
        af[b] = *reinterpret_cast<const frag*>(lA + row * BK + ((q ^ ((row >> 1) & 7)) * 8));
      }
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = pmfma(wf[a], af[b], acc[a][b]);
    }
  };

  const int nk = K / BK / KS;
  const int kbeg = grp * (K / KS);
  if constexpr (NS == 2 && KS == 2) {
    issue(lA0, lW0, kbeg);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int ks = 0; ks < nk; ks += 2) {   // K % 256 == 0: an even number of steps per group
      issue(lA1, lW1, kbeg + (ks + 1) * BK);
      compute(lA0, lW0);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      if (ks + 2 < nk) issue(lA0, lW0, kbeg + (ks + 2) * BK);
      compute(lA1, lW1);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
    // the two K halves meet: 64 accumulator tiles of 1 KB (8 waves x 8 tiles) fit group 1's four 16 KB stage buffers
    T* const park = (wave >> 1) == 0 ? hA0 : ((wave >> 1) == 1 ? hA1 : ((wave >> 1) == 2 ? hW0 : hW1));
    f32x4* const slot = reinterpret_cast<f32x4*>(park) + ((wave & 1) * (TN * TM)) * 64 + lane;
    if (grp == 1) {
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) slot[(a * TM + b) * 64] = acc[a][b];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] += slot[(a * TM + b) * 64];
  } else if constexpr (NS == 2) {
    issue(lA0, lW0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMAs have landed
    __syncthreads();
    for (int ks = 0; ks < nk; ks += 2) {   // K % 128 == 0: an even number of steps, both buffers named statically
      issue(lA1, lW1, (ks + 1) * BK);
      compute(lA0, lW0);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      if (ks + 2 < nk) issue(lA0, lW0, (ks + 2) * BK);
      compute(lA1, lW1);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
  } else if constexpr (NS == 4) {
    // ring of four stages, three in flight (csrc/joint_gemm.hip's loop): the backward ticks' products have K = 4H and about
    // one 128 x 128 tile per CU, so the step time is what the L2 -> LDS feed delivers once its latency is covered
    static_assert(ABLK + WBLK == 4, "vmcnt immediates below: four DMA instructions per wave and stage");
    auto steady = [&](const T* cA, const T* cW, T* nA, T* nW, int ks) {
      __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8) lgkmcnt(0): stages ks + 1, ks + 2 may still fly
      __builtin_amdgcn_s_barrier();
      issue(nA, nW, (ks + 3) * BK);
      compute(cA, cW);
    };
    auto drain = [&](const T* cA, const T* cW, auto w) {
      __builtin_amdgcn_s_waitcnt(decltype(w)::value);
      __builtin_amdgcn_s_barrier();
      compute(cA, cW);
    };
    using w8 = std::integral_constant<int, 0x0078>;
    using w4 = std::integral_constant<int, 0x0074>;
    using w0 = std::integral_constant<int, 0x0070>;
    issue(lA0, lW0, 0);
    issue(lA1, lW1, BK);
    issue(lA2, lW2, 2 * BK);             // the launcher sends K < 256 elsewhere: at least four steps
    int ks = 0;
    for (; ks + 7 <= nk; ks += 4) {
      steady(lA0, lW0, lA3, lW3, ks);
      steady(lA1, lW1, lA0, lW0, ks + 1);
      steady(lA2, lW2, lA1, lW1, ks + 2);
      steady(lA3, lW3, lA2, lW2, ks + 3);
    }
    switch (nk - ks) {                   // 3 .. 6 steps left, the current one in buffer 0
      case 3:
        drain(lA0, lW0, w8{}); drain(lA1, lW1, w4{}); drain(lA2, lW2, w0{});
        break;
      case 4:
        steady(lA0, lW0, lA3, lW3, ks);
        drain(lA1, lW1, w8{}); drain(lA2, lW2, w4{}); drain(lA3, lW3, w0{});
        break;
      case 5:
        steady(lA0, lW0, lA3, lW3, ks); steady(lA1, lW1, lA0, lW0, ks + 1);
        drain(lA2, lW2, w8{}); drain(lA3, lW3, w4{}); drain(lA0, lW0, w0{});
        break;
      default:
        steady(lA0, lW0, lA3, lW3, ks); steady(lA1, lW1, lA0, lW0, ks + 1); steady(lA2, lW2, lA1, lW1, ks + 2);
        drain(lA3, lW3, w8{}); drain(lA0, lW0, w4{}); drain(lA1, lW1, w0{});
        break;
    }
  } else {
    // three stages, two of them in flight while the third is multiplied: with one workgroup per CU (the 256 x 128 tile
    // fills the LDS) a single stage in flight does not cover the load latency (~2 us under load).  One barrier per
    // step: behind it every wave's share of stage ks has landed (each waited for its own DMAs, all but the youngest
    // stage's) and every wave has finished multiplying stage ks - 1, whose buffer the DMAs of stage ks + 2 now overwrite.
    // A bare s_barrier: __syncthreads() carries a fence that would drain the DMAs in flight.
    static_assert(ABLK + WBLK == 12 || ABLK + WBLK == 8 || ABLK + WBLK == 4 || NS == 2, "vmcnt immediates below");
    auto phase = [&](const T* cA, const T* cW, T* nA, T* nW, int ks) {
      // s_waitcnt through the builtin (the compiler's own wait-count bookkeeping sees it; an asm wait it would follow with
      // a vmcnt(0) of its own before the first LDS read).  simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 0 << 8.
      if (ks + 1 < nk) {
        if constexpr (ABLK + WBLK == 12) __builtin_amdgcn_s_waitcnt(0x007C);   // vmcnt(12) lgkmcnt(0)
        else if constexpr (ABLK + WBLK == 8) __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0074);                               // vmcnt(4) lgkmcnt(0)
      } else {
        __builtin_amdgcn_s_waitcnt(0x0070);                                    // vmcnt(0) lgkmcnt(0)
      }
      __builtin_amdgcn_s_barrier();
      if (ks + 2 < nk) issue(nA, nW, (ks + 2) * BK);
      compute(cA, cW);
    };
    issue(lA0, lW0, 0);
    issue(lA1, lW1, BK);           // K >= 128: at least two steps
    // steady state: a step whose two successors exist -- no conditions, so that the compiler's wait-count bookkeeping
    // stays exact across the loop (a conditional wait or issue makes it fall back to vmcnt(0) before the LDS reads)
    auto steady = [&](const T* cA, const T* cW, T* nA, T* nW, int ks) {
      if constexpr (ABLK + WBLK == 12) __builtin_amdgcn_s_waitcnt(0x007C);
      else if constexpr (ABLK + WBLK == 8) __builtin_amdgcn_s_waitcnt(0x0078);
      else __builtin_amdgcn_s_waitcnt(0x0074);
      __builtin_amdgcn_s_barrier();
      issue(nA, nW, (ks + 2) * BK);
      compute(cA, cW);
    };
    int ks = 0;
    for (; ks + 5 <= nk; ks += 3) {      // buffers named statically: the compiler tells a read of one from a DMA into another
      steady(lA0, lW0, lA2, lW2, ks);
      steady(lA1, lW1, lA0, lW0, ks + 1);
      steady(lA2, lW2, lA1, lW1, ks + 2);
    }
    // two to four steps are left
    phase(lA0, lW0, lA2, lW2, ks);
    phase(lA1, lW1, lA0, lW0, ks + 1);
    if (ks + 2 < nk) phase(lA2, lW2, lA1, lW1, ks + 2);
    if (ks + 3 < nk) phase(lA0, lW0, lA2, lW2, ks + 3);
  }

  // epilogue: lane holds, per 16 x 16 block, row m = r16 of the activations and columns 4 * kq .. + 3 of the weights
  const T* __restrict__ bias = static_cast<const T*>(P.bias);
  if constexpr (CELL) {
    const int H = cell.H;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int m = m0 + wm * WM + b * 16 + r16;
      if (m >= M) continue;
      const int64_t rin = (int64_t)(cell.slot_in[m] + 1) * H, rout = (int64_t)(cell.slot_out[m] + 1) * H;
      T* xn = cell.x_next + (int64_t)m * cell.ldx;
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        const int n = n0 + wn * WN + a * 16 + kq * 4, j = n >> 2;   // columns n .. n+3 = gates i, f, g, o of unit j
        using v4 = __attribute__((ext_vector_type(4))) T;
        const v4 bv = *reinterpret_cast<const v4*>(bias + n);
        const float zi = acc[a][b][0] + static_cast<float>(bv[0]), zf = acc[a][b][1] + static_cast<float>(bv[1]);
        const float zg = acc[a][b][2] + static_cast<float>(bv[2]), zo = acc[a][b][3] + static_cast<float>(bv[3]);
        const float gi = 1.f / (1.f + expf(-zi)), gf = 1.f / (1.f + expf(-zf));
        const float gg = tanhf(zg), go = 1.f / (1.f + expf(-zo));
        const float c = gi * gg + gf * cell.c_pool[rin + j];
        const T h = static_cast<T>(go * tanhf(c));
        cell.c_pool[rout + j] = c;
        cell.h_pool[rout + j] = h;
        xn[j] = h;
        if (cell.h_pool_next) xn[H + j] = cell.h_pool_next[rin + j];
      }
    }
    return;
  }
  T* __restrict__ Cp = static_cast<T*>(P.c);
  const int c_nseg = P.c_nseg;
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int m = m0 + wm * WM + b * 16 + r16;
    if (m >= M) continue;
    T* crow = Cp + (int64_t)(m / P.c_inner) * P.c_stride_outer + (int64_t)(m % P.c_inner) * P.c_stride_inner;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      const int n = n0 + wn * WN + a * 16 + kq * 4;
      using v4 = __attribute__((ext_vector_type(4))) T;
      v4 o;
      if (bias) {
        const v4 bv = *reinterpret_cast<const v4*>(bias + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = static_cast<T>(acc[a][b][j] + static_cast<float>(bv[j]));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = static_cast<T>(acc[a][b][j]);
      }
      const int seg = n / c_nseg;
      *reinterpret_cast<v4*>(crow + (int64_t)seg * P.c_stride_seg + (n - seg * c_nseg)) = o;
    }
  }
}

template <typename T, int BM, int BN, int NS, int NW, int KS = 1>
int launch_proj(const caiman_proj_problem_t* problems, int n, hipStream_t s) {
  ProjBatch pb;
  pb.n = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    pb.p[i] = problems[i];
    pb.tile_begin[i] = tiles;
    pb.tiles_n[i] = problems[i].N / BN;
    tiles += ((problems[i].M + BM - 1) / BM) * pb.tiles_n[i];
  }
  for (int i = n; i <= kProjMax; ++i) pb.tile_begin[i] = tiles;
  for (int i = n; i < kProjMax; ++i) { pb.p[i] = problems[0]; pb.tiles_n[i] = 1; }
  if (tiles == 0) return CAIMAN_OK;
  hipLaunchKernelGGL((proj_gemm_kernel<T, BM, BN, NS, NW, KS>), dim3((unsigned)tiles), dim3(64 * NW * KS), 0, s, pb, CellArgs<T>{});
  return check_launch("projection GEMM");
}

template <typename T>
int launch_cell_gemm(const caiman_proj_problem_t& prob, const CellArgs<T>& cell, hipStream_t s) {
  constexpr int BM = 128, BN = 128;
  ProjBatch pb;
  pb.n = 1;
  pb.p[0] = prob;
  pb.tile_begin[0] = 0;
  pb.tiles_n[0] = prob.N / BN;
  const int tiles = ((prob.M + BM - 1) / BM) * pb.tiles_n[0];
  for (int i = 1; i <= kProjMax; ++i) pb.tile_begin[i] = tiles;
  for (int i = 1; i < kProjMax; ++i) { pb.p[i] = prob; pb.tiles_n[i] = 1; }
  hipLaunchKernelGGL((proj_gemm_kernel<T, BM, BN, 2, 8, 1, true>), dim3((unsigned)tiles), dim3(64 * 8), 0, s, pb, cell);
  return check_launch("LSTM step GEMM");
}

}  // namespace
}  // namespace caiman

extern "C" int caiman_proj_gemm_supported(const caiman_proj_problem_t* p, int dtype) {
  if (!p || (dtype != CAIMAN_BF16 && dtype != CAIMAN_F16)) return 0;
  auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; };
  if (p->M < 1 || p->N < 128 || p->N % 128 || p->K < 128 || p->K % 128) return 0;
  if (p->a_inner < 1 || p->c_inner < 1 || p->a_kseg < 64 || p->a_kseg % 64 || p->K % p->a_kseg) return 0;
  if (p->c_nseg < 16 || p->c_nseg % 16 || p->N % p->c_nseg) return 0;
  if (!al(p->a, 16) || !al(p->w, 16) || !al(p->c, 8) || (p->bias && !al(p->bias, 8))) return 0;
  if ((p->a_stride_outer | p->a_stride_inner | p->a_stride_seg) & 7) return 0;     // 16-byte pieces stay aligned
  if ((p->c_stride_outer | p->c_stride_inner | p->c_stride_seg) & 3) return 0;     // 8-byte stores
  return 1;
}

extern "C" int caiman_proj_gemm(const caiman_proj_problem_t* problems, int n, int dtype, int tile, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(problems && n >= 1 && n <= kProjMax, "caiman_proj_gemm: 1 .. %d problems per call", kProjMax);
  for (int i = 0; i < n; ++i)
    CAIMAN_CHECK(caiman_proj_gemm_supported(&problems[i], dtype), "caiman_proj_gemm: problem %d is outside the kernel's geometry "
                 "(bf16 / f16; N, K %% 128 == 0; a_kseg %% 64 == 0; c_nseg %% 16 == 0; 16-byte aligned rows)", i);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // measured on the ticks of the base encoder (tools/proj_gemm_bench.py, forward / backward tick): 128 x 128 tiles with
  // 8 waves (wave tile 32 x 64, two workgroups = 16 waves per CU) 46 / 54 us; with 4 waves 54 / 60; 256 x 128 tiles, one
  // workgroup per CU: 67 / 82 (4 waves, two LDS stages), 63 / 68 (three stages), 52 / 73 (8 waves).  A K step of one
  // workgroup takes 0.55 - 1 us whatever the tile (DMA issue -> LDS reads -> MFMAs -> wait for the next stage is one
  // serial chain per wave): more waves per CU overlap more of it, fewer operand bytes per flop do not help; 64 x 128
  // tiles (three workgroups per CU) land on the same 53 us.  Round 3: 128 x 128 tiles, 8 waves, one workgroup per CU with a
  // ring of three (tile 9) or four (tile 10) stages: 53.4 / 45.4 and 51.0 / 44.8 us; in the training step, where the
  // operands are cold, the four-stage ring on the backward ticks is worth 0.35 ms per step against the K split (tile 8).
  if (tile == 0) tile = 5;
  if (tile == 10)  // the four-stage ring needs four 64-deep steps
    for (int i = 0; i < n; ++i)
      if (problems[i].K < 256) tile = 5;
  if (tile == 8)   // the in-workgroup K split needs an even number of 64-deep steps per half
    for (int i = 0; i < n; ++i)
      if (problems[i].K % 256 != 0 || problems[i].a_kseg % 128 != 0) tile = 5;
  if (dtype == CAIMAN_BF16)
    return tile == 1 ? launch_proj<bf16_t, 256, 128, 2, 4>(problems, n, s)
         : tile == 3 ? launch_proj<bf16_t, 256, 128, 3, 4>(problems, n, s)
         : tile == 4 ? launch_proj<bf16_t, 256, 128, 2, 8>(problems, n, s)
         : tile == 5 ? launch_proj<bf16_t, 128, 128, 2, 8>(problems, n, s)
         : tile == 8 ? launch_proj<bf16_t, 128, 128, 2, 8, 2>(problems, n, s)
         : tile == 9 ? launch_proj<bf16_t, 128, 128, 3, 8>(problems, n, s)
         : tile == 10 ? launch_proj<bf16_t, 128, 128, 4, 8>(problems, n, s) : launch_proj<bf16_t, 128, 128, 2, 4>(problems, n, s);
  return tile == 1 ? launch_proj<f16_t, 256, 128, 2, 4>(problems, n, s)
       : tile == 3 ? launch_proj<f16_t, 256, 128, 3, 4>(problems, n, s)
       : tile == 4 ? launch_proj<f16_t, 256, 128, 2, 8>(problems, n, s)
       : tile == 5 ? launch_proj<f16_t, 128, 128, 2, 8>(problems, n, s)
       : tile == 8 ? launch_proj<f16_t, 128, 128, 2, 8, 2>(problems, n, s)
       : tile == 9 ? launch_proj<f16_t, 128, 128, 3, 8>(problems, n, s)
         : tile == 10 ? launch_proj<f16_t, 128, 128, 4, 8>(problems, n, s) : launch_proj<f16_t, 128, 128, 2, 4>(problems, n, s);
}

// One timestep of one LSTM layer for n rows as ONE launch (CellArgs above).  X [n][ldx_in] holds [x_t | h_{t-1}] rows,
// K = the used width (multiple of 128; zero-pad x and the matching columns of W), W [4 hidden][K] with rows ordered
// [unit][gate], bias [4 hidden] in the same order.
extern "C" int caiman_lstm_step_gemm(const void* X, int64_t ldx_in, const void* W, const void* bias, int64_t n, int64_t hidden,
                                     int64_t K, float* c_pool_l, void* h_pool_l, const void* h_pool_next,
                                     const int32_t* slot_in, const int32_t* slot_out, void* X_next, int64_t ldx, int dtype,
                                     caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(dtype == CAIMAN_BF16 || dtype == CAIMAN_F16, "lstm_step_gemm: bf16 / f16 only");
  CAIMAN_CHECK(n >= 0 && n <= 0x7fffffffLL && hidden >= 32 && hidden % 32 == 0 && K >= 128 && K % 128 == 0 && ldx_in >= K &&
                   ldx_in % 8 == 0 && ldx >= (h_pool_next ? 2 * hidden : hidden) && 4 * hidden <= 0x7fffffffLL,
               "lstm_step_gemm: bad extents (hidden %% 32, K %% 128, ldx_in %% 8)");
  // the kernel's per-lane operand offsets are 32-bit byte offsets from the first row
  CAIMAN_CHECK(n * ldx_in * 2 < ((int64_t)1 << 32) && n * ldx * 2 < ((int64_t)1 << 32),
               "lstm_step_gemm: %lld rows of %lld elements exceed the kernel's 32-bit row offsets", (long long)n, (long long)ldx_in);
  if (n == 0) return CAIMAN_OK;
  CAIMAN_CHECK(X && W && bias && c_pool_l && h_pool_l && slot_in && slot_out && X_next, "lstm_step_gemm: null pointer");
  auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; };
  CAIMAN_CHECK(al(X, 16) && al(W, 16) && al(bias, 8), "lstm_step_gemm: X / W 16-byte, bias 8-byte aligned");
  caiman_proj_problem_t p{};
  p.a = X; p.w = W; p.bias = bias; p.c = X_next;
  p.M = (int32_t)n; p.N = (int32_t)(4 * hidden); p.K = (int32_t)K;
  p.a_inner = (int32_t)n; p.a_kseg = (int32_t)K; p.c_inner = (int32_t)n; p.c_nseg = (int32_t)(4 * hidden);
  p.a_stride_outer = 0; p.a_stride_inner = ldx_in; p.a_stride_seg = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == CAIMAN_BF16) {
    CellArgs<bf16_t> c{c_pool_l, static_cast<bf16_t*>(h_pool_l), static_cast<const bf16_t*>(h_pool_next), slot_in, slot_out,
                       static_cast<bf16_t*>(X_next), ldx, (int)hidden};
    return launch_cell_gemm<bf16_t>(p, c, s);
  }
  CellArgs<f16_t> c{c_pool_l, static_cast<f16_t*>(h_pool_l), static_cast<const f16_t*>(h_pool_next), slot_in, slot_out,
                    static_cast<f16_t*>(X_next), ldx, (int)hidden};
  return launch_cell_gemm<f16_t>(p, c, s);
}
