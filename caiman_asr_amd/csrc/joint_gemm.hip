// The joint projection of the transducer as a hand-written MFMA GEMM with the log-sum-exp of every output row folded
// into its epilogue (SURVEY section 8 rows a13 + a14).
//
// Reference: `self.joint_fc(h)` = torch.nn.Linear(joint_n_hid, n_classes) on the packed joint activations
// (training/caiman_asr_train/rnnt/model.py:409-439), followed, inside the loss, by a log-sum-exp over every row of the
// logits it produced (training/lib/csrc/logsumexp.cu:65-105; call site training/lib/src/rnnt_ext/transducer/loss.py).
// At LibriSpeech shapes that is C[304 000, 8704] = A[304 000, 768] · W[8704, 768]^T: 4 TFLOP, a 5.3 GB bf16 result, and a
// second full read of those 5.3 GB for the row normalisers.  Here:
//
//   C[m][n] = bf16( sum_k A[m][k] W[n][k] + bias[n] ),   both operands K-contiguous, fp32 accumulation
//   pmax[m][p], psum[m][p] = max / sum exp(. - max) over the 64 columns p of row m AS STORED (rounded to 16 bits)
//
// so the normaliser of a row becomes a reduction over N / 64 partial pairs (lse_partials_kernel) instead of a pass over
// the logits.
//
// Geometry: one workgroup of 8 waves (2 along M x 4 along N) per 256 x 256 tile, wave tile 128 x 64 = 32 MFMA blocks of
// 16 x 16 (v_mfma_f32_16x16x32, 128 accumulator registers).  K is walked in steps of 32 through a RING OF FOUR LDS stages
// (4 x (256 + 256) rows x 64 B = 128 KB): both operand tiles arrive by LDS-DMA (global_load_lds_dwordx4, no staging
// registers), THREE stages are in flight while the fourth is multiplied, and a step costs one counted wait + one bare
// barrier:   s_waitcnt vmcnt(8)  ->  s_barrier  ->  issue stage s + 3  ->  12 ds_read_b128 + 32 MFMA.
// Behind the barrier every wave's share of stage s has landed (each waited for its own DMAs: all but the two youngest
// stages') and every wave has finished reading stage s - 1, whose buffer the DMAs of stage s + 3 overwrite.  The four
// buffers are separate LDS objects named statically (loop unrolled by four), so the compiler can tell the buffer being read
// from the ones being filled and does not drain the DMAs in front of the LDS reads (csrc/proj_gemm.hip, three-stage path,
// is the same idea with two stages in flight).
// LDS image: a row of a stage is 32 K-values = 64 B = four 16-byte pieces; a DMA instruction writes 16 rows lane-linear,
// and piece q of row r is FETCHED from position q ^ ((r >> 2) & 3) of the source row, so that the 16 rows x one piece of a
// fragment read (ds_read_b128, 16 lanes at a time) fall on 16 different bank groups.
// The MFMA is issued transposed (A operand = 16 rows of W, B operand = 16 rows of the activations): a lane ends up with
// FOUR CONSECUTIVE COLUMNS of one output row -- 8-byte stores, and the row-wise max / sum of the epilogue needs two
// shuffles (the four lanes that share a row) instead of a transpose.
// Workgroups are dealt to tiles through the XCD remap of the guide (T1, bijective form): the eight XCDs each get a
// contiguous run of tiles, M-tile major, so the 34 workgroups that share an activation panel share an L2.
#include <algorithm>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "common.h"

namespace caiman {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
template <typename T>
struct jfrag {
  using type = __attribute__((ext_vector_type(8))) T;
};
__device__ __forceinline__ f32x4 jmfma(jfrag<bf16_t>::type a, jfrag<bf16_t>::type b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 jmfma(jfrag<f16_t>::type a, jfrag<f16_t>::type b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// A register swapped against a copy of itself: the pair holds, in every lane, the value of the lane's own half (lanes 0-31 /
// 32-63 for swap32, even / odd 16-lane rows for swap16) and that of the other half -- an all-reduce step without LDS.
// (The two results are copied out of a NON-const vector into plain integers first: __builtin_bit_cast on an element of a
// const result vector made hipcc, ROCm 7.2, read element 0 twice.)
struct jpair {
  float a, b;
};
__device__ __forceinline__ jpair jswap32(float x) {
  const unsigned a = __builtin_bit_cast(unsigned, x);
  auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return jpair{__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1)};
}
__device__ __forceinline__ jpair jswap16(float x) {
  const unsigned a = __builtin_bit_cast(unsigned, x);
  auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return jpair{__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1)};
}

constexpr int JBM = 256, JBN = 256, JBK = 32, JNW = 8;
constexpr int JTM = 8, JTN = 4;   // 16 x 16 blocks of a wave tile: 128 rows, 64 columns

template <typename T, bool LSE, bool PRIO>
__global__ __launch_bounds__(64 * JNW, 1) void joint_fc_gemm_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                                   const T* __restrict__ bias, T* __restrict__ C,
                                                                   float* __restrict__ pmax, float* __restrict__ psum, int M,
                                                                   int N, int K, int tiles_n, int tiles_m, int group) {
  using frag = typename jfrag<T>::type;
  __shared__ __attribute__((aligned(1024))) T sA0[JBM * JBK], sA1[JBM * JBK], sA2[JBM * JBK], sA3[JBM * JBK];
  __shared__ __attribute__((aligned(1024))) T sW0[JBN * JBK], sW1[JBN * JBK], sW2[JBN * JBK], sW3[JBN * JBK];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // tile of this workgroup: XCD remap (blocks bid, bid + 8, ... share an XCD: give them consecutive tiles)
  int t;
  {
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // tile order: groups of `group` M-tiles; inside a group N-tile major, M-tile minor.  The workgroups an XCD runs at the same
  // time (32 consecutive tiles) then share a few weight panels AND a few activation panels (both stay in the XCD's 4 MB L2),
  // instead of one activation panel and all 34 weight panels (13 MB: served from the Infinity Cache for every M-tile).
  int tm, tn;
  {
    const int per = group * tiles_n, g = t / per, w_ = t - g * per;
    const int gm = min(group, tiles_m - g * group);
    tn = w_ / gm;
    tm = g * group + (w_ - tn * gm);
  }
  const int m0 = tm * JBM, n0 = tn * JBN;

  // per-lane DMA sources as 32-bit byte offsets from wave-uniform bases.  A stage holds 16 blocks of 16 rows per operand;
  // wave w brings blocks w and w + 8; lane l of a block brings the piece that belongs at (row l >> 2, position l & 3).
  unsigned a_off[2], w_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave + JNW * i) * 16 + (lane >> 2);
    const int q = (lane & 3) ^ ((row >> 2) & 3);
    const int m = m0 + row < M ? row : M - 1 - m0;          // rows past M re-read the last row; never stored
    a_off[i] = (unsigned)((m * K + q * 8) * (int)sizeof(T));
    w_off[i] = (unsigned)((row * K + q * 8) * (int)sizeof(T));
  }
  const char* a_base = reinterpret_cast<const char*>(A + (int64_t)m0 * K);
  const char* w_base = reinterpret_cast<const char*>(W + (int64_t)n0 * K);

  auto issue = [&](T* lA, T* lW, int k0) {
    const char* ab = a_base + (int64_t)k0 * (int64_t)sizeof(T);
    const char* wb = w_base + (int64_t)k0 * (int64_t)sizeof(T);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ab + a_off[i]),
                                       (__attribute__((address_space(3))) void*)(lA + (wave + JNW * i) * 16 * JBK), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + w_off[i]),
                                       (__attribute__((address_space(3))) void*)(lW + (wave + JNW * i) * 16 * JBK), 16, 0, 0);
  };

  const int wr = wave >> 2, wc = wave & 3;
  const int r16 = lane & 15, kq = lane >> 4;
  f32x4 acc[JTN][JTM];
#pragma unroll
  for (int a = 0; a < JTN; ++a)
#pragma unroll
    for (int b = 0; b < JTM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a stage (the same for every stage): row * 32 + piece * 8 elements
  int wfo[JTN], afo[JTM];
#pragma unroll
  for (int a = 0; a < JTN; ++a) {
    const int row = wc * 64 + a * 16 + r16;
    wfo[a] = row * JBK + ((kq ^ ((row >> 2) & 3)) * 8);
  }
#pragma unroll
  for (int b = 0; b < JTM; ++b) {
    const int row = wr * 128 + b * 16 + r16;
    afo[b] = row * JBK + ((kq ^ ((row >> 2) & 3)) * 8);
  }

  constexpr bool prio = PRIO;
  auto compute = [&](const T* lA, const T* lW) {
    frag wf[JTN], af[JTM];
#pragma unroll
    for (int a = 0; a < JTN; ++a) wf[a] = *reinterpret_cast<const frag*>(lW + wfo[a]);
#pragma unroll
    for (int b = 0; b < JTM; ++b) af[b] = *reinterpret_cast<const frag*>(lA + afo[b]);
    if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int b = 0; b < JTM; ++b)
#pragma unroll
      for (int a = 0; a < JTN; ++a) acc[a][b] = jmfma(wf[a], af[b], acc[a][b]);
    if (prio) __builtin_amdgcn_s_setprio(0);
  };

  // s_waitcnt through the builtin (the compiler's own wait-count bookkeeping sees it).  simm16 = vmcnt[3:0] | expcnt 7 << 4 |
  // lgkmcnt 0 << 8: this wave's DMAs of the stage about to be read have landed, its LDS reads of the stage before returned.
  auto steady = [&](const T* cA, const T* cW, T* nA, T* nW, int s) {
    __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8) lgkmcnt(0): stages s + 1, s + 2 may still fly
    __builtin_amdgcn_s_barrier();
    issue(nA, nW, (s + 3) * JBK);
    __builtin_amdgcn_sched_barrier(0);     // keep the DMAs at the head of the step (left alone the scheduler sinks them behind the MFMAs)
    compute(cA, cW);
  };
  const int nk = K / JBK;                 // K % 128 == 0: a multiple of four stages, at least four
  issue(sA0, sW0, 0);
  issue(sA1, sW1, JBK);
  issue(sA2, sW2, 2 * JBK);
  int s = 0;
  for (; s + 4 <= nk - 3; s += 4) {       // no conditions inside: the wait counts stay exact across the loop
    steady(sA0, sW0, sA3, sW3, s);
    steady(sA1, sW1, sA0, sW0, s + 1);
    steady(sA2, sW2, sA1, sW1, s + 2);
    steady(sA3, sW3, sA2, sW2, s + 3);
  }
  steady(sA0, sW0, sA3, sW3, s);          // s = nk - 4: the last stage goes into buffer 3
  __builtin_amdgcn_s_waitcnt(0x0078);     // stage nk - 3; nk - 2, nk - 1 in flight
  __builtin_amdgcn_s_barrier();
  compute(sA1, sW1);
  __builtin_amdgcn_s_waitcnt(0x0074);     // vmcnt(4)
  __builtin_amdgcn_s_barrier();
  compute(sA2, sW2);
  __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0)
  __builtin_amdgcn_s_barrier();
  compute(sA3, sW3);

  // epilogue: lane holds, per 16 x 16 block, row m = r16 of the activations and columns 4 * kq .. + 3 of the weights
  using v4 = __attribute__((ext_vector_type(4))) T;
  const int NP = N / 64;
  float bcol[JTN][4];
#pragma unroll
  for (int a = 0; a < JTN; ++a) {
    const int n = n0 + wc * 64 + a * 16 + kq * 4;
    if (bias) {
      const v4 bv = *reinterpret_cast<const v4*>(bias + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) bcol[a][j] = static_cast<float>(bv[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) bcol[a][j] = 0.f;
    }
  }
#pragma unroll
  for (int b = 0; b < JTM; ++b) {
    const int m = m0 + wr * 128 + b * 16 + r16;
    const bool live = m < M;
    T* crow = C + (int64_t)m * N + n0 + wc * 64 + kq * 4;
    float v[JTN][4];
    float mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < JTN; ++a) {
      v4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = static_cast<T>(acc[a][b][j] + bcol[a][j]);
        if constexpr (LSE) {
          v[a][j] = static_cast<float>(o[j]);      // the normaliser is that of the STORED logits
          mx = fmaxf(mx, v[a][j]);
        }
      }
      if (live) *reinterpret_cast<v4*>(crow + a * 16) = o;
    }
    if constexpr (LSE) {
      mx = fmaxf(mx, __shfl_xor(mx, 16, kWave));
      mx = fmaxf(mx, __shfl_xor(mx, 32, kWave));
      // v == mx counts as 1 without going through exp: a piece of -inf only (or +inf) has the sum 1 per maximal element, not
      // exp(inf - inf) = NaN; a NaN element (fmaxf skips it) still makes the sum NaN, as torch.logsumexp would
      float sm = 0.f;
#pragma unroll
      for (int a = 0; a < JTN; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) sm += (v[a][j] == mx) ? 1.f : __expf(v[a][j] - mx);
      sm += __shfl_xor(sm, 16, kWave);
      sm += __shfl_xor(sm, 32, kWave);
      if (live && kq == 0) {
        const int64_t p = (int64_t)m * NP + (n0 >> 6) + wc;
        pmax[p] = mx;
        psum[p] = sm;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 4: the same product on an 8-PHASE main loop with the two waves of every SIMD in OPPOSITE roles.
//
// The ring kernel above runs all eight waves in lockstep: everyone waits, everyone reads 12 fragments, everyone issues 32
// MFMAs -- the matrix pipe idles while both waves of a SIMD read, and the LDS idles while both multiply (43.7 % MFMA busy
// under the profiler).  Here a K tile of 64 is cut into FOUR phases of 16 MFMAs per wave (one 64 x 32 quadrant of the wave's
// 128 x 64 tile over the whole K tile), every phase is  [ds_read fragments | issue one LDS-DMA unit | counted vmcnt]
// s_barrier  [lgkmcnt(0) | 16 MFMA]  s_barrier, and waves 4-7 run ONE BARRIER BEHIND waves 0-3: while one wave of a SIMD
// multiplies, its partner reads and issues, then they swap.  The matrix pipe sees 16 MFMAs from one wave, then 16 from the
// other, back to back.
//
// Staging units (16 KB each = 128 rows x 64 K values; two K tiles of four units = 128 KB of LDS, eight separate LDS objects
// named statically so that the compiler can tell the unit a ds_read touches from the units in flight):
//   type 0  A rows {wr * 128 + 0 .. 63}   (the first 64 rows of BOTH wave-row halves: what phase 0 multiplies)
//   type 1  W rows {wc * 64 + 0 .. 31}    (phase 0 and, from registers, phase 3)
//   type 2  W rows {wc * 64 + 32 .. 63}   (phases 1, 2)
//   type 3  A rows {wr * 128 + 64 .. 127} (phases 2, 3)
// Unit u = 4 * tile + type is issued at phase u - 6 by all eight waves (two 1 KB LDS-DMA instructions each), every phase ends
// its load part with vmcnt(8) -- the four youngest units may still fly, unit <= phase + 2 has landed -- and unit u is first
// read at phase u - 1 or u: one phase AFTER the wait that retires it, as the guide's placement rule demands for waves
// staggered by a barrier.  Unit u + 8 overwrites unit u at phase u + 2, two phases after the last read of u (type 0 is read
// at phase u).  Rows are 128 B; 16-byte chunk c of row r sits at position c ^ ((r >> 1) & 7) (swizzle applied to the DMA's
// SOURCE address and to the read address): the four 16-lane groups of a ds_read_b128 each hit 16 distinct bank groups.
template <typename T, bool LSE, bool BIAS>
__global__ __launch_bounds__(512, 2) void joint_fc_gemm8_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                                const T* __restrict__ bias, T* __restrict__ C,
                                                                float* __restrict__ pmax, float* __restrict__ psum, int M,
                                                                int N, int K, int tiles_n, int tiles_m, int group) {
  using frag = typename jfrag<T>::type;
  constexpr int UE = 128 * 64;   // elements of a unit
  __shared__ __attribute__((aligned(1024))) T u00[UE], u01[UE], u02[UE], u03[UE], u10[UE], u11[UE], u12[UE], u13[UE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
  int t;
  {
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tm, tn;
  {
    const int per = group * tiles_n, g = t / per, w_ = t - g * per;
    const int gm = min(group, tiles_m - g * group);
    tn = w_ / gm;
    tm = g * group + (w_ - tn * gm);
  }
  const int m0 = tm * JBM, n0 = tn * JBN;
  const int wr = wave >> 2, wc = wave & 3;
  const int r16 = lane & 15, kq = lane >> 4;

  // DMA sources: per unit type two instructions per wave; instruction i of wave w fills unit rows (2 w + i) * 8 .. + 7,
  // lane l the chunk at (row l >> 3, position l & 7)
  unsigned src[4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rho = (wave * 2 + i) * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ ((rho >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int ra = (rho >> 6) * 128 + h * 64 + (rho & 63);
      if (m0 + ra >= M) ra = M - 1 - m0;                       // rows past M re-read the last row; never stored
      const int rw = (rho >> 5) * 64 + h * 32 + (rho & 31);
      src[h ? 3 : 0][i] = (unsigned)((ra * K + ch * 8) * (int)sizeof(T));
      src[h ? 2 : 1][i] = (unsigned)((rw * K + ch * 8) * (int)sizeof(T));
    }
  }
  const char* a_base = reinterpret_cast<const char*>(A + (int64_t)m0 * K);
  const char* w_base = reinterpret_cast<const char*>(W + (int64_t)n0 * K);

#define CAIMAN_UNIT(PAR, TY) \
  ((PAR) == 0 ? ((TY) == 0 ? u00 : (TY) == 1 ? u01 : (TY) == 2 ? u02 : u03) : ((TY) == 0 ? u10 : (TY) == 1 ? u11 : (TY) == 2 ? u12 : u13))

  // ga / gw: wave-uniform pointers to column (first K tile of the pair being multiplied) of the tile's first A / W row
  const char* ga = a_base;
  const char* gw = w_base;
  auto issue = [&](auto PAR_, auto TY_, auto AHEAD_) {
    constexpr int PAR = decltype(PAR_)::value, TY = decltype(TY_)::value, AHEAD = decltype(AHEAD_)::value;
    T* dst = CAIMAN_UNIT(PAR, TY);
    // (the instruction's immediate offset is no place for the K advance: the hardware adds it to the LDS address as well)
    const char* gb = ((TY == 0 || TY == 3) ? ga : gw) + AHEAD * 64 * (int)sizeof(T);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + src[TY][i]),
                                       (__attribute__((address_space(3))) void*)(dst + (wave * 2 + i) * 8 * 64), 16, 0, 0);
  };

  // fragment read offsets inside a unit (elements): k-step 0 and 1 differ by an XOR on the chunk, hence two bases per operand
  const int swz = (r16 >> 1) & 7;
  int abase[2], wbase[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    abase[ks] = (wr * 64 + r16) * 64 + (((ks * 4 + kq) ^ swz) * 8);
    wbase[ks] = (wc * 32 + r16) * 64 + (((ks * 4 + kq) ^ swz) * 8);
  }

  // accumulators [mh][b][nh][a]: rows m0 + wr * 128 + mh * 64 + b * 16 + r16, columns n0 + wc * 64 + nh * 32 + a * 16 + kq * 4 + j.
  // They start at the bias (the lane's four columns), so the epilogue has no bias pass.  The four bias loads are the oldest
  // vector-memory operations of the wave: the first counted wait retires them with the first units.
  using v4 = __attribute__((ext_vector_type(4))) T;
  f32x4 acc[2][4][2][2];
  v4 braw[4];
  if constexpr (BIAS) {
#pragma unroll
    for (int c = 0; c < 4; ++c) braw[c] = *reinterpret_cast<const v4*>(bias + n0 + wc * 64 + c * 16 + kq * 4);
  }
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  frag af[4][2], wf[2][2][2];
  // one phase.  PAR: parity of the K tile being multiplied; PH: phase 0..3; ISSUE: stage unit (phase + 6) of K tile kt;
  // VM: the counted wait behind it (-1: none)
  auto phase = [&](auto PAR_, auto PH_, auto ISSUE_, auto VM_) {
    constexpr int PAR = decltype(PAR_)::value, PH = decltype(PH_)::value, VM = decltype(VM_)::value;
    constexpr bool ISSUE = decltype(ISSUE_)::value != 0;
    // ---- load part
    if constexpr (PH == 0) {
      const T* ua = CAIMAN_UNIT(PAR, 0);
      const T* uw = CAIMAN_UNIT(PAR, 1);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wf[0][a][ks] = *reinterpret_cast<const frag*>(uw + wbase[ks] + a * 16 * 64);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[b][ks] = *reinterpret_cast<const frag*>(ua + abase[ks] + b * 16 * 64);
    } else if constexpr (PH == 1) {
      const T* uw = CAIMAN_UNIT(PAR, 2);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wf[1][a][ks] = *reinterpret_cast<const frag*>(uw + wbase[ks] + a * 16 * 64);
    } else if constexpr (PH == 2) {
      const T* ua = CAIMAN_UNIT(PAR, 3);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[b][ks] = *reinterpret_cast<const frag*>(ua + abase[ks] + b * 16 * 64);
    }
    if constexpr (ISSUE) {
      // unit phase + 6: phases 0, 1 stage types 2, 3 of the NEXT K tile (other parity), phases 2, 3 types 0, 1 of the one after
      // (ga / gw point at K tile kt - PAR: the loop advances them once per pair of tiles)
      if constexpr (PH == 0) issue(std::integral_constant<int, PAR ^ 1>{}, I2{}, std::integral_constant<int, PAR + 1>{});
      if constexpr (PH == 1) issue(std::integral_constant<int, PAR ^ 1>{}, I3{}, std::integral_constant<int, PAR + 1>{});
      if constexpr (PH == 2) issue(std::integral_constant<int, PAR>{}, I0{}, std::integral_constant<int, PAR + 2>{});
      if constexpr (PH == 3) issue(std::integral_constant<int, PAR>{}, I1{}, std::integral_constant<int, PAR + 2>{});
    }
    if constexpr (VM >= 0) __builtin_amdgcn_s_waitcnt(0x0F70 | VM);       // vmcnt(VM), lgkmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);                                  // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
    // ---- multiply part: quadrant (mh, nh) = (0,0) (0,1) (1,1) (1,0)
    constexpr int mh = PH >> 1, nh = (PH == 1 || PH == 2) ? 1 : 0;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[mh][b][nh][a] = jmfma(wf[nh][a][ks], af[b][ks], acc[mh][b][nh][a]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using VN = std::integral_constant<int, -1>;
  using V8 = std::integral_constant<int, 8>;
  using V6 = std::integral_constant<int, 6>;
  using V4 = std::integral_constant<int, 4>;
  using V2 = std::integral_constant<int, 2>;
  using V0 = std::integral_constant<int, 0>;

  const int nk = K / 64;                  // K % 128 == 0: an even number of K tiles, at least two
  issue(I0{}, I0{}, I0{});
  issue(I0{}, I1{}, I0{});
  issue(I0{}, I2{}, I0{});
  issue(I0{}, I3{}, I0{});
  issue(I1{}, I0{}, I1{});
  issue(I1{}, I1{}, I1{});
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (BIAS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = static_cast<float>(braw[c][j]);
    }
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[mh][b][c >> 1][c & 1] = bv;
  }
  __builtin_amdgcn_s_waitcnt(0x0F78);     // vmcnt(8): units 0, 1 (and the bias) have landed
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind from here on
  int kt = 0;
  for (; kt < nk - 2; kt += 2) {
    phase(I0{}, I0{}, I1{}, V8{});
    phase(I0{}, I1{}, I1{}, V8{});
    phase(I0{}, I2{}, I1{}, V8{});
    phase(I0{}, I3{}, I1{}, V8{});
    phase(I1{}, I0{}, I1{}, V8{});
    phase(I1{}, I1{}, I1{}, V8{});
    phase(I1{}, I2{}, I1{}, V8{});
    phase(I1{}, I3{}, I1{}, V8{});
    ga += 2 * 64 * sizeof(T);
    gw += 2 * 64 * sizeof(T);
  }
  // the last two K tiles: only the last two units are still to be issued
  phase(I0{}, I0{}, I1{}, V8{});
  phase(I0{}, I1{}, I1{}, V8{});
  phase(I0{}, I2{}, I0{}, V6{});
  phase(I0{}, I3{}, I0{}, V4{});
  phase(I1{}, I0{}, I0{}, V2{});
  phase(I1{}, I1{}, I0{}, V0{});
  phase(I1{}, I2{}, I0{}, VN{});
  phase(I1{}, I3{}, I0{}, VN{});
  if (wr == 0) __builtin_amdgcn_s_barrier();   // waves 0-3 meet the last barrier of waves 4-7
#undef CAIMAN_UNIT

  // epilogue.  Per block of 16 rows a lane holds, for its row r16, the four-column packets (c, kq): columns c * 16 + kq * 4 .. + 3.
  // Two register swaps between lane groups (v_permlane32_swap, v_permlane16_swap: a 4 x 4 transpose of packets between the
  // register index c and the lane group kq) leave it with columns kq * 16 .. + 15 -- 32 contiguous bytes, two 16-byte stores, and
  // the four lanes of a row cover one full 128-byte line.  The row's (max, sum exp) over the wave's 64 columns is taken on the
  // values AS STORED; the reference point is the row maximum clamped to +-2e38, so that rows of -inf / +inf need no per-element
  // special case (exp2(-inf) = 0; +inf - finite = +inf).
  using u2 = __attribute__((ext_vector_type(2))) unsigned;
  using u4 = __attribute__((ext_vector_type(4))) unsigned;
  const int NP = N / 64;
  constexpr float kLog2e = 1.4426950408889634f;
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int m = m0 + wr * 128 + mh * 64 + b * 16 + r16;
      const bool live = m < M;
      u2 P[4];
      float v[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = static_cast<T>(acc[mh][b][c >> 1][c & 1][j]);
          if constexpr (LSE) v[c][j] = static_cast<float>(o[j]);
        }
        P[c] = __builtin_bit_cast(u2, o);
      }
      if constexpr (LSE) {
        float mx = fmaxf(fmaxf(v[0][0], v[0][1]), fmaxf(v[0][2], v[0][3]));
#pragma unroll
        for (int c = 1; c < 4; ++c) mx = fmaxf(fmaxf(mx, fmaxf(v[c][0], v[c][1])), fmaxf(v[c][2], v[c][3]));
        {                                               // over the four lanes (kq) that share the row
          const jpair h = jswap32(mx);
          mx = fmaxf(h.a, h.b);
          const jpair q = jswap16(mx);
          mx = fmaxf(q.a, q.b);
        }
        mx = fminf(fmaxf(mx, -2e38f), 2e38f);
        const float nmx = -mx * kLog2e;
        float sm = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < 4; ++j) sm += __builtin_amdgcn_exp2f(__builtin_fmaf(v[c][j], kLog2e, nmx));
        {
          const jpair h = jswap32(sm);
          sm = h.a + h.b;
          const jpair q = jswap16(sm);
          sm = q.a + q.b;
        }
        if (live && kq == 0) {
          const int64_t p = (int64_t)m * NP + (n0 >> 6) + wc;
          pmax[p] = mx;
          psum[p] = sm;
        }
      }
      // packets (c, kq) -> (kq, c)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        auto s02 = __builtin_amdgcn_permlane32_swap(P[0][d], P[2][d], false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(P[1][d], P[3][d], false, false);
        auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        P[0][d] = t01[0];
        P[1][d] = t01[1];
        P[2][d] = t23[0];
        P[3][d] = t23[1];
      }
      if (live) {
        T* crow = C + (int64_t)m * N + n0 + wc * 64 + kq * 16;
        *reinterpret_cast<u4*>(crow) = u4{P[0][0], P[0][1], P[1][0], P[1][1]};
        *reinterpret_cast<u4*>(crow + 8) = u4{P[2][0], P[2][1], P[3][0], P[3][1]};
      }
    }
}

// row normaliser from the partial pairs: lse = M + log(sum_p psum_p * exp(pmax_p - M)), M = max_p pmax_p.  One wave per row.
__global__ __launch_bounds__(256) void lse_partials_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                          float* __restrict__ lse, int64_t rows, int NP) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* pm = pmax + row * NP;
  const float* ps = psum + row * NP;
  float mx = -INFINITY;
  for (int p = lane; p < NP; p += 64) mx = fmaxf(mx, pm[p]);
  mx = wave_reduce(mx, [](float a, float b) { return fmaxf(a, b); });
  float sm = 0.f;
  for (int p = lane; p < NP; p += 64) {
    const float m = pm[p];
    sm += (m == mx) ? ps[p] : ps[p] * __expf(m - mx);     // m == mx: also the all -inf / +inf rows (sum >= 1, no inf - inf)
  }
  sm = wave_reduce(sm, [](float a, float b) { return a + b; });
  if (lane == 0) lse[row] = mx + logf(sm);
}

template <typename T>
int launch_joint_fc(const T* A, const T* W, const T* bias, T* C, float* lse, float* ws, int64_t M, int64_t N, int64_t K,
                    hipStream_t s) {
  const int tiles_n = (int)(N / JBN), tiles_m = (int)((M + JBM - 1) / JBM);
  const int64_t tiles = (int64_t)tiles_m * tiles_n;
  const int NP = (int)(N / 64);
  // measurement knobs (tools/joint_gemm_bench.py): M-tiles per group of the tile order, s_setprio around the MFMA clusters
  // measured (tools/joint_gemm_bench.py, 304 000 x 768 x 8704): forward + LSE 5.57 ms M-tile major, 5.18 in groups of 8, 5.57 in
  // groups of 16; the input gradient (N = 768: three N-tiles) 3.63 / 3.76 / 3.91 -- so 8 with the epilogue, 1 without
  static const int group_env = std::getenv("CAIMAN_JOINT_GROUP") ? std::atoi(std::getenv("CAIMAN_JOINT_GROUP")) : 0;
  static const bool prio = std::getenv("CAIMAN_JOINT_PRIO") != nullptr && std::atoi(std::getenv("CAIMAN_JOINT_PRIO")) != 0;
  const int group = std::max(1, std::min(group_env > 0 ? group_env : (lse ? 8 : 1), tiles_m));
  float* pmax = lse ? ws : nullptr;
  float* psum = lse ? ws + M * NP : nullptr;
  // CAIMAN_JOINT_KERNEL=ring: the round-3 four-stage ring kernel (A/B only)
  static const bool ring = std::getenv("CAIMAN_JOINT_KERNEL") != nullptr && std::string(std::getenv("CAIMAN_JOINT_KERNEL")) == "ring";
  if (!ring) {
#define CAIMAN_JGEMM8(L, B)                                                                                               \
  hipLaunchKernelGGL((joint_fc_gemm8_kernel<T, L, B>), dim3((unsigned)tiles), dim3(512), 0, s, A, W, bias, C, pmax, psum, (int)M, \
                     (int)N, (int)K, tiles_n, tiles_m, group)
    if (lse) {
      if (bias) CAIMAN_JGEMM8(true, true); else CAIMAN_JGEMM8(true, false);
      hipLaunchKernelGGL(lse_partials_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, pmax, psum, lse, M, NP);
    } else {
      if (bias) CAIMAN_JGEMM8(false, true); else CAIMAN_JGEMM8(false, false);
    }
#undef CAIMAN_JGEMM8
    return check_launch("joint projection GEMM");
  }
#define CAIMAN_JGEMM(L, P)                                                                                                \
  hipLaunchKernelGGL((joint_fc_gemm_kernel<T, L, P>), dim3((unsigned)tiles), dim3(64 * JNW), 0, s, A, W, bias, C, pmax, psum, \
                     (int)M, (int)N, (int)K, tiles_n, tiles_m, group)
  if (lse) {
    if (prio) CAIMAN_JGEMM(true, true); else CAIMAN_JGEMM(true, false);
    hipLaunchKernelGGL(lse_partials_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, pmax, psum, lse, M, NP);
  } else {
    if (prio) CAIMAN_JGEMM(false, true); else CAIMAN_JGEMM(false, false);
  }
#undef CAIMAN_JGEMM
  return check_launch("joint projection GEMM");
}

}  // namespace
}  // namespace caiman

extern "C" int64_t caiman_joint_fc_workspace_elems(int64_t M, int64_t N) { return 2 * M * (N / 64); }

extern "C" int caiman_joint_fc_supported(int64_t M, int64_t N, int64_t K, int dtype) {
  return (dtype == CAIMAN_BF16 || dtype == CAIMAN_F16) && M >= 1 && N >= 256 && N % 256 == 0 && K >= 128 && K % 128 == 0 &&
                 M * N < ((int64_t)1 << 40) && (M + 255) / 256 * (N / 256) < ((int64_t)1 << 31) && 256 * K * 2 < ((int64_t)1 << 31)
             ? 1 : 0;
}

extern "C" int caiman_joint_fc_forward(const void* A, const void* W, const void* bias, void* C, float* lse, float* workspace,
                                       int64_t M, int64_t N, int64_t K, int dtype, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(caiman_joint_fc_supported(M, N, K, dtype),
               "joint_fc_forward: bf16 / f16, N %% 256 == 0, K %% 128 == 0 (got M %lld N %lld K %lld dtype %d)", (long long)M,
               (long long)N, (long long)K, dtype);
  CAIMAN_CHECK(A && W && C && (lse == nullptr || workspace != nullptr), "joint_fc_forward: null pointer");
  auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; };
  CAIMAN_CHECK(al(A, 16) && al(W, 16) && al(C, 16) && (!bias || al(bias, 8)), "joint_fc_forward: operands and C 16-byte, bias 8-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == CAIMAN_BF16)
    return launch_joint_fc<bf16_t>((const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)bias, (bf16_t*)C, lse, workspace, M, N, K, s);
  return launch_joint_fc<f16_t>((const f16_t*)A, (const f16_t*)W, (const f16_t*)bias, (f16_t*)C, lse, workspace, M, N, K, s);
}
