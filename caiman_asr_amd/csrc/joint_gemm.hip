// The joint projection of the transducer as a hand-written MFMA GEMM with the log-sum-exp of every output row folded
// into its epilogue (SURVEY section 8 rows a13 + a14).
//
// Reference: `self.joint_fc(h)` = torch.nn.Linear(joint_n_hid, n_classes) on the packed joint activations
// (training/caiman_asr_train/rnnt/model.py:409-439), followed, inside the loss, by a log-sum-exp over every row of the
// logits it produced (training/lib/csrc/logsumexp.cu:65-105; call site training/lib/src/rnnt_ext/transducer/loss.py).
// At LibriSpeech shapes that is C[304 000, 8704] = A[304 000, 768] · W[8704, 768]^T: 4 TFLOP, a 5.3 GB bf16 result, and a
// second full read of those 5.3 GB for the row normalisers.  Here:
//
//   C[m][n] = bf16( bias[n] + sum_k A[m][k] W[n][k] ),   both operands K-contiguous, fp32 accumulation
//   pmax[p][m], psum[p][m] = reference point / sum exp(. - reference) over the 64 columns p of row m AS STORED
//
// so the normaliser of a row becomes a reduction over N / 64 partial pairs (lse_partials_kernel) instead of a pass over
// the logits.  The same kernel without bias and epilogue reduction is the projection's input gradient dY · W (the caller
// passes the transposed weight copy).
//
// Geometry: 256 x 256 output tiles, one workgroup of 8 waves (2 along M x 4 along N, wave tile 128 x 64 = 32 MFMA blocks of
// 16 x 16, v_mfma_f32_16x16x32, 128 accumulator registers), K in tiles of 64.
//
// 8-PHASE MAIN LOOP, THE TWO WAVES OF EVERY SIMD IN OPPOSITE ROLES.  A K tile is cut into FOUR phases of 16 MFMAs per wave
// (one 64 x 32 quadrant of the wave tile over the whole K tile); every phase is
//     [ds_read fragments | issue one LDS-DMA unit | counted vmcnt]  s_barrier  [lgkmcnt(0) | 16 MFMA]  s_barrier
// and waves 4-7 run ONE BARRIER BEHIND waves 0-3: while one wave of a SIMD multiplies, its partner reads and issues, then
// they swap.  The matrix pipe sees 16 MFMAs from one wave, then 16 from the other, back to back (the round-3 kernel ran all
// eight waves in lockstep -- everyone reads, then everyone multiplies -- and sat at 43.7 % MFMA busy).
//
// Staging units (16 KB each = 128 rows x 64 K values; two K tiles of four units = 128 KB of LDS, eight separate LDS objects
// named statically so that the compiler can tell the unit a ds_read touches from the units in flight):
//   type 0  A rows {wr * 128 + 0 .. 63}   (the first 64 rows of BOTH wave-row halves: what phase 0 multiplies)
//   type 1  W rows {wc * 64 + 0 .. 31}    (phase 0 and, from registers, phase 3)
//   type 2  W rows {wc * 64 + 32 .. 63}   (phases 1, 2)
//   type 3  A rows {wr * 128 + 64 .. 127} (phases 2, 3)
// Unit u = 4 * tile + type is issued at phase u - 6 by all eight waves (two 1 KB LDS-DMA instructions each), every phase ends
// its load part with vmcnt(6) -- the three youngest units may still fly, unit <= phase + 3 has landed -- and unit u is first
// read at phase u - 2 (type 1: a K tile's first W fragments are read in the read-free last phase of the tile before, which
// evens the fragment reads out to 8 / 4 / 8 / 4 per phase), u - 1 or u: at least one phase AFTER the wait that retires it, as
// waves staggered by a barrier need.  Unit u + 8
// overwrites unit u at phase u + 2, two phases after the last read of u (type 0 is read at phase u).  Rows are 128 B; 16-byte
// chunk c of row r sits at position c ^ ((r >> 1) & 7) (swizzle applied to the DMA's SOURCE address and to the read address):
// the four 16-lane groups of a ds_read_b128 each hit 16 distinct bank groups.
//
// PERSISTENT WORKGROUPS.  A workgroup walks a static list of tiles (one workgroup per CU).  The last six phases of a tile,
// which have nothing of their own left to stage, issue units 0..5 of the NEXT tile, so a tile's first MFMA never waits for
// memory; and the epilogue first does half of its arithmetic, then waits for those six units (landed by then), THEN issues
// its stores (the second half's behind that half's arithmetic) and walks straight into the next tile's phase 0: the 128 KB of C drain to HBM under the next main loop instead of
// holding the CU until they are acknowledged (a wave cannot retire with stores pending, and all 256 CUs would be draining
// at once: the fixed cost per tile was 7.9 us = the chip's HBM write time for one round of tiles).  Stores count in vmcnt
// with the DMAs, so the new tile's first four phases wait for nothing (everything they read landed before the stores were
// issued) and the first counted wait comes four phases -- 1.5 us -- behind the stores.
//
// The MFMA is issued transposed (A operand = 16 rows of W, B operand = 16 rows of the activations): a lane ends up with
// FOUR CONSECUTIVE COLUMNS of one output row; two register swaps between lane groups (v_permlane32_swap, v_permlane16_swap)
// turn the lane's four 4-column packets into 16 consecutive columns: 16-byte stores, four lanes per 128-byte line.
// Workgroups are dealt to tiles XCD-wise (blocks b, b + 8, ... share an XCD): an XCD owns a contiguous run of tiles and its
// 32 workgroups take 32 consecutive ones per round; tiles are ordered in groups of `group` M-tiles, N-tile major inside, so
// those 32 share a few weight panels AND a few activation panels in the XCD's L2.
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

// v_max3_f32 as it is (fmaxf on values that come out of integer operations gets a canonicalising v_max(x, x) per operand)
__device__ __forceinline__ float jmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

constexpr int JBM = 256, JBN = 256;

template <int V>
using IC = std::integral_constant<int, V>;

// buffer descriptors of a tile's operand panels: A rows 0.. and 64.. of every 128 (each ends at row M), W
struct panels {
  __amdgpu_buffer_rsrc_t a0, a1, w;
};

template <typename T, bool LSE, bool BIAS>
__global__ __launch_bounds__(512, 2) void joint_fc_gemm8_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                                const T* __restrict__ bias, T* __restrict__ C,
                                                                float* __restrict__ pmax, float* __restrict__ psum, int M,
                                                                int N, int K, int tiles_n, int tiles_m, int group) {
  using frag = typename jfrag<T>::type;
  using v4 = __attribute__((ext_vector_type(4))) T;
  using u2 = __attribute__((ext_vector_type(2))) unsigned;
  using u4 = __attribute__((ext_vector_type(4))) unsigned;
  constexpr int UE = 128 * 64;   // elements of a unit
  __shared__ __attribute__((aligned(1024))) T u00[UE], u01[UE], u02[UE], u03[UE], u10[UE], u11[UE], u12[UE], u13[UE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int r16 = lane & 15, kq = lane >> 4;

  // ---- the workgroup's tiles: XCD x = bid & 7 owns tiles [xs, xs + xn) and deals them to its workgroups round-robin
  int t_first, t_end, t_step;
  {
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x, nt = tiles_m * tiles_n;
    if ((nwg & 7) == 0) {
      const int q = nt >> 3, r = nt & 7, x = bid & 7;
      const int xs = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
      t_step = nwg >> 3;
      t_first = xs + (bid >> 3);
      t_end = xs + (x < r ? q + 1 : q);
    } else {
      t_step = nwg;
      t_first = bid;
      t_end = nt;
    }
  }
  if (t_first >= t_end) return;           // (grid <= tiles, so only with fewer than 8 tiles on some XCD)
  auto coords = [&](int t, int& m0_, int& n0_) {
    const int per = group * tiles_n, g = t / per, w_ = t - g * per;
    const int gm = min(group, tiles_m - g * group);
    const int tn = w_ / gm;
    m0_ = (g * group + (w_ - tn * gm)) * JBM;
    n0_ = tn * JBN;
  };

  // ---- DMA sources: per unit two instructions per wave; instruction i of wave w fills unit rows (2 w + i) * 8 .. + 7, lane l
  // the chunk at (row l >> 3, position l & 7).  Per-lane offsets are those of the first half (types 0, 1); the second half
  // (types 3, 2) is 64 / 32 rows further on: for W through the scalar offset, for A through a descriptor of its own whose
  // NUM_RECORDS ends at row M -- rows past M are out of range for the buffer unit (they read as zero and are never stored), so
  // the offsets are the same for every tile.
  unsigned srca[2], srcw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rho = (wave * 2 + i) * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ ((rho >> 1) & 7);
    srca[i] = (unsigned)((((rho >> 6) * 128 + (rho & 63)) * K + ch * 8) * (int)sizeof(T));
    srcw[i] = (unsigned)((((rho >> 5) * 64 + (rho & 31)) * K + ch * 8) * (int)sizeof(T));
  }
  const int w_half = 32 * K * (int)sizeof(T);   // W rows nh * 32: scalar offset

#define CAIMAN_UNIT(PAR, TY) \
  ((PAR) == 0 ? ((TY) == 0 ? u00 : (TY) == 1 ? u01 : (TY) == 2 ? u02 : u03) : ((TY) == 0 ? u10 : (TY) == 1 ? u11 : (TY) == 2 ? u12 : u13))

  // one unit by buffer LDS-DMA: descriptor (SGPRs) = the tile's first row of the half, soff (SGPR) = byte offset of the K tile
  // wanted, per-lane 32-bit offset: no vector address arithmetic in the loop.  (The K advance must not sit in the
  // instruction's immediate offset: the hardware adds that to the LDS address as well.)
  auto issue = [&](auto PAR_, auto TY_, const panels& p, int soff) {
    constexpr int PAR = decltype(PAR_)::value, TY = decltype(TY_)::value;
    T* dst = CAIMAN_UNIT(PAR, TY);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      auto lds = (__attribute__((address_space(3))) void*)(dst + (wave * 2 + i) * 8 * 64);
      if constexpr (TY == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(p.a0, lds, 16, (int)srca[i], soff, 0, 0);
      if constexpr (TY == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(p.a1, lds, 16, (int)srca[i], soff, 0, 0);
      if constexpr (TY == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(p.w, lds, 16, (int)srcw[i], soff, 0, 0);
      if constexpr (TY == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(p.w, lds, 16, (int)srcw[i], soff + w_half, 0, 0);
    }
  };
  // (descriptor inputs go through readfirstlane: hipcc wraps every buffer operation whose descriptor it cannot PROVE
  // wave-uniform in a waterfall loop)
  auto rsrc = [&](const T* base, unsigned bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    // (readfirstlane returns a SIGNED int: widened as it is, a low half with bit 31 set smears ones over the high half)
    const uint64_t u = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a) |
                       ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T*>(u), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  };
  auto make_panels = [&](int m0_, int n0_) {
    // NUM_RECORDS: the rows of this half that exist.  A half with none keeps a base inside the tensor and one row of
    // records (what its lanes then fetch is real memory, and never stored): if the range check subtracts the scalar offset
    // from NUM_RECORDS, zero records would wrap around and put EVERY lane in range of memory behind the tensor.
    auto rows = [&](int first) { return (unsigned)(max(1, min(M - first, 256)) * K * (int)sizeof(T)); };
    panels p;
    p.a0 = rsrc(A + (int64_t)m0_ * K, rows(m0_));
    p.a1 = rsrc(A + (int64_t)(m0_ + 64 < M ? m0_ + 64 : m0_) * K, rows(m0_ + 64));
    p.w = rsrc(W + (int64_t)n0_ * K, 0x7fffffffu);
    return p;
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
  // They start at the bias (the lane's four columns), so the epilogue has no bias pass.
  f32x4 acc[2][4][2][2];
  auto init_acc = [&](const v4 (&braw)[4]) {
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
  };
  auto load_bias = [&](int n0_, int kq_, v4 (&braw)[4]) {
    if constexpr (BIAS) {
#pragma unroll
      for (int c = 0; c < 4; ++c) braw[c] = *reinterpret_cast<const v4*>(bias + n0_ + wc * 64 + c * 16 + kq_ * 4);
    }
  };

  // ---- state of the tile being multiplied, and of the one behind it
  int m0, n0, m0n = 0, n0n = 0;
  panels pc, pn;                                         // descriptors of the tile's A / W panels, and the next tile's
  int kb = 0;                                            // byte offset (in a row) of the K tile pair being multiplied
  frag af[4][2], wf[2][2];     // A fragments of the current row half; W fragments of the nh = 1 column half
  frag wz[2][2][2];        // [parity][a][ks]: the nh = 0 fragments of the K tile of that parity, read one phase early
#ifdef JG_NO_READS
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "=v"((&af[0][0])[i]), "=v"((&wz[0][0][0])[i]));   // defined, opaque
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" : "=v"((&wf[0][0])[i]));
#endif

  // one phase.  PAR: parity of the K tile being multiplied; PH: phase 0..3; SRC: 0 nothing to issue, 1 unit (phase + 6) of
  // this tile, 2 the unit of the NEXT tile that takes its place; VM: the counted wait behind it (-1: none)
  auto phase = [&](auto PAR_, auto PH_, auto SRC_, auto VM_) {
    constexpr int PAR = decltype(PAR_)::value, PH = decltype(PH_)::value, SRC = decltype(SRC_)::value, VM = decltype(VM_)::value;
    // ---- load part (JG_NO_READS / JG_NO_DMA: measurement builds that leave one of its two halves out -- results invalid;
    // tools/jgemm_ablate.py: input-gradient instance 2.80-2.93 ms shipped, 2.38 without the reads, 2.21 without the DMA issue,
    // MFMA floor 1.63 at 2.4 GHz.  Issuing each phase's second DMA instruction behind its MFMAs instead: 2.87 vs 2.80, slower)
#ifndef JG_NO_READS
    if constexpr (PH == 0) {
      const T* ua = CAIMAN_UNIT(PAR, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[b][ks] = *reinterpret_cast<const frag*>(ua + abase[ks] + b * 16 * 64);
    } else if constexpr (PH == 3) {      // the next K tile's first W unit (other parity), one phase ahead of its phase 0
      const T* uw = CAIMAN_UNIT(PAR ^ 1, 1);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wz[PAR ^ 1][a][ks] = *reinterpret_cast<const frag*>(uw + wbase[ks] + a * 16 * 64);
    } else if constexpr (PH == 1) {
      const T* uw = CAIMAN_UNIT(PAR, 2);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wf[a][ks] = *reinterpret_cast<const frag*>(uw + wbase[ks] + a * 16 * 64);
    } else if constexpr (PH == 2) {
      const T* ua = CAIMAN_UNIT(PAR, 3);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[b][ks] = *reinterpret_cast<const frag*>(ua + abase[ks] + b * 16 * 64);
    }
#endif
#ifdef JG_NO_DMA
    if constexpr (false) {
#else
    if constexpr (SRC != 0) {
#endif
      // unit phase + 6: phases 0, 1 fill types 2, 3 of the other parity (the next K tile), phases 2, 3 types 0, 1 of this
      // parity (the K tile after that).  kb is the row offset of the FIRST K tile of the pair being multiplied (parity 0).
      constexpr int DP = PH < 2 ? PAR ^ 1 : PAR, TY = (PH + 2) & 3;
      constexpr int AHEAD = PAR + (PH < 2 ? 1 : 2);                         // K tiles ahead of ga / gw
      if constexpr (SRC == 1) {
        issue(IC<DP>{}, IC<TY>{}, pc, kb + AHEAD * 64 * (int)sizeof(T));
      } else {
        constexpr int KT = AHEAD - 2;                                       // K tile 0 or 1 of the next tile
        static_assert(KT == 0 || KT == 1, "next-tile units start behind this tile's last K tile");
        issue(IC<DP>{}, IC<TY>{}, pn, KT * 64 * (int)sizeof(T));
      }
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
        for (int a = 0; a < 2; ++a) {
          if constexpr (nh == 0) acc[mh][b][nh][a] = jmfma(wz[PAR][a][ks], af[b][ks], acc[mh][b][nh][a]);
          else acc[mh][b][nh][a] = jmfma(wf[a][ks], af[b][ks], acc[mh][b][nh][a]);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using VN = IC<-1>;
  using V8 = IC<6>;        // vmcnt(6) behind every issue: the three youngest units may still fly, unit <= phase + 3 has landed
  using I0 = IC<0>;
  using I1 = IC<1>;
  using I2 = IC<2>;
  using I3 = IC<3>;

  const int nk = K / 64;                  // K % 128 == 0, K >= 256: an even number of K tiles, at least four

  // ---- first tile: its first six units, everything landed before phase 0 (as for every later tile)
  int t = t_first;
  coords(t, m0, n0);
  pc = make_panels(m0, n0);
  {
    v4 braw[4];
    load_bias(n0, kq, braw);
    issue(I0{}, I0{}, pc, 0);
    issue(I0{}, I1{}, pc, 0);
    issue(I0{}, I2{}, pc, 0);
    issue(I0{}, I3{}, pc, 0);
    issue(I1{}, I0{}, pc, 64 * (int)sizeof(T));
    issue(I1{}, I1{}, pc, 64 * (int)sizeof(T));
    init_acc(braw);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // the first tile's first W unit: every later one is read by the phase 3 in front of it (the last phase of a tile reads
  // the next tile's)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wz[0][a][ks] = *reinterpret_cast<const frag*>(u01 + wbase[ks] + a * 16 * 64);

  for (;;) {
    const int tnext = t + t_step;
    const bool has_next = tnext < t_end;
    coords(has_next ? tnext : t, m0n, n0n);
    pn = make_panels(m0n, n0n);
    if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind from here on
    // K tiles 0, 1: what the first four phases read landed before the previous tile's stores went out; no wait may come
    // before those stores have had time to drain (they count in vmcnt)
    phase(I0{}, I0{}, I1{}, VN{});
    phase(I0{}, I1{}, I1{}, VN{});
    phase(I0{}, I2{}, I1{}, VN{});
    phase(I0{}, I3{}, I1{}, VN{});
    phase(I1{}, I0{}, I1{}, V8{});
    phase(I1{}, I1{}, I1{}, V8{});
    phase(I1{}, I2{}, I1{}, V8{});
    phase(I1{}, I3{}, I1{}, V8{});
    kb = 2 * 64 * (int)sizeof(T);
    for (int kt = 2; kt < nk - 2; kt += 2) {
      phase(I0{}, I0{}, I1{}, V8{});
      phase(I0{}, I1{}, I1{}, V8{});
      phase(I0{}, I2{}, I1{}, V8{});
      phase(I0{}, I3{}, I1{}, V8{});
      phase(I1{}, I0{}, I1{}, V8{});
      phase(I1{}, I1{}, I1{}, V8{});
      phase(I1{}, I2{}, I1{}, V8{});
      phase(I1{}, I3{}, I1{}, V8{});
      kb += 2 * 64 * (int)sizeof(T);
    }
    // the last two K tiles: two units of this tile are still to be issued; the six phases behind them stage the next tile.
    // (The last tile of a workgroup stages its OWN first six units again, into buffers nobody reads any more: one code path,
    // the same wait counts, 96 KB of L2 reads once per workgroup -- a second copy of these eight phases with other counts
    // made the register allocator shuffle the accumulators through scratch at the join.)
    phase(I0{}, I0{}, I1{}, V8{});
    phase(I0{}, I1{}, I1{}, V8{});
    phase(I0{}, I2{}, I2{}, V8{});
    phase(I0{}, I3{}, I2{}, V8{});
    phase(I1{}, I0{}, I2{}, V8{});
    phase(I1{}, I1{}, I2{}, V8{});
    phase(I1{}, I2{}, I2{}, V8{});
    phase(I1{}, I3{}, I2{}, V8{});
    if (wr == 0) __builtin_amdgcn_s_barrier();   // waves 0-3 meet the last barrier of waves 4-7

    // ---- epilogue, arithmetic first.  Per block of 16 rows a lane holds, for its row r16, the four-column packets (c, kq):
    // columns c * 16 + kq * 4 .. + 3.  Two register swaps between lane groups (a 4 x 4 transpose of packets between the
    // register index c and the lane group kq) leave it with columns kq * 16 .. + 15 -- 32 contiguous bytes.  The row's
    // (reference, sum exp) over the wave's 64 columns is taken on the values AS STORED; the reference point is the row maximum
    // clamped to +-2e38, so that rows of -inf / +inf need no per-element special case (exp2(-inf) = 0; +inf - finite = +inf).
    // (the lane id is taken afresh, inside an asm: derived from the kernel's lane variable, the 64-bit store and bias addresses
    // -- or the lane id itself -- live in registers across the whole main loop and get spilled, and the reload sits behind
    // a vmcnt(0) at the head of the epilogue)
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int r16e = lane_e & 15, kqe = lane_e >> 4;
    v4 braw[4];
    load_bias(n0n, kqe, braw);
    constexpr float kLog2e = 1.4426950408889634f;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      // four blocks of 16 rows at a time, stage by stage: every stage is four independent pieces of work, so that the
      // cross-lane swaps, the exponentials and the additions of one block run in the shadow of the others' (as one block
      // after the other this epilogue was a single dependency chain: 1050 cycles per block for 90 instructions)
      u4 outv[4][2];
      float pm[4], ps[4];
      u2 P[4][4];
      float v[4][4][4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = static_cast<T>(acc[mh][b][c >> 1][c & 1][j]);
          P[b][c] = __builtin_bit_cast(u2, o);
          if constexpr (LSE) {
            // the stored values as floats, out of the packed words (one shift or mask each; left to the compiler every
            // element was rounded a second time on its own)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              if constexpr (std::is_same<T, bf16_t>::value) {
                v[b][c][2 * d] = __builtin_bit_cast(float, P[b][c][d] << 16);
                v[b][c][2 * d + 1] = __builtin_bit_cast(float, P[b][c][d] & 0xffff0000u);
              } else {
                v[b][c][2 * d] = static_cast<float>(o[2 * d]);
                v[b][c][2 * d + 1] = static_cast<float>(o[2 * d + 1]);
              }
            }
          }
        }
      }
      if constexpr (LSE) {
        float mx[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float m0_ = jmax3(jmax3(v[b][0][0], v[b][0][1], v[b][0][2]), v[b][0][3], v[b][1][0]);
          const float m1_ = jmax3(jmax3(v[b][1][1], v[b][1][2], v[b][1][3]), v[b][2][0], v[b][2][1]);
          const float m2_ = jmax3(jmax3(v[b][2][2], v[b][2][3], v[b][3][0]), v[b][3][1], v[b][3][2]);
          mx[b] = jmax3(jmax3(m0_, m1_, m2_), v[b][3][3], v[b][3][3]);
        }
        jpair h[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) h[b] = jswap32(mx[b]);          // over the four lanes (kq) that share the row
#pragma unroll
        for (int b = 0; b < 4; ++b) mx[b] = fmaxf(h[b].a, h[b].b);
#pragma unroll
        for (int b = 0; b < 4; ++b) h[b] = jswap16(mx[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          mx[b] = fminf(fmaxf(fmaxf(h[b].a, h[b].b), -2e38f), 2e38f);
          pm[b] = mx[b];
        }
        float sm[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float nmx = -mx[b] * kLog2e;
          float sc[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            sc[c] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[b][c][0], kLog2e, nmx));
#pragma unroll
            for (int j = 1; j < 4; ++j) sc[c] += __builtin_amdgcn_exp2f(__builtin_fmaf(v[b][c][j], kLog2e, nmx));
          }
          sm[b] = (sc[0] + sc[1]) + (sc[2] + sc[3]);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) h[b] = jswap32(sm[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b) sm[b] = h[b].a + h[b].b;
#pragma unroll
        for (int b = 0; b < 4; ++b) h[b] = jswap16(sm[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b) ps[b] = h[b].a + h[b].b;
      }
      // packets (c, kq) -> (kq, c)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          auto s02 = __builtin_amdgcn_permlane32_swap(P[b][0][d], P[b][2][d], false, false);
          auto s13 = __builtin_amdgcn_permlane32_swap(P[b][1][d], P[b][3][d], false, false);
          auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
          auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
          P[b][0][d] = t01[0];
          P[b][1][d] = t01[1];
          P[b][2][d] = t23[0];
          P[b][3][d] = t23[1];
        }
        outv[b][0] = u4{P[b][0][0], P[b][0][1], P[b][1][0], P[b][1][1]};
        outv[b][1] = u4{P[b][2][0], P[b][2][1], P[b][3][0], P[b][3][1]};
      }
      // the next tile's first six units (issued over the last six phases) have landed by the time half the arithmetic is
      // done: wait for them BEFORE the first store goes out, so that no wait of the next tile has these stores in front of it
      if (mh == 0) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int m = m0 + wr * 128 + mh * 64 + b * 16 + r16e;
        if (m < M) {
          T* crow = C + (int64_t)m * N + n0 + wc * 64 + kqe * 16;
          *reinterpret_cast<u4*>(crow) = outv[b][0];
          *reinterpret_cast<u4*>(crow + 8) = outv[b][1];
          if constexpr (LSE) {
            if (kqe == 0) {     // partials are [N / 64][M]: the 16 rows of a block are 64 contiguous bytes
              const int64_t p = (int64_t)((n0 >> 6) + wc) * M + m;
              pmax[p] = pm[b];
              psum[p] = ps[b];
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!has_next) break;
    // ---- walk into the next tile
    t = tnext;
    m0 = m0n;
    n0 = n0n;
    pc = pn;
    kb = 0;
    init_acc(braw);
    __builtin_amdgcn_s_barrier();           // every wave's share of units 0..5 has landed (each waited for its own above)
    __builtin_amdgcn_sched_barrier(0);
  }
#undef CAIMAN_UNIT
}

// row normaliser from the partial pairs: lse = R + log(sum_p psum_p * exp(pmax_p - R)), R = max_p pmax_p.  Partials are
// [N / 64][M] (written 64 contiguous bytes at a time by the GEMM; as [M][N / 64] every store instruction touched 16 lines with
// four bytes each and the epilogue reduction cost 0.7 ms at 304 000 x 8704): one THREAD per row, coalesced across rows.
__global__ __launch_bounds__(256) void lse_partials_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                          float* __restrict__ lse, int64_t rows, int NP) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  float mx = -INFINITY;
  for (int p = 0; p < NP; ++p) mx = fmaxf(mx, pmax[(int64_t)p * rows + row]);
  float sm = 0.f;
  for (int p = 0; p < NP; ++p) {
    const float m = pmax[(int64_t)p * rows + row];
    const float q = psum[(int64_t)p * rows + row];
    sm += (m == mx) ? q : q * __expf(m - mx);     // reference points are finite (clamped): m - mx never inf - inf
  }
  lse[row] = mx + logf(sm);
}

template <typename T>
int launch_joint_fc(const T* A, const T* W, const T* bias, T* C, float* lse, float* ws, int64_t M, int64_t N, int64_t K,
                    hipStream_t s) {
  const int tiles_n = (int)(N / JBN), tiles_m = (int)((M + JBM - 1) / JBM);
  const int64_t tiles = (int64_t)tiles_m * tiles_n;
  const int NP = (int)(N / 64);
  // tile order: M-tiles per group (measured, 304 000 x 768 x 8704: 8 with many N-tiles; 1 for the input gradient's three)
  static const int group_env = std::getenv("CAIMAN_JOINT_GROUP") ? std::atoi(std::getenv("CAIMAN_JOINT_GROUP")) : 0;
  const int group = std::max(1, std::min(group_env > 0 ? group_env : (tiles_n >= 8 ? 8 : 1), tiles_m));
  // one persistent workgroup per CU (CAIMAN_JOINT_WGS overrides: measurement knob)
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    if (const char* e = std::getenv("CAIMAN_JOINT_WGS")) n = std::max(1, std::atoi(e));
    return n;
  }();
  const unsigned grid = (unsigned)std::min<int64_t>(tiles, cus);
  float* pmax = lse ? ws : nullptr;
  float* psum = lse ? ws + M * NP : nullptr;
#define CAIMAN_JGEMM8(L, B)                                                                                              \
  hipLaunchKernelGGL((joint_fc_gemm8_kernel<T, L, B>), dim3(grid), dim3(512), 0, s, A, W, bias, C, pmax, psum, (int)M, (int)N, \
                     (int)K, tiles_n, tiles_m, group)
  if (lse) {
    if (bias) CAIMAN_JGEMM8(true, true); else CAIMAN_JGEMM8(true, false);
    hipLaunchKernelGGL(lse_partials_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, pmax, psum, lse, M, NP);
  } else {
    if (bias) CAIMAN_JGEMM8(false, true); else CAIMAN_JGEMM8(false, false);
  }
#undef CAIMAN_JGEMM8
  return check_launch("joint projection GEMM");
}

}  // namespace
}  // namespace caiman

extern "C" int64_t caiman_joint_fc_workspace_elems(int64_t M, int64_t N) { return 2 * M * (N / 64); }

extern "C" int caiman_joint_fc_supported(int64_t M, int64_t N, int64_t K, int dtype) {
  return (dtype == CAIMAN_BF16 || dtype == CAIMAN_F16) && M >= 1 && N >= 256 && N % 256 == 0 && K >= 256 && K % 128 == 0 &&
                 M * N < ((int64_t)1 << 40) && (M + 255) / 256 * (N / 256) < ((int64_t)1 << 31) && 256 * K * 2 < ((int64_t)1 << 31)
             ? 1 : 0;
}

extern "C" int caiman_joint_fc_forward(const void* A, const void* W, const void* bias, void* C, float* lse, float* workspace,
                                       int64_t M, int64_t N, int64_t K, int dtype, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(caiman_joint_fc_supported(M, N, K, dtype),
               "joint_fc_forward: bf16 / f16, N %% 256 == 0, K %% 128 == 0, K >= 256 (got M %lld N %lld K %lld dtype %d)",
               (long long)M, (long long)N, (long long)K, dtype);
  CAIMAN_CHECK(A && W && C && (lse == nullptr || workspace != nullptr), "joint_fc_forward: null pointer");
  auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; };
  CAIMAN_CHECK(al(A, 16) && al(W, 16) && al(C, 16) && (!bias || al(bias, 8)), "joint_fc_forward: operands and C 16-byte, bias 8-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == CAIMAN_BF16)
    return launch_joint_fc<bf16_t>((const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)bias, (bf16_t*)C, lse, workspace, M, N, K, s);
  return launch_joint_fc<f16_t>((const f16_t*)A, (const f16_t*)W, (const f16_t*)bias, (f16_t*)C, lse, workspace, M, N, K, s);
}
