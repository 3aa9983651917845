// Weight gradient of the joint projection as a hand-written MFMA GEMM (backward).
//
// Reference: autograd of `self.joint_fc(h)` = torch.nn.Linear(joint_n_hid, n_classes)
// (training/caiman_asr_train/rnnt/model.py:409-439): dW = dY^T . h with dY [M, N] the gradient of the packed logits
// (M = 304 000 lattice cells at LibriSpeech shapes, N = 8704) and h [M, K] the joint activations (K = 768).  The
// reduction runs over M, the SLOW dimension of both operands: for the matrix cores every fragment is a transposed read.
// hipBLASLt serves it at 0.92 PF/s (4.4 ms per training step; as 16 row chunks in one batched call, train_utils/overlap.py).
//
//   dW[n][k] = sum_m dY[m][n] * h[m][k]        bf16 / f16 operands, fp32 accumulation, fp32 output slabs
//
// Geometry: csrc/joint_gemm.hip's -- 256 x 256 output tile per workgroup, 8 waves (2 along n x 4 along k), wave tile
// 128 x 64, the reduction walked in steps of 32 rows of m through a ring of four LDS stages filled by LDS-DMA with three
// stages in flight (one counted vmcnt + one bare barrier per step) -- with two differences:
//  * a stage holds 32 ROWS OF m of both operands as they lie in memory ([32][256] elements, 512-byte rows), and the MFMA
//    fragments (16 n or k columns x 32 m) are read with `ds_read_b64_tr_b16` (gfx950's transposing LDS read: a group of 16
//    lanes reads a block of 4 rows x 16 columns and every lane receives one column, guide T10): two reads per fragment.
//    The 32-byte granules of a row are XOR-swizzled on the DMA source with x(row) = (row & 7) ^ (((row >> 3) & 1) << 2), so
//    that the two 4-row blocks a 32-lane half reads at a time (rows r .. r+3 and r+8 .. r+11) cover all eight granules of a
//    256-byte bank window.
//  * the output tile is tiny and the reduction is long, so M is split: grid = (N/256) * (K/256) tiles x S slices of M, each
//    workgroup writes its fp32 partial tile into slab s; the caller adds the S slabs (and the < 128 S rows the slices do not
//    cover) in a fixed order.  A workgroup's loop is M / (32 S) steps long (~1 900): no prologue / epilogue to speak of.
// The three k-tiles that share a dY panel are consecutive workgroups (same XCD after the remap), so the 5.3 GB of dY cross
// HBM once.
#include <algorithm>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "common.h"

namespace caiman {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
template <typename T>
struct wfrag {
  using type = __attribute__((ext_vector_type(8))) T;
};
__device__ __forceinline__ f32x4 wmfma(wfrag<bf16_t>::type a, wfrag<bf16_t>::type b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 wmfma(wfrag<f16_t>::type a, wfrag<f16_t>::type b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// transposing read: lane 4q + p of a 16-lane group supplies the address of row q, columns 4p .. 4p+3 of a 4 x 16 block;
// lane i of the group receives column i, row q in element q
// The address is formed on the stage's own LDS array (array + byte offset): the compiler then knows which stage a read
// touches and does not wait for the LDS-DMA in flight into the other stages.
__device__ __forceinline__ bf16x4 tr_read(const bf16_t* stage, unsigned byte) {
  const auto* p = (const __attribute__((address_space(3))) char*)stage + byte;
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}
__device__ __forceinline__ f16x4 tr_read(const f16_t* stage, unsigned byte) {
  using s16x4 = __attribute__((__vector_size__(4 * sizeof(short)))) short;
  const auto* p = (const __attribute__((address_space(3))) char*)stage + byte;
  const s16x4 raw = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  f16x4 v;
  __builtin_memcpy(&v, &raw, 8);
  return v;
}

constexpr int WBN = 256, WBK = 256, WBM = 32, WNW = 8;   // tile: 256 n x 256 k, 32 rows of m per stage
constexpr int WTN = 8, WTK = 4;                            // 16 x 16 blocks of a wave tile: 128 n, 64 k
constexpr int WROW = 256;                                  // elements per LDS row of a stage (both operands)
template <int V>
using IC8 = std::integral_constant<int, V>;

template <typename T>
__global__ __launch_bounds__(64 * WNW, 1) void joint_wgrad_kernel(const T* __restrict__ dY, const T* __restrict__ Hm,
                                                                 float* __restrict__ slabs, int N, int K, int rows_per_slice,
                                                                 int tiles_k, int n_tiles, int slices, int64_t stride_y,
                                                                 int64_t stride_h, int split, const T* __restrict__ dY2,
                                                                 int64_t stride_y2, const T* __restrict__ Hm2,
                                                                 int64_t stride_h2) {
  using frag = typename wfrag<T>::type;
  __shared__ __attribute__((aligned(1024))) T sA0[WBM * WROW], sA1[WBM * WROW], sA2[WBM * WROW], sA3[WBM * WROW];   // dY rows
  __shared__ __attribute__((aligned(1024))) T sB0[WBM * WROW], sB1[WBM * WROW], sB2[WBM * WROW], sB3[WBM * WROW];   // h rows

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t;
  {
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // t -> (problem, slice of M, n-tile, k-tile): k-tile fastest, so the workgroups that share a dY panel are neighbours
  const int pslice = t / n_tiles, tt = t - pslice * n_tiles;      // pslice = problem * slices + slice: the slab this tile writes
  const int prob = pslice / slices, slice = pslice - prob * slices;
  const int n0 = (tt / tiles_k) * WBN, k0 = (tt % tiles_k) * WBK;
  const int64_t m_begin = (int64_t)slice * rows_per_slice;
  // products [0, split) lie a constant stride apart from (dY, Hm), products [split, ..) from (dY2, Hm2): two strided groups
  // of one shape in one launch (the layers' dR and dW: same gradients, two different activation buffers)
  if (prob < split) {
    dY += (int64_t)prob * stride_y;
    Hm += (int64_t)prob * stride_h;
  } else {
    dY = dY2 + (int64_t)(prob - split) * stride_y2;
    Hm = Hm2 + (int64_t)(prob - split) * stride_h2;
  }

  // DMA sources: a stage = 32 rows x 512 B per operand = 16 instructions of 2 rows; wave w brings instructions w and w + 8.
  // Lane l writes the 16-byte piece at (row l >> 5, position l & 31) and fetches the piece that belongs there: granule
  // (position >> 1) ^ x(row), same half.
  unsigned a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave + WNW * i) * 2 + (lane >> 5), pos = lane & 31;
    const int x = (row & 7) ^ (((row >> 3) & 1) << 2);
    const int src_piece = (((pos >> 1) ^ x) << 1) | (pos & 1);
    a_off[i] = (unsigned)((row * N + src_piece * 8) * (int)sizeof(T));
    b_off[i] = (unsigned)((row * K + src_piece * 8) * (int)sizeof(T));
  }
  const char* a_base = reinterpret_cast<const char*>(dY + m_begin * N + n0);
  const char* b_base = reinterpret_cast<const char*>(Hm + m_begin * K + k0);
  const int64_t a_step = (int64_t)WBM * N * (int64_t)sizeof(T), b_step = (int64_t)WBM * K * (int64_t)sizeof(T);

  auto issue = [&](T* lA, T* lB, int s) {
    const char* ab = a_base + (int64_t)s * a_step;
    const char* bb = b_base + (int64_t)s * b_step;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ab + a_off[i]),
                                       (__attribute__((address_space(3))) void*)(lA + (wave + WNW * i) * 2 * WROW), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bb + b_off[i]),
                                       (__attribute__((address_space(3))) void*)(lB + (wave + WNW * i) * 2 * WROW), 16, 0, 0);
  };

  const int wr = wave >> 2, wc = wave & 3;        // wave tile: n [128 wr, +128), k [64 wc, +64)
  const int kq = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
  // transposed-read addresses (bytes inside a stage): block rows 8 kq + 4 h + q4 (h = 0, 1), columns 16 c + 4 p4.  Column
  // block c is granule c of the row, stored at granule c ^ x(row); bits 5..8 of a row's byte offset hold nothing else,
  // so address(c, h) = base[h] ^ (c << 5) with x folded into the base: one v_xor with a constant per read.
  unsigned baseA[2], baseB[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * kq + 4 * h + q4;
    const int x = (row & 7) ^ (((row >> 3) & 1) << 2);
    const unsigned b0 = (unsigned)(row * WROW * (int)sizeof(T) + p4 * 8 + (x << 5));
    baseA[h] = b0 ^ (unsigned)(wr << 8);   // c = 8 wr + a
    baseB[h] = b0 ^ (unsigned)(wc << 7);   // c = 4 wc + b
  }
  f32x4 acc[WTN][WTK];
#pragma unroll
  for (int a = 0; a < WTN; ++a)
#pragma unroll
    for (int b = 0; b < WTK; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const T* lA, const T* lB) {
    frag bf[WTK], af[2];
    // the 24 read addresses are two instructions away from these four registers; keeping them opaque per step stops the
    // compiler from hoisting 24 x 4 stage addresses out of the loop (and spilling them)
    asm volatile("" : "+v"(baseA[0]), "+v"(baseA[1]), "+v"(baseB[0]), "+v"(baseB[1]));
    auto read_a = [&](int a, frag& f) {
      const auto lo = tr_read(lA, baseA[0] ^ (unsigned)(a << 5));
      const auto hi = tr_read(lA, baseA[1] ^ (unsigned)(a << 5));
#pragma unroll
      for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
    };
#pragma unroll
    for (int b = 0; b < WTK; ++b) {
      const auto lo = tr_read(lB, baseB[0] ^ (unsigned)(b << 5));
      const auto hi = tr_read(lB, baseB[1] ^ (unsigned)(b << 5));
#pragma unroll
      for (int e = 0; e < 4; ++e) { bf[b][e] = lo[e]; bf[b][4 + e] = hi[e]; }
    }
    read_a(0, af[0]);
#pragma unroll
    for (int a = 0; a < WTN; ++a) {
      if (a + 1 < WTN) read_a(a + 1, af[(a + 1) & 1]);     // one fragment ahead of the MFMAs that consume af[a & 1]
#pragma unroll
      for (int b = 0; b < WTK; ++b) acc[a][b] = wmfma(af[a & 1], bf[b], acc[a][b]);
    }
  };

  auto steady = [&](const T* cA, const T* cB, T* nA, T* nB, int s) {
    __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8) lgkmcnt(0): stages s + 1, s + 2 may still fly
    __builtin_amdgcn_s_barrier();
    issue(nA, nB, s + 3);
    __builtin_amdgcn_sched_barrier(0);
    compute(cA, cB);
  };
  // drain step: `w` = the wait that leaves the later stages in flight (vmcnt 8, 4, 0 for the last three steps)
  auto drain = [&](const T* cA, const T* cB, auto w) {
    __builtin_amdgcn_s_waitcnt(decltype(w)::value);
    __builtin_amdgcn_s_barrier();
    compute(cA, cB);
  };
  using w8 = std::integral_constant<int, 0x0078>;
  using w4 = std::integral_constant<int, 0x0074>;
  using w0 = std::integral_constant<int, 0x0070>;
  const int nk = rows_per_slice / WBM;    // at least three
  issue(sA0, sB0, 0);
  issue(sA1, sB1, 1);
  issue(sA2, sB2, 2);
  int s = 0;
  for (; s + 7 <= nk; s += 4) {           // four steady steps need stages s + 3 .. s + 6
    steady(sA0, sB0, sA3, sB3, s);
    steady(sA1, sB1, sA0, sB0, s + 1);
    steady(sA2, sB2, sA1, sB1, s + 2);
    steady(sA3, sB3, sA2, sB2, s + 3);
  }
  // 3 .. 6 steps are left, the current one in buffer 0: (left - 3) steady steps, then the three that only drain
  switch (nk - s) {
    case 3:
      drain(sA0, sB0, w8{}); drain(sA1, sB1, w4{}); drain(sA2, sB2, w0{});
      break;
    case 4:
      steady(sA0, sB0, sA3, sB3, s);
      drain(sA1, sB1, w8{}); drain(sA2, sB2, w4{}); drain(sA3, sB3, w0{});
      break;
    case 5:
      steady(sA0, sB0, sA3, sB3, s); steady(sA1, sB1, sA0, sB0, s + 1);
      drain(sA2, sB2, w8{}); drain(sA3, sB3, w4{}); drain(sA0, sB0, w0{});
      break;
    default:
      steady(sA0, sB0, sA3, sB3, s); steady(sA1, sB1, sA0, sB0, s + 1); steady(sA2, sB2, sA1, sB1, s + 2);
      drain(sA3, sB3, w8{}); drain(sA0, sB0, w4{}); drain(sA1, sB1, w0{});
      break;
  }

  // epilogue: D[i][j] of a block sits in lane (j = lane & 15, i = 4 (lane >> 4) + reg): rows = n (A operand), columns = k
  float* out = slabs + ((int64_t)pslice * N + n0 + wr * 128) * K + k0 + wc * 64;
#pragma unroll
  for (int a = 0; a < WTN; ++a)
#pragma unroll
    for (int b = 0; b < WTK; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) out[(int64_t)(a * 16 + kq * 4 + e) * K + b * 16 + li] = acc[a][b][e];
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 4: the same product on the 8-phase main loop of csrc/joint_gemm.hip (the two waves of a SIMD in opposite roles: one
// multiplies 16 MFMAs while its partner reads fragments and issues LDS-DMA, then they swap; waves 4-7 run one barrier behind).
// The reduction over m advances in tiles of 64 rows = four phases (one 64 n x 32 k quadrant of the wave's 128 x 64 tile each);
// a tile is staged as four units of 16 KB -- [64 rows of m][128 columns], 256-byte rows:
//   type 0  dY columns {wr * 128 + 0 .. 63}    type 3  dY columns {wr * 128 + 64 .. 127}
//   type 1  h  columns {wc * 64 + 0 .. 31}     type 2  h  columns {wc * 64 + 32 .. 63}
// unit u = 4 * tile + type issued at phase u - 6, vmcnt(8) behind every phase's issue, read one phase after the wait that
// retires it (joint_gemm.hip has the hazard analysis).  Fragments by ds_read_b64_tr_b16 as above; a row's eight 32-byte granules
// are stored at g ^ x(row), x = (row & 3) | (((row >> 3) & 1) << 2): the eight rows a 32-lane half touches (r .. r + 3 and
// r + 8 .. r + 11) then cover all eight granules of the 256-byte bank window.  Operands arrive by BUFFER LDS-DMA (descriptor +
// scalar row offset + per-lane 32-bit offset: no vector address arithmetic in the loop).
template <typename T>
__global__ __launch_bounds__(512, 2) void joint_wgrad8_kernel(const T* __restrict__ dY, const T* __restrict__ Hm,
                                                              float* __restrict__ slabs, int N, int K, int rows_per_slice,
                                                              int tiles_k, int n_tiles, int slices, int64_t stride_y,
                                                              int64_t stride_h, int split, const T* __restrict__ dY2,
                                                              int64_t stride_y2, const T* __restrict__ Hm2, int64_t stride_h2,
                                                              int64_t rows_total) {
  using frag = typename wfrag<T>::type;
  constexpr int UE = 64 * 128;
  __shared__ __attribute__((aligned(1024))) T u00[UE], u01[UE], u02[UE], u03[UE], u10[UE], u11[UE], u12[UE], u13[UE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
  int t;
  {
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int pslice = t / n_tiles, tt = t - pslice * n_tiles;
  const int prob = pslice / slices, slice = pslice - prob * slices;
  const int n0 = (tt / tiles_k) * WBN, k0 = (tt % tiles_k) * WBK;
  const int64_t m_begin = (int64_t)slice * rows_per_slice;
  if (prob < split) {
    dY += (int64_t)prob * stride_y;
    Hm += (int64_t)prob * stride_h;
  } else {
    dY = dY2 + (int64_t)(prob - split) * stride_y2;
    Hm = Hm2 + (int64_t)(prob - split) * stride_h2;
  }
  const int wr = wave >> 2, wc = wave & 3;
  const int kq = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;

  // DMA: instruction i of wave w fills unit rows (2 w + i) * 4 .. + 3; lane l the 16-byte piece at (row l >> 4, position l & 15)
  unsigned srcy[2], srch[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 4 + (lane >> 4), p = lane & 15;
    const int x = (r & 3) | (((r >> 3) & 1) << 2);
    const int cu = (((p >> 1) ^ x) << 4) | ((p & 1) << 3);       // unit column of the piece that belongs at this position
    srcy[i] = (unsigned)((r * N + (cu >> 6) * 128 + (cu & 63)) * (int)sizeof(T));
    srch[i] = (unsigned)((r * K + (cu >> 5) * 64 + (cu & 31)) * (int)sizeof(T));
  }
  // The descriptors end with the operand's last row: the last slice also takes the rows the slices do not cover
  // (rows_total - slices * rows_per_slice of them, fewer than 128 per slice), rounded up to whole pairs of tiles -- the rows
  // past the end read as zeros (an out-of-range buffer load returns 0, and that is what LDS-DMA writes), exact in the sum.
  auto rsrc = [&](const T* base, int64_t valid_elems) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    const uint64_t u = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a) |
                       ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32);
    const int64_t bytes = valid_elems * (int64_t)sizeof(T);
    const unsigned rec = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : bytes));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T*>(u), 0, (int)rec, 0x00020000);
  };
  const int64_t rows_left = rows_total - m_begin;     // rows of this product from the slice's first row to the end
  const __amdgpu_buffer_rsrc_t ry = rsrc(dY + m_begin * N + n0, rows_left * N - n0), rh = rsrc(Hm + m_begin * K + k0, rows_left * K - k0);
  const unsigned ystep = (unsigned)(64 * N * (int)sizeof(T)), hstep = (unsigned)(64 * K * (int)sizeof(T));   // one tile of m
  unsigned yb = 0, hb = 0;        // byte offsets of the first tile of the pair being multiplied

#define CAIMAN_WUNIT(PAR, TY) \
  ((PAR) == 0 ? ((TY) == 0 ? u00 : (TY) == 1 ? u01 : (TY) == 2 ? u02 : u03) : ((TY) == 0 ? u10 : (TY) == 1 ? u11 : (TY) == 2 ? u12 : u13))
  auto issue = [&](auto PAR_, auto TY_, unsigned yoff, unsigned hoff) {
    constexpr int PAR = decltype(PAR_)::value, TY = decltype(TY_)::value;
    T* dst = CAIMAN_WUNIT(PAR, TY);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      auto lds = (__attribute__((address_space(3))) void*)(dst + (wave * 2 + i) * 4 * 128);
      if constexpr (TY == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, lds, 16, (int)srcy[i], (int)yoff, 0, 0);
      if constexpr (TY == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, lds, 16, (int)srcy[i], (int)(yoff + 64 * sizeof(T)), 0, 0);
      if constexpr (TY == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, lds, 16, (int)srch[i], (int)hoff, 0, 0);
      if constexpr (TY == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, lds, 16, (int)srch[i], (int)(hoff + 32 * sizeof(T)), 0, 0);
    }
  };

  // transposed-read addresses (bytes in a unit), m-step 0; m-step 1 is 32 rows = 8192 bytes further on
  unsigned adA[4][2], adB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * kq + 4 * h + q4;
    const int x = (row & 3) | (((row >> 3) & 1) << 2);
#pragma unroll
    for (int a = 0; a < 4; ++a) adA[a][h] = (unsigned)(row * 256 + (((wr * 4 + a) ^ x) << 5) + p4 * 8);
#pragma unroll
    for (int b = 0; b < 2; ++b) adB[b][h] = (unsigned)(row * 256 + (((wc * 2 + b) ^ x) << 5) + p4 * 8);
  }
  f32x4 acc[2][4][2][2];     // [nh][a][kh][b]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[i][a][j][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  frag af[4][2], bf[2][2][2];   // [a][ms], [kh][b][ms]
  auto read_frag = [&](const T* unit, unsigned a0, unsigned a1, int ms, frag& f) {
    const auto lo = tr_read(unit, a0 + (unsigned)(ms * 8192));
    const auto hi = tr_read(unit, a1 + (unsigned)(ms * 8192));
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
  };

  auto phase = [&](auto PAR_, auto PH_, auto SRC_, auto VM_) {
    constexpr int PAR = decltype(PAR_)::value, PH = decltype(PH_)::value, SRC = decltype(SRC_)::value, VM = decltype(VM_)::value;
    if constexpr (PH == 0) {
      const T* ub = CAIMAN_WUNIT(PAR, 1);
      const T* ua = CAIMAN_WUNIT(PAR, 0);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) read_frag(ub, adB[b][0], adB[b][1], ms, bf[0][b][ms]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) read_frag(ua, adA[a][0], adA[a][1], ms, af[a][ms]);
    } else if constexpr (PH == 1) {
      const T* ub = CAIMAN_WUNIT(PAR, 2);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) read_frag(ub, adB[b][0], adB[b][1], ms, bf[1][b][ms]);
    } else if constexpr (PH == 2) {
      const T* ua = CAIMAN_WUNIT(PAR, 3);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) read_frag(ua, adA[a][0], adA[a][1], ms, af[a][ms]);
    }
    if constexpr (SRC != 0) {
      constexpr int DP = PH < 2 ? PAR ^ 1 : PAR, TY = (PH + 2) & 3, AHEAD = PAR + (PH < 2 ? 1 : 2);
      issue(IC8<DP>{}, IC8<TY>{}, yb + AHEAD * ystep, hb + AHEAD * hstep);
    }
    if constexpr (VM >= 0) __builtin_amdgcn_s_waitcnt(0x0F70 | VM);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);                                  // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
    constexpr int nh = PH >> 1, kh = (PH == 1 || PH == 2) ? 1 : 0;       // quadrants (0,0) (0,1) (1,1) (1,0)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[nh][a][kh][b] = wmfma(af[a][ms], bf[kh][b][ms], acc[nh][a][kh][b]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = IC8<0>; using I1 = IC8<1>; using I2 = IC8<2>; using I3 = IC8<3>;
  using VN = IC8<-1>; using V8 = IC8<8>;

  // tiles of 64 rows: even, at least four; the last slice runs on to the operand's end
  const int nk = slice == slices - 1 ? (int)((rows_left + 127) / 128) * 2 : rows_per_slice / 64;
  issue(I0{}, I0{}, 0u, 0u);
  issue(I0{}, I1{}, 0u, 0u);
  issue(I0{}, I2{}, 0u, 0u);
  issue(I0{}, I3{}, 0u, 0u);
  issue(I1{}, I0{}, ystep, hstep);
  issue(I1{}, I1{}, ystep, hstep);
  __builtin_amdgcn_s_waitcnt(0x0F78);     // vmcnt(8): units 0, 1 have landed
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind from here on
  for (int kt = 0; kt < nk - 2; kt += 2) {
    phase(I0{}, I0{}, I1{}, V8{});
    phase(I0{}, I1{}, I1{}, V8{});
    phase(I0{}, I2{}, I1{}, V8{});
    phase(I0{}, I3{}, I1{}, V8{});
    phase(I1{}, I0{}, I1{}, V8{});
    phase(I1{}, I1{}, I1{}, V8{});
    phase(I1{}, I2{}, I1{}, V8{});
    phase(I1{}, I3{}, I1{}, V8{});
    yb += 2 * ystep;
    hb += 2 * hstep;
  }
  phase(I0{}, I0{}, I1{}, V8{});
  phase(I0{}, I1{}, I1{}, V8{});
  phase(I0{}, I2{}, I0{}, IC8<6>{});
  phase(I0{}, I3{}, I0{}, IC8<4>{});
  phase(I1{}, I0{}, I0{}, IC8<2>{});
  phase(I1{}, I1{}, I0{}, IC8<0>{});
  phase(I1{}, I2{}, I0{}, VN{});
  phase(I1{}, I3{}, I0{}, VN{});
  if (wr == 0) __builtin_amdgcn_s_barrier();
#undef CAIMAN_WUNIT

  // epilogue: D[i][j] of a block sits in lane (j = lane & 15, i = 4 (lane >> 4) + reg): rows = n (A operand), columns = k
  float* out = slabs + ((int64_t)pslice * N + n0 + wr * 128) * K + k0 + wc * 64;
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            out[(int64_t)(nh * 64 + a * 16 + kq * 4 + e) * K + kh * 32 + b * 16 + li] = acc[nh][a][kh][b][e];
}

}  // namespace
}  // namespace caiman

namespace caiman {
namespace {
// One workgroup per CU (128 KB of LDS), so the grid runs in rounds of 256 workgroups and a slice count is as good as its
// last round is full: time(s) = rounds(s) x (steps per slice x 0.86 us + 17 us) + s slabs per product written and read back at
// ~5 TB/s + the caller's library product of the rows left over.  0.86 us = the measured step of 32 rows; 17 us = what a
// workgroup spends outside its loop (three stages of DMA latency in front, 256 KB of fp32 partial tile behind: short slices
// pay it per round -- five LSTM layers at 8 900 rows measured 669 us where the loop alone predicts 350).
// 8704 x 768 (joint): 102 tiles -> 5 slices (510 workgroups, 2 rounds); 17408 x 1024: 272 tiles -> 16 slices (17 rounds; one slice
// would leave the second round 6 % full); six LSTM layers of 4096 x 1024 at 8 900 rows: 384 tiles -> 2 slices (3 rounds).
int wgrad_plan(int64_t M, int64_t N, int64_t K, int batch, int dtype, int64_t* rows_per_slice, double* seconds) {
  if (!(dtype == CAIMAN_BF16 || dtype == CAIMAN_F16) || N < 256 || N % 256 || K < 256 || K % 256 || M < 256 || batch < 1)
    return 0;
  if (N * 2 * 32 >= ((int64_t)1 << 31) || M * N >= ((int64_t)1 << 46)) return 0;
  const int64_t tiles = (N / 256) * (K / 256) * batch;
  static const int env_s = std::getenv("CAIMAN_WGRAD_SLICES") ? std::atoi(std::getenv("CAIMAN_WGRAD_SLICES")) : 0;
  // rounds are counted in workgroups per CU of THIS device (one workgroup per CU: 128 KB of LDS)
  static const int64_t kCus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return (int64_t)(n > 0 ? n : 256);
  }();
  // (round 4: slices are multiples of 128 rows -- pairs of the 8-phase kernel's 64-row tiles; 1.40 us per tile and 35 us per
  // round of workgroups fitted to tools/wgrad_tn_bench.py: 153 / 443 / 597 / 1391 us measured for 1 x 17 792, 5 and 6 x 8 896, 6 x 35 584 rows)
  // (the rows the slices do not cover -- fewer than 128 per slice -- ride in the last slice since the descriptor-bounded
  // kernel; the `rest` term, fitted when they were a library product of the caller's, still ranks the slice counts the way
  // the measurements do: 16 slices beat 15 / 11 at 0.8 - 1.25 M rows x 17408 x 1024 by 1 - 2 %, tools/joint_wgrad_rows.py)
  auto cost = [&](int64_t c) {
    const int64_t rounds = (tiles * c + kCus - 1) / kCus, steps = M / (128 * c) * 2, rest = M - c * steps * 64;
    return (double)rounds * ((double)steps * 1.40e-6 + 35e-6) + (double)(c * batch) * (double)(N * K) * 8.0 / 5e12 +
           (rest > 0 ? 15e-6 + (double)(rest * batch) * (double)(N * K) * 2.0 / 0.9e15 : 0.0);
  };
  int64_t s = 1;
  double best = 1e30;
  for (int64_t c = 1; c <= 32 && M / (128 * c) >= 2; ++c)
    if (cost(c) < best) best = cost(c), s = c;
  if (env_s > 0) s = env_s;
  while (s > 1 && M / (128 * s) < 2) --s;
  const int64_t per = M / (128 * s) * 128;
  const int64_t last = (M - (s - 1) * per + 127) / 128 * 128;   // the last slice, with the uncovered rows, in whole tile pairs
  if (per < 256 || last * N * 2 >= ((int64_t)1 << 32) || last * K * 2 >= ((int64_t)1 << 32)) return 0;
  if (rows_per_slice) *rows_per_slice = per;
  if (seconds) *seconds = cost(s);
  return (int)s;
}
}  // namespace
}  // namespace caiman

// Slices of M the kernel wants for `batch` products [M, N]^T x [M, K] of one shape on this chip (>= 1), and the rows they
// cover: rows_covered = slices * rows_per_slice with rows_per_slice a multiple of 32 (>= 128); 0 slices: shape not supported.
extern "C" int caiman_wgrad_tn_plan(int64_t M, int64_t N, int64_t K, int batch, int dtype, int64_t* rows_per_slice) {
  return caiman::wgrad_plan(M, N, K, batch, dtype, rows_per_slice, nullptr);
}
// What the plan's cost model expects the call to take, in microseconds (negative: shape not supported) -- for callers that
// have a library product to fall back on and want the kernel only where it is expected to win.
extern "C" double caiman_wgrad_tn_estimate_us(int64_t M, int64_t N, int64_t K, int batch, int dtype) {
  double sec = 0.0;
  return caiman::wgrad_plan(M, N, K, batch, dtype, nullptr, &sec) > 0 ? sec * 1e6 : -1.0;
}

// `batch` + `batch2` products of one shape in one launch: product p < batch at dY + p * stride_y / H + p * stride_h, product
// batch + q at dY2 + q * stride_y2 / H2 + q * stride_h2 (elements; rows of N / K elements, contiguous; batch2 = 0: one
// group).  slabs [batch + batch2][slices][N][K] fp32 (written, not accumulated): slab (p, s) = sum over rows
// [s * rows_per_slice, +rows_per_slice) of dY_p[m][n] * H_p[m][k]; the last slab also holds the rows from
// slices * rows_per_slice to M where caiman_wgrad_tn_covers_remainder() says so (else those rows are the caller's).
extern "C" int caiman_wgrad_tn2(const void* dY, int64_t stride_y, const void* H, int64_t stride_h, int batch, const void* dY2,
                                int64_t stride_y2, const void* H2, int64_t stride_h2, int batch2, float* slabs, int64_t M,
                                int64_t N, int64_t K, int slices, int64_t rows_per_slice, int dtype, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(batch >= 1 && batch2 >= 0, "wgrad_tn: batch >= 1, batch2 >= 0");
  CAIMAN_CHECK(caiman_wgrad_tn_plan(M, N, K, batch + batch2, dtype, nullptr) > 0, "wgrad_tn: bf16 / f16, N, K %% 256 == 0, M >= 128");
  CAIMAN_CHECK(slices >= 1 && rows_per_slice >= 96 && rows_per_slice % 32 == 0 && (int64_t)slices * rows_per_slice <= M,
               "wgrad_tn: slices x rows_per_slice must be multiples of 32 rows (>= 96) inside M");
  CAIMAN_CHECK(dY && H && slabs && (batch2 == 0 || (dY2 && H2)), "wgrad_tn: null pointer");
  auto al = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; };
  CAIMAN_CHECK(al(dY, 16) && al(H, 16) && al(slabs, 16) && stride_y % 8 == 0 && stride_h % 8 == 0 && stride_y >= 0 && stride_h >= 0,
               "wgrad_tn: 16-byte aligned operands and batch strides");
  CAIMAN_CHECK(batch2 == 0 || (al(dY2, 16) && al(H2, 16) && stride_y2 % 8 == 0 && stride_h2 % 8 == 0 && stride_y2 >= 0 && stride_h2 >= 0),
               "wgrad_tn: 16-byte aligned operands and batch strides (second group)");
  const int tiles_k = (int)(K / WBK), n_tiles = (int)(N / WBN) * tiles_k;
  const int64_t grid = (int64_t)n_tiles * slices * (batch + batch2);
  CAIMAN_CHECK(grid < ((int64_t)1 << 31), "wgrad_tn: too many tiles");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the 8-phase kernel wherever the slices are pairs of its 64-row tiles (what caiman_wgrad_tn_plan hands out);
  // CAIMAN_WGRAD_KERNEL=ring: the round-3 four-stage ring kernel (A/B)
  static const bool ring = std::getenv("CAIMAN_WGRAD_KERNEL") != nullptr && std::string(std::getenv("CAIMAN_WGRAD_KERNEL")) == "ring";
  const int64_t last_rows = (M - (int64_t)(slices - 1) * rows_per_slice + 127) / 128 * 128;
  if (!ring && rows_per_slice % 128 == 0 && rows_per_slice >= 256 && last_rows * N * 2 < ((int64_t)1 << 32) &&
      last_rows * K * 2 < ((int64_t)1 << 32)) {
    if (dtype == CAIMAN_BF16)
      hipLaunchKernelGGL((joint_wgrad8_kernel<bf16_t>), dim3((unsigned)grid), dim3(512), 0, s, (const bf16_t*)dY, (const bf16_t*)H,
                         slabs, (int)N, (int)K, (int)rows_per_slice, tiles_k, n_tiles, slices, stride_y, stride_h, batch,
                         (const bf16_t*)dY2, stride_y2, (const bf16_t*)H2, stride_h2, M);
    else
      hipLaunchKernelGGL((joint_wgrad8_kernel<f16_t>), dim3((unsigned)grid), dim3(512), 0, s, (const f16_t*)dY, (const f16_t*)H,
                         slabs, (int)N, (int)K, (int)rows_per_slice, tiles_k, n_tiles, slices, stride_y, stride_h, batch,
                         (const f16_t*)dY2, stride_y2, (const f16_t*)H2, stride_h2, M);
    return check_launch("transposed-read weight gradient (8-phase)");
  }
  if (dtype == CAIMAN_BF16)
    hipLaunchKernelGGL((joint_wgrad_kernel<bf16_t>), dim3((unsigned)grid), dim3(64 * WNW), 0, s, (const bf16_t*)dY, (const bf16_t*)H,
                       slabs, (int)N, (int)K, (int)rows_per_slice, tiles_k, n_tiles, slices, stride_y, stride_h, batch,
                       (const bf16_t*)dY2, stride_y2, (const bf16_t*)H2, stride_h2);
  else
    hipLaunchKernelGGL((joint_wgrad_kernel<f16_t>), dim3((unsigned)grid), dim3(64 * WNW), 0, s, (const f16_t*)dY, (const f16_t*)H,
                       slabs, (int)N, (int)K, (int)rows_per_slice, tiles_k, n_tiles, slices, stride_y, stride_h, batch,
                       (const f16_t*)dY2, stride_y2, (const f16_t*)H2, stride_h2);
  return check_launch("transposed-read weight gradient");
}
// 1 when a call with this plan also sums the rows behind slices * rows_per_slice (the 8-phase kernel: they ride in the last
// slice), 0 when they are the caller's (the round-3 ring kernel, or slices that are not pairs of 64-row tiles).
extern "C" int caiman_wgrad_tn_covers_remainder(int64_t M, int64_t N, int64_t K, int slices, int64_t rows_per_slice) {
  static const bool ring = std::getenv("CAIMAN_WGRAD_KERNEL") != nullptr && std::string(std::getenv("CAIMAN_WGRAD_KERNEL")) == "ring";
  if (ring || slices < 1 || rows_per_slice < 256 || rows_per_slice % 128 != 0 || (int64_t)slices * rows_per_slice > M) return 0;
  const int64_t last_rows = (M - (int64_t)(slices - 1) * rows_per_slice + 127) / 128 * 128;
  return last_rows * N * 2 < ((int64_t)1 << 32) && last_rows * K * 2 < ((int64_t)1 << 32) ? 1 : 0;
}
extern "C" int caiman_wgrad_tn(const void* dY, int64_t stride_y, const void* H, int64_t stride_h, float* slabs, int batch,
                               int64_t M, int64_t N, int64_t K, int slices, int64_t rows_per_slice, int dtype,
                               caiman_stream_t stream) {
  return caiman_wgrad_tn2(dY, stride_y, H, stride_h, batch, nullptr, 0, nullptr, 0, 0, slabs, M, N, K, slices, rows_per_slice,
                          dtype, stream);
}

// dst[i] += slabs[0][i] + slabs[1][i] + ... (slabs of n floats, one behind the other, added in order): the partial products of
// a sliced weight gradient straight into the parameter's fp32 `.grad`, one pass instead of a reduction and an accumulate.
namespace caiman {
namespace {
__global__ __launch_bounds__(256) void slab_accumulate_kernel(const float* __restrict__ slabs, int slices, int64_t n4,
                                                              float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 v = reinterpret_cast<const float4*>(slabs)[i];
  for (int s = 1; s < slices; ++s) {
    const float4 w = reinterpret_cast<const float4*>(slabs)[(int64_t)s * n4 + i];
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
  }
  float4 a = reinterpret_cast<float4*>(dst)[i];
  a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  reinterpret_cast<float4*>(dst)[i] = a;
}
}  // namespace
}  // namespace caiman
extern "C" int caiman_slab_accumulate(const float* slabs, int slices, int64_t n, float* dst, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(slabs && dst && slices >= 1 && slices <= 4096 && n >= 4 && n % 4 == 0, "slab_accumulate: 1 .. 4096 slabs of n %% 4 == 0 floats");
  CAIMAN_CHECK(((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0, "slab_accumulate: 16-byte aligned pointers");
  CAIMAN_CHECK((n / 4 + 255) / 256 < ((int64_t)1 << 31), "slab_accumulate: too many elements");
  hipLaunchKernelGGL(slab_accumulate_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     slabs, slices, n / 4, dst);
  return check_launch("slab accumulate");
}

// The joint projection's instance (one product): include/caiman_rnnt.h.
extern "C" int caiman_joint_fc_wgrad_plan(int64_t M, int64_t N, int64_t K, int dtype, int64_t* rows_per_slice) {
  return M < 512 ? 0 : caiman_wgrad_tn_plan(M, N, K, 1, dtype, rows_per_slice);
}
extern "C" int caiman_joint_fc_wgrad(const void* dY, const void* H, float* slabs, int64_t M, int64_t N, int64_t K, int slices,
                                     int64_t rows_per_slice, int dtype, caiman_stream_t stream) {
  return caiman_wgrad_tn(dY, 0, H, 0, slabs, 1, M, N, K, slices, rows_per_slice, dtype, stream);
}
