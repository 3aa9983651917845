// Device half of the beam search (include/caiman_beam.h, part 1): for every pending hypothesis, the k best
// entries of log_softmax(logits / T) after the EOS correction, plus the blank log-probability.
//
// Replaces the torch op chain the reference launches per expansion round
// (training/caiman_asr_train/rnnt/decoder.py:159-172 `_joint_step`: divide, log_softmax, EOS correction;
// beam.py:535-546: topk, max, compare, boolean-index, column slice, D2H of each) with one kernel whose
// outputs have a fixed shape, so a single copy brings a round's results to the host.
//
// One 256-thread workgroup per row.  The row (8704 logits = 17-35 KB) is read three times -- max, sum of
// exponentials, selection -- and only the first read reaches HBM.  Selection: every thread keeps the k best
// of its strided slice in registers, then k rounds of a workgroup arg-max merge the 256 sorted lists.
#include <cstdlib>

#include "common.h"
#include "../../include/caiman_beam.h"

namespace caiman {
namespace {

constexpr int kTopkThreads = 256;
constexpr int kMaxK = 8;

struct Cand {
  float v;
  int32_t i;
};

__device__ __forceinline__ bool better(float av, int32_t ai, float bv, int32_t bi) {
  return av > bv || (av == bv && ai < bi);  // ties: lower token id
}

struct TopkParams {
  int64_t n, vocab, row_stride;
  float temp;
  int32_t blank_idx, eos_mode, eos_idx;
  float eos_alpha, eos_log_beta;  // log(beta), or -inf when beta <= 0
  int32_t k;
};

template <typename T>
__global__ __launch_bounds__(kTopkThreads) void beam_topk_kernel(const T* __restrict__ logits, TopkParams p,
                                                                float* __restrict__ top_scores,
                                                                int32_t* __restrict__ top_tokens,
                                                                float* __restrict__ blank_logp) {
  __shared__ float red_f[kTopkThreads / kWave];
  __shared__ Cand red_c[kTopkThreads / kWave];
  const int64_t row = blockIdx.x;
  const T* x = logits + row * p.row_stride;
  const int tid = threadIdx.x;
  const float NEG_INF = -INFINITY;

  // ---- max and sum of exponentials of x / T ------------------------------------------------------------
  float m = NEG_INF;
  for (int64_t j = tid; j < p.vocab; j += kTopkThreads) m = fmaxf(m, (float)x[j] / p.temp);
  m = block_reduce<kTopkThreads / kWave>(m, [](float a, float b) { return fmaxf(a, b); }, red_f);
  float s = 0.f;
  for (int64_t j = tid; j < p.vocab; j += kTopkThreads) s += expf((float)x[j] / p.temp - m);
  s = block_reduce<kTopkThreads / kWave>(s, [](float a, float b) { return a + b; }, red_f);
  const float lse = m + logf(s);

  // ---- EOS correction touches two columns only ------------------------------------------------------------
  float lp_blank = (float)x[p.blank_idx] / p.temp - lse;
  float lp_eos = NEG_INF;
  if (p.eos_mode != CAIMAN_EOS_NONE) {
    lp_eos = (float)x[p.eos_idx] / p.temp - lse;
    if (p.eos_mode == CAIMAN_EOS_IGNORE) {
      lp_eos = NEG_INF;
    } else if (p.eos_mode == CAIMAN_EOS_BLANK) {
      const float hi = fmaxf(lp_blank, lp_eos), lo = fminf(lp_blank, lp_eos);
      lp_blank = hi == NEG_INF ? NEG_INF : hi + log1pf(expf(lo - hi));
      lp_eos = NEG_INF;
    } else {
      lp_eos *= p.eos_alpha;
      if (!(lp_eos > p.eos_log_beta)) lp_eos = NEG_INF;
    }
  }
  auto logp = [&](int64_t j) -> float {
    if (j == p.blank_idx) return lp_blank;
    if (p.eos_mode != CAIMAN_EOS_NONE && j == p.eos_idx) return lp_eos;
    return (float)x[j] / p.temp - lse;
  };

  // ---- thread-local k best (sorted, best first) -----------------------------------------------------------
  Cand mine[kMaxK];
#pragma unroll
  for (int q = 0; q < kMaxK; ++q) mine[q] = {NEG_INF, INT32_MAX};
  for (int64_t j = tid; j < p.vocab; j += kTopkThreads) {
    Cand c{logp(j), (int32_t)j};
    if (c.v != c.v) c.v = NEG_INF;  // NaN never wins
#pragma unroll
    for (int q = 0; q < kMaxK; ++q) {
      if (q < p.k && better(c.v, c.i, mine[q].v, mine[q].i)) {
        const Cand t = mine[q];
        mine[q] = c;
        c = t;
      }
    }
  }
  // ---- merge: k rounds of arg-max over the heads of the 256 lists ------------------------------------------
  int head = 0;
  for (int r = 0; r < p.k; ++r) {
    Cand c{NEG_INF, INT32_MAX};
#pragma unroll
    for (int q = 0; q < kMaxK; ++q)
      if (q == head) c = mine[q];
    Cand w = c;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
      const float ov = __shfl_xor(w.v, off, kWave);
      const int32_t oi = __shfl_xor(w.i, off, kWave);
      if (better(ov, oi, w.v, w.i)) w = {ov, oi};
    }
    if ((tid & (kWave - 1)) == 0) red_c[tid / kWave] = w;
    __syncthreads();
    Cand g = red_c[0];
#pragma unroll
    for (int q = 1; q < kTopkThreads / kWave; ++q)
      if (better(red_c[q].v, red_c[q].i, g.v, g.i)) g = red_c[q];
    if (c.i == g.i && g.i != INT32_MAX) ++head;  // token ids are unique: exactly one thread owns the winner
    if (tid == 0) {
      top_scores[row * p.k + r] = g.v;
      top_tokens[row * p.k + r] = g.i == INT32_MAX ? 0 : g.i;
    }
    __syncthreads();
  }
  if (tid == 0) blank_logp[row] = lp_blank;
}


// Fast path (16-bit logits, vocab % 8 == 0, vocab <= 8 * 256 * kRegChunks, 16-byte aligned rows, k <= 4): a beam round at
// 2 000 streams has ~200 pending rows, fewer workgroups than CUs, so the kernel above is a pure latency chain -- 3 passes of
// 34 scalar loads per thread with an fp32 division each, and a predicated 8-slot insertion per element: 42 us of a ~160 us
// round (profiles/r03_decode_summary.md, before).  Here the row is read ONCE with 16-byte loads into registers (x / T kept as
// fp32: 40 values per thread), the maximum, the sum of exponentials and the selection run from registers, and the insertion
// list has exactly K slots.  Same arithmetic per element (x / T, expf, logf) and the same tie rule, so the selected tokens
// and scores are those of the kernel above.
constexpr int kRegChunks = 5;   // 16-byte pieces per thread: vocab <= 10 240

template <typename T, int K>
__global__ __launch_bounds__(kTopkThreads) void beam_topk_reg_kernel(const T* __restrict__ logits, TopkParams p,
                                                                    float* __restrict__ top_scores,
                                                                    int32_t* __restrict__ top_tokens,
                                                                    float* __restrict__ blank_logp) {
  using v8 = __attribute__((ext_vector_type(8))) T;
  __shared__ float red_f[kTopkThreads / kWave];
  __shared__ Cand red_c[kTopkThreads / kWave];
  const int64_t row = blockIdx.x;
  const T* x = logits + row * p.row_stride;
  const int tid = threadIdx.x;
  const float NEG_INF = -INFINITY;
  const int chunks = (int)(p.vocab >> 3);

  float v[kRegChunks][8];
  v8 raw[kRegChunks];
#pragma unroll
  for (int c = 0; c < kRegChunks; ++c) {
    const int ch = tid + c * kTopkThreads;
    if (ch < chunks) raw[c] = *reinterpret_cast<const v8*>(x + (int64_t)ch * 8);
  }
  float m = NEG_INF;
#pragma unroll
  for (int c = 0; c < kRegChunks; ++c) {
    const bool on = tid + c * kTopkThreads < chunks;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[c][e] = on ? (float)raw[c][e] / p.temp : NEG_INF;
      m = fmaxf(m, v[c][e]);
    }
  }
  m = block_reduce<kTopkThreads / kWave>(m, [](float a, float b) { return fmaxf(a, b); }, red_f);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < kRegChunks; ++c) {
    const bool on = tid + c * kTopkThreads < chunks;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += on ? expf(v[c][e] - m) : 0.f;
  }
  s = block_reduce<kTopkThreads / kWave>(s, [](float a, float b) { return a + b; }, red_f);
  const float lse = m + logf(s);

  float lp_blank = (float)x[p.blank_idx] / p.temp - lse;
  float lp_eos = NEG_INF;
  if (p.eos_mode != CAIMAN_EOS_NONE) {
    lp_eos = (float)x[p.eos_idx] / p.temp - lse;
    if (p.eos_mode == CAIMAN_EOS_IGNORE) {
      lp_eos = NEG_INF;
    } else if (p.eos_mode == CAIMAN_EOS_BLANK) {
      const float hi = fmaxf(lp_blank, lp_eos), lo = fminf(lp_blank, lp_eos);
      lp_blank = hi == NEG_INF ? NEG_INF : hi + log1pf(expf(lo - hi));
      lp_eos = NEG_INF;
    } else {
      lp_eos *= p.eos_alpha;
      if (!(lp_eos > p.eos_log_beta)) lp_eos = NEG_INF;
    }
  }

  Cand mine[K];
#pragma unroll
  for (int q = 0; q < K; ++q) mine[q] = {NEG_INF, INT32_MAX};
#pragma unroll
  for (int c = 0; c < kRegChunks; ++c) {
    const int ch = tid + c * kTopkThreads;
    if (ch < chunks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int32_t j = ch * 8 + e;
        Cand cd{v[c][e] - lse, j};
        if (j == p.blank_idx) cd.v = lp_blank;
        if (p.eos_mode != CAIMAN_EOS_NONE && j == p.eos_idx) cd.v = lp_eos;
        if (cd.v != cd.v) cd.v = NEG_INF;  // NaN never wins
#pragma unroll
        for (int q = 0; q < K; ++q) {
          if (q < p.k && better(cd.v, cd.i, mine[q].v, mine[q].i)) {
            const Cand t = mine[q];
            mine[q] = cd;
            cd = t;
          }
        }
      }
    }
  }
  int head = 0;
  for (int r = 0; r < p.k; ++r) {
    Cand c{NEG_INF, INT32_MAX};
#pragma unroll
    for (int q = 0; q < K; ++q)
      if (q == head) c = mine[q];
    Cand w = c;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
      const float ov = __shfl_xor(w.v, off, kWave);
      const int32_t oi = __shfl_xor(w.i, off, kWave);
      if (better(ov, oi, w.v, w.i)) w = {ov, oi};
    }
    if ((tid & (kWave - 1)) == 0) red_c[tid / kWave] = w;
    __syncthreads();
    Cand g = red_c[0];
#pragma unroll
    for (int q = 1; q < kTopkThreads / kWave; ++q)
      if (better(red_c[q].v, red_c[q].i, g.v, g.i)) g = red_c[q];
    if (c.i == g.i && g.i != INT32_MAX) ++head;
    if (tid == 0) {
      top_scores[row * p.k + r] = g.v;
      top_tokens[row * p.k + r] = g.i == INT32_MAX ? 0 : g.i;
    }
    __syncthreads();
  }
  if (tid == 0) blank_logp[row] = lp_blank;
}

}  // namespace
}  // namespace caiman

extern "C" int caiman_beam_topk(const void* logits, int64_t n, int64_t vocab, int64_t row_stride, int dtype,
                                float temperature, int32_t blank_idx, int eos_mode, int32_t eos_idx, float eos_alpha,
                                float eos_beta, int32_t k, float* top_scores, int32_t* top_tokens, float* blank_logp,
                                caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(n >= 0 && vocab >= 1 && row_stride >= vocab, "beam_topk: bad extents (n=%lld vocab=%lld stride=%lld)",
               (long long)n, (long long)vocab, (long long)row_stride);
  CAIMAN_CHECK(n <= 0x7fffffffLL, "beam_topk: too many rows");
  CAIMAN_CHECK(k >= 1 && k <= kMaxK && k <= vocab, "beam_topk: k must be in [1, %d] and <= vocab (got %d)", kMaxK, k);
  CAIMAN_CHECK(temperature > 0.f, "beam_topk: temperature must be positive");
  CAIMAN_CHECK(blank_idx >= 0 && blank_idx < vocab, "beam_topk: blank index %d outside [0, %lld)", blank_idx, (long long)vocab);
  CAIMAN_CHECK(eos_mode >= CAIMAN_EOS_NONE && eos_mode <= CAIMAN_EOS_PREDICT, "beam_topk: unknown EOS mode %d", eos_mode);
  CAIMAN_CHECK(eos_mode == CAIMAN_EOS_NONE || (eos_idx >= 0 && eos_idx < vocab && eos_idx != blank_idx),
               "beam_topk: EOS index %d outside [0, %lld) or equal to blank", eos_idx, (long long)vocab);
  if (n == 0) return CAIMAN_OK;
  CAIMAN_CHECK(logits && top_scores && top_tokens && blank_logp, "beam_topk: null pointer");
  TopkParams p{n, vocab, row_stride, temperature, blank_idx, eos_mode, eos_idx, eos_alpha,
               eos_beta > 0.f ? logf(eos_beta) : -INFINITY, k};
  hipStream_t st = static_cast<hipStream_t>(stream);
  return CAIMAN_DISPATCH(dtype, "beam_topk", [&]() -> int {
    if constexpr (std::is_same<scalar_t, double>::value) {
      set_error("beam_topk: f64 logits are not supported");
      return CAIMAN_ERR_UNSUPPORTED;
    } else {
      if constexpr (sizeof(scalar_t) == 2) {
        static const bool reg_path = !(std::getenv("CAIMAN_TOPK_REG") && std::atoi(std::getenv("CAIMAN_TOPK_REG")) == 0);
        if (reg_path && k <= 4 && vocab % 8 == 0 && vocab <= 8 * kTopkThreads * kRegChunks && row_stride % 8 == 0 &&
            (reinterpret_cast<uintptr_t>(logits) & 15u) == 0) {
          hipLaunchKernelGGL((beam_topk_reg_kernel<scalar_t, 4>), dim3((unsigned)n), dim3(kTopkThreads), 0, st,
                             static_cast<const scalar_t*>(logits), p, top_scores, top_tokens, blank_logp);
          return check_launch("caiman_beam_topk");
        }
      }
      hipLaunchKernelGGL((beam_topk_kernel<scalar_t>), dim3((unsigned)n), dim3(kTopkThreads), 0, st,
                         static_cast<const scalar_t*>(logits), p, top_scores, top_tokens, blank_logp);
      return check_launch("caiman_beam_topk");
    }
  });
}
