// Log-mel frontend on gfx950: pad -> dither -> pre-emphasis -> Hann frames -> 512-pt FFT power ->
// mel filterbank -> ln, and the (blended) per-feature normalisation.
//
// Replaces the third-party DALI operator chain the reference drives on this path
// (training/caiman_asr_train/data/dali/pipeline.py:260-315 construction, :439-462 graph tail;
// normalisation training/caiman_asr_train/data/dali/mel_normalization.py:103-118).  DALI is not
// vendored under /root/reference; operator semantics are restated from its documentation
//  and the golden tensor of the reference's own test
// (training/tests/data/dali/test_data_loader.py:235-258) is the intended pin.
//
// One wave64 per frame (8 frames per workgroup): the frame is windowed into LDS in bit-reversed
// order, 9 radix-2 stages run in LDS, lanes 0..nmel-1 then each fold one triangular filter.
// Window, twiddles and filter weights are tables built once on the host in double precision.
#include "common.h"

namespace caiman {
namespace {

constexpr int kFramesPerBlock = 8;
constexpr int kMelRun = 32;     // weights per mel filter kept in LDS (a triangular filter of the 80-band bank spans <= 24 bins of a 512-point FFT)

__device__ __forceinline__ float gauss_from(uint64_t seed, uint64_t idx) {
  // counter-based N(0,1): two splitmix64 uniforms -> Box-Muller
  auto mix = [](uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  const uint64_t a = mix(seed + 0x9E3779B97F4A7C15ull * (2 * idx + 1));
  const uint64_t b = mix(seed + 0x9E3779B97F4A7C15ull * (2 * idx + 2));
  const float u1 = ((float)(a >> 40) + 1.0f) * (1.0f / 16777217.0f);  // (0,1]
  const float u2 = (float)(b >> 40) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

struct MelParams {
  const float* audio;       // [B, max_samples]
  const int32_t* audio_len; // [B]
  const float* window;      // [win_len]
  const float* tw_cos;      // [nfft/2]
  const float* tw_sin;      // [nfft/2]
  const float* mel_w;       // [nmel, nfft/2+1] dense
  const int32_t* mel_lo;    // [nmel] first bin with non-zero weight
  const int32_t* mel_hi;    // [nmel] one past the last
  int64_t B, max_samples, max_frames;
  int win_len, hop, nfft, log2_nfft, nmel, pad;
  float preemph, dither, log_floor;
  uint64_t seed;
};

template <int NFFT>
__global__ __launch_bounds__(kFramesPerBlock* kWave) void logmel_kernel(MelParams p, float* __restrict__ out,
                                                                        int32_t* __restrict__ out_len) {
  __shared__ float re[kFramesPerBlock][NFFT];
  __shared__ float im[kFramesPerBlock][NFFT];
  extern __shared__ float sig[];        // the block's padded + dithered samples: (frames - 1) * hop + win_len + 1 of them,
  //                                       then the tables every wave walks: twiddles, window, the filters' non-zero runs
  const int w = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
  const int b = blockIdx.y;
  const int64_t frame0 = (int64_t)blockIdx.x * kFramesPerBlock, frame = frame0 + w;
  const int64_t n = p.audio_len[b];
  const int64_t total = n + p.pad;  // padded signal length
  const int64_t nframes = total >= p.win_len ? (total - p.win_len) / p.hop + 1 : 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) out_len[b] = (int32_t)nframes;
  const bool live = frame < nframes;
  const float* x = p.audio + (int64_t)b * p.max_samples;

  // ---- the samples the block's frames touch, ONCE (round 4: every frame used to evaluate its 2 x 400 dithered samples --
  // a counter-based Gaussian each -- itself, 4.2 x the work; values and arithmetic per element are unchanged)
  const int nsig = (kFramesPerBlock - 1) * p.hop + p.win_len + 1;
  // tables in LDS (round 4: the radix-2 stages fetched two twiddles per butterfly and the filters one weight per bin from
  // GLOBAL memory inside dependent loops -- 12 us per frame-wave for 50 kFLOP): cos / sin [NFFT / 2], window [win_len],
  // and per filter the first kMelRun weights of its non-zero run (longer runs read the rest from global memory)
  float* twc = sig + ((nsig + 3) & ~3);
  float* tws = twc + NFFT / 2;
  float* win = tws + NFFT / 2;
  float* melw = win + ((p.win_len + 3) & ~3);                 // [nmel][kMelRun]
  for (int i = threadIdx.x; i < NFFT / 2; i += kFramesPerBlock * kWave) {
    twc[i] = p.tw_cos[i];
    tws[i] = p.tw_sin[i];
  }
  for (int i = threadIdx.x; i < p.win_len; i += kFramesPerBlock * kWave) win[i] = p.window[i];
  for (int i = threadIdx.x; i < p.nmel * kMelRun; i += kFramesPerBlock * kWave) {
    const int m = i / kMelRun, k = p.mel_lo[m] + (i - m * kMelRun);
    melw[i] = k < p.mel_hi[m] ? p.mel_w[(int64_t)m * (NFFT / 2 + 1) + k] : 0.f;
  }
  const int64_t q0 = frame0 * p.hop - 1;                       // sig[i] = padded signal at index q0 + i (clamped at 0)
  for (int i = threadIdx.x; i < nsig; i += kFramesPerBlock * kWave) {
    int64_t q = q0 + i;
    if (q < 0) q = 0;                      // PreemphasisFilter border = clamp
    float v = (q >= p.pad && q - p.pad < n) ? x[q - p.pad] : 0.f;
    if (p.dither != 0.f) v += p.dither * gauss_from(p.seed, (uint64_t)b * (uint64_t)(p.max_samples + p.pad) + (uint64_t)q);
    sig[i] = v;
  }
  __syncthreads();
  // ---- window into LDS, bit-reversed -----------------------------------------------------------
  const float* sw = sig + w * p.hop;       // sw[i + 1] = sample(frame * hop + i), sw[i] = the sample before it
  for (int i = lane; i < NFFT; i += kWave) {
    float v = 0.f;
    if (live && i < p.win_len) v = (sw[i + 1] - p.preemph * sw[i]) * win[i];
    const int j = __brev((unsigned)i) >> (32 - p.log2_nfft);
    re[w][j] = v;
    im[w][j] = 0.f;
  }
  __syncthreads();
  // ---- radix-2 DIT --------------------------------------------------------------------------
  for (int s = 0; s < p.log2_nfft; ++s) {
    const int half = 1 << s;
    const int tstride = NFFT >> (s + 1);
    for (int k = lane; k < NFFT / 2; k += kWave) {
      const int grp = k >> s, pos = k & (half - 1);
      const int i0 = (grp << (s + 1)) + pos, i1 = i0 + half;
      const float c = twc[pos * tstride], sn = tws[pos * tstride];  // e^{-2 pi i pos / (2 half)}
      const float ar = re[w][i0], ai = im[w][i0], br = re[w][i1], bi = im[w][i1];
      const float tr = br * c + bi * sn, ti = bi * c - br * sn;
      re[w][i0] = ar + tr; im[w][i0] = ai + ti;
      re[w][i1] = ar - tr; im[w][i1] = ai - ti;
    }
    __syncthreads();
  }
  // ---- power spectrum (bins 0..NFFT/2) into re[] ---------------------------------------------------
  float pw[(NFFT / 2 + kWave) / kWave];
#pragma unroll
  for (int q = 0; q < (NFFT / 2 + kWave) / kWave; ++q) {
    const int k = lane + q * kWave;
    pw[q] = (k <= NFFT / 2) ? re[w][k] * re[w][k] + im[w][k] * im[w][k] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < (NFFT / 2 + kWave) / kWave; ++q) {
    const int k = lane + q * kWave;
    if (k <= NFFT / 2) re[w][k] = pw[q];
  }
  __syncthreads();
  // ---- mel + log: staged per block in im[] as [mel][frame of the block], then written as runs of consecutive frames ------
  float* stage = &im[0][0];                // 8 x NFFT floats: room for nmel <= NFFT rows of 8
  {
    const int nb = NFFT / 2 + 1;
    for (int m = lane; m < p.nmel; m += kWave) {
      float acc = 0.f;
      if (live) {
        const float* wrow = p.mel_w + (int64_t)m * nb;
        const int lo = p.mel_lo[m], hi = p.mel_hi[m], run = min(hi, lo + kMelRun);
        for (int k = lo; k < run; ++k) acc += melw[m * kMelRun + (k - lo)] * re[w][k];      // same products, same order
        for (int k = run; k < hi; ++k) acc += wrow[k] * re[w][k];
        acc = logf(fmaxf(acc, p.log_floor));
      }
      stage[m * kFramesPerBlock + w] = live ? acc : 0.f;  // Pad: fill 0
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < p.nmel * kFramesPerBlock; i += kFramesPerBlock * kWave) {
    const int m = i / kFramesPerBlock, fr = i - m * kFramesPerBlock;
    if (frame0 + fr < p.max_frames) out[((int64_t)b * p.nmel + m) * p.max_frames + frame0 + fr] = stage[i];
  }
}

// (x - mean)/std over the valid frames of each (utterance, mel bin) row, optionally blended with
// dataset statistics: r*(x-mu_d)/sd_d + (1-r)*(x-mu_u)/sd_u.  Population std (ddof = 0).
__global__ __launch_bounds__(256) void mel_normalize_kernel(float* __restrict__ x, const int32_t* __restrict__ len,
                                                            int64_t T, int nmel, const float* __restrict__ ds_mean,
                                                            const float* __restrict__ ds_std, float ratio) {
  __shared__ double sm[256 / kWave];
  const int b = blockIdx.y, m = blockIdx.x;
  float* row = x + ((int64_t)b * nmel + m) * T;
  const int64_t n = len[b];
  double s = 0;
  for (int64_t t = threadIdx.x; t < n; t += 256) s += row[t];
  s = block_reduce<256 / kWave>(s, [](double a, double c) { return a + c; }, sm);
  const double mean = n > 0 ? s / (double)n : 0.0;
  double v = 0;
  for (int64_t t = threadIdx.x; t < n; t += 256) { const double d = row[t] - mean; v += d * d; }
  v = block_reduce<256 / kWave>(v, [](double a, double c) { return a + c; }, sm);
  const float sd = n > 0 ? (float)sqrt(v / (double)n) : 1.f;
  const float mu = (float)mean;
  const float r = ds_mean ? ratio : 0.f;
  const float dm = ds_mean ? ds_mean[m] : 0.f, dsd = ds_std ? ds_std[m] : 1.f;
  for (int64_t t = threadIdx.x; t < T; t += 256) {
    if (t < n) {
      const float xv = row[t];
      float o = 0.f;
      if (r < 1.f) o += (1.f - r) * (xv - mu) / sd;
      if (r > 0.f) o += r * (xv - dm) / dsd;
      row[t] = o;
    } else {
      row[t] = 0.f;
    }
  }
}


// ---------------------------------------------------------------------------
// SpecAugment masks applied + frame splicing + PermuteAudio in one pass:
//   out[t1][b][n * F + f] = masked(x[b][f][t1 * subsampling + n])      n < stacking, zero past T
// where `masked` zeroes frequency rows inside any [f0, f0 + fw) and frames inside any [t0, t0 + tw) of utterance b
// (training/caiman_asr_train/data/features.py:34-115 `SpecAugment`, :118-139 `stack_subsample_frames`, :160-162
// `PermuteAudio`).  As torch ops this was masked_fill + two cats + a strided slice + a permuted copy (five passes over up to
// 51 MB) behind a dozen small launches that built the boolean mask; the mask GEOMETRY (a few numbers per utterance) stays
// where it was drawn.  A workgroup takes 32 output frames of one utterance: reads the (32 - 1) * subsampling + stacking input
// frames of all F rows along t (coalesced) into LDS, writes 32 rows of F * stacking values (coalesced).
// ---------------------------------------------------------------------------
constexpr int kSpliceFrames = 32;
__global__ __launch_bounds__(256) void specaug_splice_kernel(const float* __restrict__ x, int64_t F, int64_t T,
                                                             const float* __restrict__ f0, const float* __restrict__ fw, int nf,
                                                             const float* __restrict__ t0, const float* __restrict__ tw, int nt,
                                                             int stacking, int subsampling, int64_t T_out, int64_t B,
                                                             float* __restrict__ out) {
  extern __shared__ float sm[];
  const int span = (kSpliceFrames - 1) * subsampling + stacking;
  const int pitch = span | 1;                       // odd: the F-strided reads of the store phase hit distinct banks
  float* tile = sm;                                 // [F][pitch]
  unsigned char* fmask = reinterpret_cast<unsigned char*>(sm + F * pitch);   // [F]
  unsigned char* tmask = fmask + F;                                           // [span]
  const int b = blockIdx.y, tid = threadIdx.x;
  const int64_t t1_0 = (int64_t)blockIdx.x * kSpliceFrames, tb = t1_0 * subsampling;
  for (int f = tid; f < F; f += blockDim.x) {
    bool m = false;
    for (int i = 0; i < nf; ++i) {
      const float a = f0[(int64_t)b * nf + i];
      m |= (float)f >= a && (float)f < a + fw[(int64_t)b * nf + i];
    }
    fmask[f] = m;
  }
  for (int d = tid; d < span; d += blockDim.x) {
    const float t = (float)(tb + d);
    bool m = false;
    for (int i = 0; i < nt; ++i) {
      const float a = t0[(int64_t)b * nt + i];
      m |= t >= a && t < a + tw[(int64_t)b * nt + i];
    }
    tmask[d] = m;
  }
  __syncthreads();
  const float* xb = x + (int64_t)b * F * T;
  for (int idx = tid; idx < F * span; idx += blockDim.x) {
    const int f = idx / span, d = idx - f * span;
    const int64_t t = tb + d;
    float v = t < T ? xb[(int64_t)f * T + t] : 0.f;
    if (fmask[f] | tmask[d]) v = 0.f;
    tile[f * pitch + d] = v;
  }
  __syncthreads();
  const int C = (int)F * stacking;
  for (int idx = tid; idx < kSpliceFrames * C; idx += blockDim.x) {
    const int tt = idx / C, c = idx - tt * C;
    const int n = c / (int)F, f = c - n * (int)F;
    const int64_t t1 = t1_0 + tt;
    if (t1 < T_out) out[(t1 * B + b) * C + c] = tile[f * pitch + tt * subsampling + n];
  }
}

// SpecAugment mask geometry from uniform draws, one thread per (utterance, mask index): float arithmetic in the order
// data/features.py::SpecAugment.geometry_from_draws performs it (torch.floor / torch.round = floorf / rintf), so the two
// agree bit for bit on the same draws.  rnd [B][2 nf + 2 nt]: columns fw | f0 | tw | t0;  out: fw [B][nf], f0 [B][nf],
// tw [B][nt], t0 [B][nt], one after the other.
__global__ __launch_bounds__(256) void specaug_geometry_kernel(const float* __restrict__ rnd, const void* __restrict__ lens,
                                                               int lens_kind, int B, float F, float T, int nf, float min_freq,
                                                               float freq_span, float time_masks, int nt, float min_time,
                                                               float max_time, float* __restrict__ out) {
  const int per = nf > nt ? nf : nt;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * per) return;
  const int b = idx / per, j = idx % per;
  const int cols = 2 * nf + 2 * nt;
  const float* r = rnd + (int64_t)b * cols;
  float* fw = out;
  float* f0 = fw + (int64_t)B * nf;
  float* tw = f0 + (int64_t)B * nf;
  float* t0 = tw + (int64_t)B * nt;
  if (j < nf) {
    const float w = floorf(r[j] * freq_span) + min_freq;
    const float room = fmaxf((F - w) + 1.f, 1.f);
    fw[(int64_t)b * nf + j] = w;
    f0[(int64_t)b * nf + j] = floorf(r[nf + j] * room);
  }
  if (j < nt) {
    const float len = lens_kind == 0 ? (float)static_cast<const int32_t*>(lens)[b]
                      : lens_kind == 1 ? (float)static_cast<const int64_t*>(lens)[b] : static_cast<const float*>(lens)[b];
    const float n_masks = (time_masks > 0.f && time_masks < 1.f) ? rintf(len * time_masks) : time_masks;
    const float max_t = (max_time > 0.f && max_time < 1.f) ? rintf(len * max_time) : max_time;
    const float w = floorf(r[2 * nf + j] * ((max_t - min_time) + 1.f)) + min_time;
    const float room = fmaxf((T - w) + 1.f, 1.f);
    tw[(int64_t)b * nt + j] = (float)j < n_masks ? w : 0.f;
    t0[(int64_t)b * nt + j] = floorf(r[2 * nf + nt + j] * room);
  }
}

}  // namespace
}  // namespace caiman

extern "C" int caiman_logmel_forward(const float* audio, const int32_t* audio_len, int64_t B, int64_t max_samples,
                                     int win_len, int hop, int nfft, int nmel, int initial_pad, float preemph,
                                     float dither, uint64_t seed, float log_floor, const float* window,
                                     const float* tw_cos, const float* tw_sin, const float* mel_w,
                                     const int32_t* mel_lo, const int32_t* mel_hi, float* out, int32_t* out_len,
                                     int64_t max_frames, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(B >= 1 && B <= 65535 && max_samples >= 1 && max_frames >= 1, "logmel_forward: bad extents");
  CAIMAN_CHECK(win_len >= 1 && hop >= 1 && win_len <= nfft && nmel >= 1, "logmel_forward: bad window / hop / nfft");
  CAIMAN_CHECK(nfft == 256 || nfft == 512 || nfft == 1024, "logmel_forward: nfft must be 256, 512 or 1024 (got %d)", nfft);
  CAIMAN_CHECK(audio && audio_len && window && tw_cos && tw_sin && mel_w && mel_lo && mel_hi && out && out_len,
               "logmel_forward: null pointer");
  int lg = 0;
  while ((1 << lg) < nfft) ++lg;
  MelParams p{audio, audio_len, window, tw_cos, tw_sin, mel_w, mel_lo, mel_hi, B, max_samples, max_frames,
              win_len, hop, nfft, lg, nmel, initial_pad, preemph, dither, log_floor, seed};
  const dim3 grid((unsigned)((max_frames + kFramesPerBlock - 1) / kFramesPerBlock), (unsigned)B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t sig_bytes = (size_t)((((kFramesPerBlock - 1) * hop + win_len + 1 + 3) & ~3) + nfft + ((win_len + 3) & ~3) +
                                    nmel * kMelRun) * sizeof(float);
  CAIMAN_CHECK(nmel <= nfft && sig_bytes + (size_t)2 * kFramesPerBlock * nfft * sizeof(float) <= 150 * 1024,
               "logmel_forward: nmel <= nfft and %d frames of hop %d / window %d must fit the block's sample buffer",
               kFramesPerBlock, hop, win_len);
#define CAIMAN_LOGMEL(NF)                                                                                                  \
  do {                                                                                                                     \
    auto kern = logmel_kernel<NF>;                                                                                         \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,              \
                            (int)sig_bytes) != hipSuccess)                                                                 \
      return check_launch("logmel_forward attribute");                                                                     \
    hipLaunchKernelGGL(kern, grid, dim3(kFramesPerBlock * kWave), sig_bytes, st, p, out, out_len);                         \
  } while (0)
  if (nfft == 256) CAIMAN_LOGMEL(256);
  else if (nfft == 512) CAIMAN_LOGMEL(512);
  else CAIMAN_LOGMEL(1024);
#undef CAIMAN_LOGMEL
  return check_launch("caiman_logmel_forward");
}

extern "C" int caiman_mel_normalize(float* x, const int32_t* len, int64_t B, int nmel, int64_t T, const float* ds_mean,
                                    const float* ds_std, float ratio, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(B >= 1 && B <= 65535 && nmel >= 1 && T >= 1, "mel_normalize: bad extents");
  CAIMAN_CHECK(x && len, "mel_normalize: null pointer");
  CAIMAN_CHECK(ratio >= 0.f && ratio <= 1.f, "mel_normalize: ratio must be in [0,1]");
  CAIMAN_CHECK(ratio == 0.f || (ds_mean && ds_std), "mel_normalize: dataset statistics required when ratio > 0");
  hipLaunchKernelGGL(mel_normalize_kernel, dim3((unsigned)nmel, (unsigned)B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, len, T, nmel, ds_mean, ds_std, ratio);
  return check_launch("caiman_mel_normalize");
}

extern "C" int caiman_specaug_geometry(const float* rnd, const void* lens, int lens_kind, int64_t B, int64_t F, int64_t T,
                                       int nf, float min_freq, float freq_span, float time_masks, int nt, float min_time,
                                       float max_time, float* out, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(B >= 1 && B <= 65535 && F >= 1 && T >= 1, "specaug_geometry: bad extents");
  CAIMAN_CHECK(nf >= 0 && nt >= 0 && nf + nt >= 1 && nf <= 1024 && nt <= 65536, "specaug_geometry: mask counts");
  CAIMAN_CHECK(rnd && out && (nt == 0 || lens), "specaug_geometry: null pointer");
  CAIMAN_CHECK(lens_kind >= 0 && lens_kind <= 2, "specaug_geometry: lens_kind 0 (int32), 1 (int64) or 2 (float)");
  const int64_t n = B * (nf > nt ? nf : nt);
  hipLaunchKernelGGL(specaug_geometry_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     rnd, lens, lens_kind, (int)B, (float)F, (float)T, nf, min_freq, freq_span, time_masks, nt,
                     min_time, max_time, out);
  return check_launch("caiman_specaug_geometry");
}

extern "C" int caiman_specaug_splice(const float* x, int64_t B, int64_t F, int64_t T, const float* f0, const float* fw, int nf,
                                     const float* t0, const float* tw, int nt, int stacking, int subsampling, int64_t T_out,
                                     float* out, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(B >= 1 && B <= 65535 && F >= 1 && F <= 1024 && T >= 1, "specaug_splice: bad extents");
  CAIMAN_CHECK(stacking >= 1 && stacking <= 16 && subsampling >= 1 && subsampling <= 16, "specaug_splice: stacking / subsampling in [1, 16]");
  CAIMAN_CHECK(T_out >= 1 && T_out <= (T + subsampling - 1) / subsampling, "specaug_splice: T_out must be in [1, ceil(T / subsampling)]");
  CAIMAN_CHECK(nf >= 0 && nt >= 0 && (nf == 0 || (f0 && fw)) && (nt == 0 || (t0 && tw)), "specaug_splice: mask arrays");
  CAIMAN_CHECK(x && out, "specaug_splice: null pointer");
  const int span = (kSpliceFrames - 1) * subsampling + stacking, pitch = span | 1;
  const size_t lds = (size_t)F * pitch * sizeof(float) + (size_t)F + (size_t)span + 8;
  CAIMAN_CHECK(lds <= 64 * 1024, "specaug_splice: %lld feature rows x %d frames do not fit the tile", (long long)F, span);
  const dim3 grid((unsigned)((T_out + kSpliceFrames - 1) / kSpliceFrames), (unsigned)B);
  hipLaunchKernelGGL(specaug_splice_kernel, grid, dim3(256), lds, static_cast<hipStream_t>(stream), x, F, T, f0, fw, nf, t0, tw,
                     nt, stacking, subsampling, T_out, B, out);
  return check_launch("caiman_specaug_splice");
}
