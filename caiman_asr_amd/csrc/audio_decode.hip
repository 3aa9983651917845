// Host-side audio decode for the data feed (include/caiman_data.h): FLAC (RFC 9639) and RIFF/WAVE -> mono f32.
//
// The reference decodes with DALI's `readers.file` + `decoders.audio(downmix=True, dtype=FLOAT)`
// (training/caiman_asr_train/data/dali/pipeline.py:253-259,400-414), i.e. libsndfile / libFLAC inside DALI -- third
// party, not vendored.  This is a from-the-specification decoder: no device code (the file is a .hip only so that
// the library's one build rule picks it up); a batch of files is decoded by a few host threads straight into the
// pinned staging buffer the caller passes, from where one H2D copy feeds the log-mel kernel.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/caiman_data.h"
#include "common.h"

namespace caiman {
namespace audio {

struct Error {
  std::string msg;
};

// ---- big-endian bit reader ------------------------------------------------------------------------------------
class Bits {
 public:
  Bits(const uint8_t* d, size_t n) : d_(d), n_(n) {}
  size_t byte_pos() const { return (pos_ + 7) >> 3; }
  void seek_byte(size_t b) { pos_ = b * 8; }
  uint64_t read(int n) {  // n <= 57
    if (n == 0) return 0;
    if (pos_ + (size_t)n > n_ * 8) throw Error{"unexpected end of stream"};
    uint64_t v = 0;
    int need = n;
    while (need > 0) {
      const size_t byte = pos_ >> 3;
      const int avail = 8 - (int)(pos_ & 7);
      const int take = avail < need ? avail : need;
      v = (v << take) | ((d_[byte] >> (avail - take)) & ((1u << take) - 1));
      pos_ += take;
      need -= take;
    }
    return v;
  }
  int64_t read_signed(int n) {
    if (n == 0) return 0;
    const uint64_t v = read(n);
    return (v >> (n - 1)) ? (int64_t)v - ((int64_t)1 << n) : (int64_t)v;
  }
  uint32_t unary() {  // count of 0 bits before the next 1
    uint32_t z = 0;
    for (;;) {
      if (pos_ >= n_ * 8) throw Error{"unexpected end of stream"};
      const size_t byte = pos_ >> 3;
      const int off = (int)(pos_ & 7);
      const uint8_t rest = (uint8_t)(d_[byte] << off);
      if (rest == 0) {
        z += 8 - off;
        pos_ += 8 - off;
        continue;
      }
      const int lead = __builtin_clz((uint32_t)rest) - 24;
      z += lead;
      pos_ += lead + 1;
      return z;
    }
  }
  void align() { pos_ = (pos_ + 7) & ~(size_t)7; }

 private:
  const uint8_t* d_;
  size_t n_;
  size_t pos_ = 0;
};

static uint8_t crc8(const uint8_t* p, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= p[i];
    for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1);
  }
  return c;
}
static uint16_t crc16(const uint8_t* p, size_t n) {
  static uint16_t table[256];
  static std::atomic<bool> ready{false};
  if (!ready.load()) {
    for (int i = 0; i < 256; ++i) {
      uint16_t c = (uint16_t)(i << 8);
      for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1);
      table[i] = c;
    }
    ready.store(true);
  }
  uint16_t c = 0;
  for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ table[(c >> 8) ^ p[i]]);
  return c;
}

struct Info {
  int32_t sample_rate = 0, channels = 0, bits = 0;
  int64_t frames = 0;
  size_t data_pos = 0, data_bytes = 0;  // WAV: PCM payload; FLAC: first audio frame
  int wav_format = 0;                   // 1 = integer PCM, 3 = IEEE float
  bool is_flac = false;
};

// ---- FLAC -------------------------------------------------------------------------------------------------------
static void flac_residual(Bits& br, int order, int blocksize, int64_t* out) {
  const int method = (int)br.read(2);
  if (method > 1) throw Error{"reserved residual coding method"};
  const int pbits = method == 0 ? 4 : 5;
  const int porder = (int)br.read(4);
  const int nparts = 1 << porder;
  if ((blocksize >> porder) << porder != blocksize && porder > 0) throw Error{"partition order does not divide the block"};
  int idx = order;
  for (int part = 0; part < nparts; ++part) {
    const int n = (blocksize >> porder) - (part == 0 ? order : 0);
    if (n < 0) throw Error{"partition shorter than the predictor order"};
    const int k = (int)br.read(pbits);
    if (k == (1 << pbits) - 1) {
      const int nb = (int)br.read(5);
      for (int i = 0; i < n; ++i) out[idx++] = nb ? br.read_signed(nb) : 0;
    } else {
      for (int i = 0; i < n; ++i) {
        const uint64_t q = br.unary();
        const uint64_t u = (q << k) | br.read(k);
        out[idx++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
      }
    }
  }
}

static void flac_subframe(Bits& br, int bps, int blocksize, int64_t* s) {
  if (br.read(1) != 0) throw Error{"subframe padding bit set"};
  const int type = (int)br.read(6);
  int wasted = 0;
  if (br.read(1)) {
    wasted = (int)br.unary() + 1;
    bps -= wasted;
  }
  if (bps <= 0 || bps > 33) throw Error{"bad subframe sample size"};
  static const int kFixed[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
  if (type == 0) {
    const int64_t v = br.read_signed(bps);
    for (int i = 0; i < blocksize; ++i) s[i] = v;
  } else if (type == 1) {
    for (int i = 0; i < blocksize; ++i) s[i] = br.read_signed(bps);
  } else if (type >= 8 && type <= 12) {
    const int order = type - 8;
    if (order > blocksize) throw Error{"predictor order exceeds the block"};
    for (int i = 0; i < order; ++i) s[i] = br.read_signed(bps);
    flac_residual(br, order, blocksize, s);
    for (int i = order; i < blocksize; ++i) {
      int64_t p = 0;
      for (int j = 0; j < order; ++j) p += (int64_t)kFixed[order][j] * s[i - 1 - j];
      s[i] += p;
    }
  } else if (type >= 32) {
    const int order = type - 31;
    if (order > blocksize) throw Error{"predictor order exceeds the block"};
    for (int i = 0; i < order; ++i) s[i] = br.read_signed(bps);
    const int prec = (int)br.read(4) + 1;
    if (prec == 16) throw Error{"invalid LPC precision"};
    const int shift = (int)br.read_signed(5);
    if (shift < 0) throw Error{"negative LPC shift"};
    int64_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = br.read_signed(prec);
    flac_residual(br, order, blocksize, s);
    for (int i = order; i < blocksize; ++i) {
      int64_t p = 0;
      for (int j = 0; j < order; ++j) p += coef[j] * s[i - 1 - j];
      s[i] += p >> shift;
    }
  } else {
    throw Error{"reserved subframe type"};
  }
  if (wasted)
    for (int i = 0; i < blocksize; ++i) s[i] *= ((int64_t)1 << wasted);
}

static Info flac_info(const uint8_t* d, size_t n) {
  Info in;
  in.is_flac = true;
  size_t pos = 4;
  bool have_streaminfo = false;
  for (;;) {
    if (pos + 4 > n) throw Error{"truncated FLAC metadata"};
    const uint8_t hdr = d[pos];
    const size_t size = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
    if (pos + 4 + size > n) throw Error{"truncated FLAC metadata block"};
    if ((hdr & 0x7F) == 0) {
      if (size < 34) throw Error{"short STREAMINFO"};
      Bits b(d + pos + 4, size);
      b.read(16); b.read(16); b.read(24); b.read(24);
      in.sample_rate = (int32_t)b.read(20);
      in.channels = (int32_t)b.read(3) + 1;
      in.bits = (int32_t)b.read(5) + 1;
      in.frames = (int64_t)b.read(36);
      have_streaminfo = true;
    }
    pos += 4 + size;
    if (hdr & 0x80) break;
  }
  if (!have_streaminfo || in.sample_rate <= 0) throw Error{"FLAC stream without STREAMINFO"};
  in.data_pos = pos;
  in.data_bytes = n - pos;
  return in;
}

// decodes into mono f32 (mean of the channels, scaled by 2^-(bits-1)); returns frames written
static int64_t flac_decode(const uint8_t* d, size_t n, const Info& in, float* out, int64_t cap) {
  Bits br(d, n);
  br.seek_byte(in.data_pos);
  std::vector<int64_t> ch[2];
  const float scale = 1.0f / (float)((int64_t)1 << (in.bits - 1));
  int64_t written = 0;
  const int64_t total = in.frames > 0 ? in.frames : INT64_MAX;
  if (in.channels > 2) throw Error{"more than two FLAC channels are not supported"};
  while (written < total && br.byte_pos() + 2 <= n) {
    const size_t frame_start = br.byte_pos();
    if (br.read(14) != 0x3FFE) throw Error{"lost FLAC frame sync"};
    br.read(1);
    br.read(1);
    const int bs_code = (int)br.read(4), sr_code = (int)br.read(4), ch_code = (int)br.read(4), bps_code = (int)br.read(3);
    br.read(1);
    const int first = (int)br.read(8);  // UTF-8 style frame / sample number
    int extra = 0;
    while (first & (0x80 >> extra)) ++extra;
    for (int i = 0; i < (extra > 0 ? extra - 1 : 0); ++i) br.read(8);
    int blocksize;
    if (bs_code == 0) throw Error{"reserved block size code"};
    else if (bs_code == 1) blocksize = 192;
    else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
    else if (bs_code == 6) blocksize = (int)br.read(8) + 1;
    else if (bs_code == 7) blocksize = (int)br.read(16) + 1;
    else blocksize = 256 << (bs_code - 8);
    if (sr_code == 12) br.read(8);
    else if (sr_code == 13 || sr_code == 14) br.read(16);
    else if (sr_code == 15) throw Error{"invalid sample rate code"};
    const size_t hdr_end = br.byte_pos();
    const uint8_t want8 = (uint8_t)br.read(8);
    if (crc8(d + frame_start, hdr_end - frame_start) != want8) throw Error{"FLAC frame header CRC mismatch"};
    static const int kBps[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    const int fbps = bps_code == 0 ? in.bits : kBps[bps_code];
    if (fbps <= 0) throw Error{"reserved sample size code"};
    const int nch = ch_code < 8 ? ch_code + 1 : 2;
    if (ch_code > 10) throw Error{"reserved channel assignment"};
    if (nch != in.channels) throw Error{"channel count changes mid-stream"};
    for (int c = 0; c < nch; ++c) ch[c].resize(blocksize);
    if (ch_code < 8) {
      for (int c = 0; c < nch; ++c) flac_subframe(br, fbps, blocksize, ch[c].data());
    } else {
      const int side = ch_code == 9 ? 0 : 1;  // which of the two coded channels is the side signal
      flac_subframe(br, fbps + (side == 0 ? 1 : 0), blocksize, ch[0].data());
      flac_subframe(br, fbps + (side == 1 ? 1 : 0), blocksize, ch[1].data());
      for (int i = 0; i < blocksize; ++i) {
        const int64_t a = ch[0][i], b = ch[1][i];
        if (ch_code == 8) {            // left / side
          ch[1][i] = a - b;
        } else if (ch_code == 9) {     // side / right
          ch[0][i] = a + b;
        } else {                       // mid / side
          const int64_t m = (a << 1) | (b & 1);
          ch[0][i] = (m + b) >> 1;
          ch[1][i] = (m - b) >> 1;
        }
      }
    }
    br.align();
    const size_t body_end = br.byte_pos();
    const uint16_t want16 = (uint16_t)br.read(16);
    if (crc16(d + frame_start, body_end - frame_start) != want16) throw Error{"FLAC frame CRC mismatch"};
    int64_t take = blocksize;
    if (written + take > total) take = total - written;
    if (written + take > cap) throw Error{"output buffer too small"};
    if (nch == 1)
      for (int64_t i = 0; i < take; ++i) out[written + i] = (float)ch[0][i] * scale;
    else
      for (int64_t i = 0; i < take; ++i) out[written + i] = 0.5f * ((float)ch[0][i] + (float)ch[1][i]) * scale;
    written += take;
  }
  if (in.frames > 0 && written != in.frames) throw Error{"FLAC stream ends before the announced length"};
  return written;
}

// ---- RIFF / WAVE --------------------------------------------------------------------------------------------------
static uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

static Info wav_info(const uint8_t* d, size_t n) {
  Info in;
  if (n < 12 || memcmp(d + 8, "WAVE", 4) != 0) throw Error{"not a RIFF/WAVE file"};
  size_t pos = 12;
  bool have_fmt = false;
  while (pos + 8 <= n) {
    const uint32_t size = le32(d + pos + 4);
    const uint8_t* body = d + pos + 8;
    if (memcmp(d + pos, "fmt ", 4) == 0) {
      if (size < 16 || pos + 8 + size > n) throw Error{"truncated fmt chunk"};
      in.wav_format = le16(body);
      in.channels = le16(body + 2);
      in.sample_rate = (int32_t)le32(body + 4);
      in.bits = le16(body + 14);
      if (in.wav_format == 0xFFFE && size >= 26) in.wav_format = le16(body + 24);  // WAVE_FORMAT_EXTENSIBLE
      have_fmt = true;
    } else if (memcmp(d + pos, "data", 4) == 0) {
      if (!have_fmt) throw Error{"data chunk before fmt chunk"};
      in.data_pos = pos + 8;
      in.data_bytes = (size_t)size <= n - in.data_pos ? size : n - in.data_pos;  // tolerate streamed files
      break;
    }
    pos += 8 + (size_t)size + (size & 1);
  }
  if (!have_fmt || in.data_pos == 0) throw Error{"WAVE file without fmt / data chunk"};
  if (in.channels < 1 || in.sample_rate <= 0) throw Error{"bad WAVE format fields"};
  const bool ok = (in.wav_format == 1 && (in.bits == 8 || in.bits == 16 || in.bits == 24 || in.bits == 32)) ||
                  (in.wav_format == 3 && (in.bits == 32 || in.bits == 64));
  if (!ok) throw Error{"unsupported WAVE sample format (integer PCM 8/16/24/32 or float 32/64 only)"};
  in.frames = (int64_t)(in.data_bytes / ((size_t)in.channels * (in.bits / 8)));
  return in;
}

static int64_t wav_decode(const uint8_t* d, const Info& in, float* out, int64_t cap) {
  if (in.frames > cap) throw Error{"output buffer too small"};
  const int bytes = in.bits / 8, nch = in.channels;
  const uint8_t* p = d + in.data_pos;
  const float inv_ch = 1.0f / (float)nch;
  for (int64_t i = 0; i < in.frames; ++i) {
    float acc = 0.f;
    for (int c = 0; c < nch; ++c, p += bytes) {
      float v;
      if (in.wav_format == 3) {
        if (bytes == 4) { float f; memcpy(&f, p, 4); v = f; }
        else { double f; memcpy(&f, p, 8); v = (float)f; }
      } else if (bytes == 1) {
        v = ((float)p[0] - 128.f) / 128.f;
      } else if (bytes == 2) {
        v = (float)(int16_t)le16(p) / 32768.f;
      } else if (bytes == 3) {
        const int32_t s = (int32_t)((p[0] << 8) | (p[1] << 16) | ((uint32_t)p[2] << 24)) >> 8;
        v = (float)s / 8388608.f;
      } else {
        v = (float)(int32_t)le32(p) / 2147483648.f;
      }
      acc += v;
    }
    out[i] = nch == 1 ? acc : acc * inv_ch;
  }
  return in.frames;
}

static Info probe(const uint8_t* d, size_t n) {
  if (n >= 4 && memcmp(d, "fLaC", 4) == 0) return flac_info(d, n);
  if (n >= 4 && memcmp(d, "RIFF", 4) == 0) return wav_info(d, n);
  throw Error{"unknown audio container (FLAC and RIFF/WAVE are supported)"};
}

static int64_t decode(const uint8_t* d, size_t n, float* out, int64_t cap, int32_t* sr) {
  const Info in = probe(d, n);
  if (sr) *sr = in.sample_rate;
  return in.is_flac ? flac_decode(d, n, in, out, cap) : wav_decode(d, in, out, cap);
}

static bool read_file(const char* path, std::vector<uint8_t>* buf) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf->resize(sz > 0 ? (size_t)sz : 0);
  const size_t got = sz > 0 ? fread(buf->data(), 1, (size_t)sz, f) : 0;
  fclose(f);
  return got == buf->size();
}

}  // namespace audio
}  // namespace caiman

using namespace caiman::audio;

extern "C" int caiman_audio_info(const uint8_t* data, int64_t size, int32_t* sample_rate, int32_t* channels,
                                 int64_t* frames) {
  CAIMAN_CHECK(data && size > 0, "audio_info: empty input");
  try {
    const Info in = probe(data, (size_t)size);
    if (sample_rate) *sample_rate = in.sample_rate;
    if (channels) *channels = in.channels;
    if (frames) *frames = in.frames;
  } catch (const Error& e) {
    caiman::set_error("audio_info: %s", e.msg.c_str());
    return CAIMAN_ERR_INVALID;
  }
  return CAIMAN_OK;
}

extern "C" int caiman_audio_decode(const uint8_t* data, int64_t size, float* out, int64_t capacity, int64_t* frames,
                                   int32_t* sample_rate) {
  CAIMAN_CHECK(data && size > 0 && out && frames && capacity >= 0, "audio_decode: null / empty argument");
  try {
    *frames = decode(data, (size_t)size, out, capacity, sample_rate);
  } catch (const Error& e) {
    caiman::set_error("audio_decode: %s", e.msg.c_str());
    return CAIMAN_ERR_INVALID;
  }
  return CAIMAN_OK;
}

extern "C" int caiman_audio_decode_files(const char* const* paths, int32_t n, float* out, int64_t max_frames,
                                         int32_t* lengths, int32_t* sample_rates, int32_t n_threads) {
  CAIMAN_CHECK(paths && out && lengths && n >= 0 && max_frames > 0, "audio_decode_files: null argument");
  std::atomic<int32_t> next{0};
  std::atomic<int32_t> failed{-1};
  std::string err;
  std::mutex mu;
  auto work = [&]() {
    std::vector<uint8_t> buf;
    for (;;) {
      const int32_t i = next.fetch_add(1);
      if (i >= n) return;
      float* row = out + (int64_t)i * max_frames;
      try {
        if (!paths[i] || !read_file(paths[i], &buf)) throw Error{std::string("cannot read ") + (paths[i] ? paths[i] : "(null)")};
        int32_t sr = 0;
        const int64_t got = decode(buf.data(), buf.size(), row, max_frames, &sr);
        memset(row + got, 0, (size_t)(max_frames - got) * sizeof(float));
        lengths[i] = (int32_t)got;
        if (sample_rates) sample_rates[i] = sr;
      } catch (const Error& e) {
        std::lock_guard<std::mutex> g(mu);
        if (failed.load() < 0) {
          failed.store(i);
          err = std::string(paths[i] ? paths[i] : "(null)") + ": " + e.msg;
        }
      }
    }
  };
  const int nt = n_threads < 1 ? 1 : (n_threads > n ? (n > 0 ? n : 1) : n_threads);
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  if (failed.load() >= 0) {
    caiman::set_error("audio_decode_files: %s", err.c_str());
    return CAIMAN_ERR_INVALID;
  }
  return CAIMAN_OK;
}
