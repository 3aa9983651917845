/*
 * caiman_data.h — C-ABI of the host-side data feed: audio files -> mono f32 samples.
 *
 * Replaces the DALI reader + audio decoder the reference drives
 * (training/caiman_asr_train/data/dali/pipeline.py:253-259 `fn.readers.file`, :400-414 `fn.decoders.audio(...,
 * downmix=True, dtype=FLOAT)`; DALI is third party and not vendored).  Decoders follow the format specifications:
 * FLAC per RFC 9639 (all subframe types, stereo decorrelation, frame CRC-8 / CRC-16 verified), RIFF/WAVE integer PCM
 * 8/16/24/32 bit and IEEE float 32/64.  Output: channels averaged to mono, integer samples scaled by 2^-(bits-1).
 * No resampling: the sample rate is reported and the caller decides (the shipped configs and LibriSpeech are 16 kHz).
 *
 * Host functions; thread-safe; 0 on success, else caiman_last_error() holds the reason.
 */
#ifndef CAIMAN_DATA_H_
#define CAIMAN_DATA_H_

#include <stdint.h>

#include "caiman_rnnt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Container probe: sample rate, channel count, frames per channel (0 if a FLAC stream does not announce it). */
int caiman_audio_info(const uint8_t* data, int64_t size, int32_t* sample_rate, int32_t* channels,
                      int64_t* frames);
/* Decode one in-memory file into out[0 .. *frames); fails if capacity is too small. */
int caiman_audio_decode(const uint8_t* data, int64_t size, float* out, int64_t capacity,
                        int64_t* frames, int32_t* sample_rate);
/* Read and decode n files with n_threads host threads into out [n, max_frames] (zero padded; typically pinned
 * memory that is then copied to the device in one transfer).  lengths [n] receives the frame counts,
 * sample_rates [n] (may be NULL) the rates.  An utterance longer than max_frames is an error. */
int caiman_audio_decode_files(const char* const* paths, int32_t n, float* out, int64_t max_frames,
                              int32_t* lengths, int32_t* sample_rates, int32_t n_threads);

/* Levenshtein distance between two sequences of ids (insert / delete / substitute, unit costs): the kernel of the
 * reference's WER, which it takes from the Rust extension `levenshtein_rs`
 * (training/caiman_asr_train/evaluate/metrics.py:20,123).  Returns the distance, or -1 on error. */
int64_t caiman_levenshtein(const int32_t* a, int64_t n, const int32_t* b, int64_t m);

#ifdef __cplusplus
}
#endif
#endif /* CAIMAN_DATA_H_ */
