/*
 * caiman_beam.h — C-ABI of the beam-search decoder for thousands of concurrent streams.
 *
 * Two halves, both behind plain C entry points:
 *
 *  (1) caiman_beam_topk — DEVICE: what the reference computes per expansion round with
 *      `log_softmax(logits / T)`, the EOS strategy, `topk(beam_width)` and the blank column
 *      (training/caiman_asr_train/rnnt/decoder.py:139-172, beam.py:533-546), as one pass over the
 *      logits of all pending hypotheses; fixed-shape outputs so one D2H copy brings them back.
 *
 *  (2) caiman_beam_* search object — HOST: the per-utterance search the reference runs as Python
 *      generators (beam.py:285-516: best-first expansion of the open set, hypothesis merging by text
 *      hash hypothesis.py:114-122, width / length-normalised pruning beam.py:661-683, common-prefix
 *      finals and partials serialise_responses.py:28-205, keyword boosting keywords/trie.py), restated
 *      as one object that owns every stream's beam.  Frames are pushed as they arrive, so the same
 *      object serves offline batches and real-time streams.
 *
 * Prediction-network states never leave the device: the search refers to them by SLOT index into a
 * pool the caller keeps in HBM ([layers, slots, hidden] for h and c).  Per round the caller
 *     n = caiman_beam_requests(...)      -> (stream, last token, state slot in, state slot out) per pending hypothesis
 *     gathers states by slot, runs prediction + joint + caiman_beam_topk, scatters new states to `slot out`
 *     caiman_beam_feed(...)              <- top-k scores / tokens / blank log-prob per request
 * until no stream has a request left for the frames pushed so far.
 *
 * Every function returns 0 on success (or a count where stated) and a non-zero code with
 * caiman_last_error() set otherwise.  Host functions are not thread-safe per handle.
 */
#ifndef CAIMAN_BEAM_H_
#define CAIMAN_BEAM_H_

#include <stdint.h>

#include "caiman_rnnt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* EOS handling of the reference's decoders (rnnt/eos_strategy.py:7-27, decoder.py:139-157). */
typedef enum {
  CAIMAN_EOS_NONE = 0,
  CAIMAN_EOS_IGNORE = 1,  /* log p(eos) = -inf */
  CAIMAN_EOS_BLANK = 2,   /* p(blank) += p(eos); log p(eos) = -inf */
  CAIMAN_EOS_PREDICT = 3  /* log p(eos) *= alpha; -inf unless > log(beta) (beta > 0) */
} caiman_eos_mode_t;

/* logits [n, vocab] (dtype tag, `row_stride` elements between rows) ->
 *   top_scores [n, k] f32 (descending), top_tokens [n, k] i32, blank_logp [n] f32
 * of log_softmax(logits / temperature) after the EOS correction.  k <= 8, k <= vocab.
 * Ties resolve to the lower token id. */
int caiman_beam_topk(const void* logits, int64_t n, int64_t vocab, int64_t row_stride, int dtype,
                     float temperature, int32_t blank_idx, int eos_mode, int32_t eos_idx,
                     float eos_alpha, float eos_beta, int32_t k, float* top_scores,
                     int32_t* top_tokens, float* blank_logp, caiman_stream_t stream);

/* The memory-bound steps of one expansion round (prediction step + joint for n pending hypotheses; beam.py:564-612,
 * rnnt/model.py:344-439).  With one timestep per call each LSTM layer is one library GEMM
 * gates = [x | h_prev] · [W_ih | W_hh]^T + (b_ih + b_hh); these entry points do the gathers, the cell and the
 * scatter around it.  State pools are [layers, 1 + slots, hidden]: row 0 is the zero start state, search slot s
 * is row s + 1; h pools have the compute dtype, c pools are f32.
 *   gather_inputs: X[i] = [ embed[y_last_i] (zeros when y_last_i < 0) | h_pool_l0[slot_in_i + 1] ]      X [n, ldx]
 *   lstm_cell    : gates [n, 4·hidden] (order i,f,g,o: training/lib/csrc/lstm.cu:99-123) and c_pool_l[slot_in_i+1]
 *                  -> c, h stored at row slot_out_i + 1 of this layer's pools; X_next[i] = [ h | h_pool_next[slot_in_i+1] ]
 *                  (h_pool_next = the next layer's h pool, or NULL for the last layer: X_next[i] = h)
 *   joint_act    : A[i] = relu(f_rows[row_i] + g[i])                          (rnnt/model.py:409-439, dropout off) */
int caiman_beam_gather_inputs(const void* embed, int64_t embed_dim, const void* h_pool_l0, int64_t hidden,
                              const int32_t* y_last, const int32_t* slot_in, int64_t n, void* X,
                              int64_t ldx, int dtype, caiman_stream_t stream);
int caiman_beam_lstm_cell(const void* gates, int64_t hidden, float* c_pool_l, void* h_pool_l,
                          const void* h_pool_next, const int32_t* slot_in, const int32_t* slot_out,
                          int64_t n, void* X_next, int64_t ldx, int dtype, caiman_stream_t stream);
int caiman_beam_joint_act(const void* f_rows, const int64_t* row, const void* g, int64_t n,
                          int64_t joint_dim, void* A, int dtype, caiman_stream_t stream);

/* One timestep of one LSTM layer for n rows (live streams of the streaming encoder, pending hypotheses of a beam round)
 * as ONE launch: the 4-gate GEMM [x_t | h_prev] · [W_ih | W_hh]^T on MFMA with the cell update, the scatter of (c, h)
 * into the state pools and the next layer's input row as its epilogue -- replaces the library GEMM + caiman_beam_lstm_cell
 * pair above, i.e. the reference's cuBLAS GEMM + pointwise kernel per step (training/lib/csrc/lstm.cu:259-271, :85-135)
 * as the decoders drive it (training/caiman_asr_train/rnnt/beam.py:564-612, batched_greedy.py:59-166).
 *   X      [n, ldx_in]  rows [x_t | h_prev], K = used width (K % 128 == 0: zero-pad x and the matching columns of W)
 *   W      [4·hidden, K] rows ordered [unit][gate] (row 4u + q = gate q (i,f,g,o) of unit u), bias [4·hidden] likewise
 *   pools, slots, X_next, ldx: as caiman_beam_lstm_cell.  bf16 / f16. */
int caiman_lstm_step_gemm(const void* X, int64_t ldx_in, const void* W, const void* bias, int64_t n, int64_t hidden,
                          int64_t K, float* c_pool_l, void* h_pool_l, const void* h_pool_next,
                          const int32_t* slot_in, const int32_t* slot_out, void* X_next, int64_t ldx, int dtype,
                          caiman_stream_t stream);

/* Search parameters: constructor arguments of RNNTBeamDecoder (beam.py:115-137). Thresholds < 0 mean
 * "off" (infinite), as in the reference (:153-154,190-200). */
typedef struct {
  int32_t blank_idx;
  int32_t beam_width;
  int32_t max_symbols_per_step;   /* <= 0: unlimited */
  int32_t max_symbol_per_sample;  /* < 0: unlimited */
  double beam_prune_score_thresh; /* nats per token */
  double beam_prune_topk_thresh;  /* nats */
  double eos_vad_threshold;       /* seconds of silence that end a stream */
  double final_emission_thresh;   /* seconds without a final before the beam is forced to agree */
  double frame_width;             /* seconds per encoder frame */
  int32_t eos_terminal_idx;       /* token that ends a stream when predicted, or -1 */
  int32_t return_partials;
  /* Serving safeguard, not in the reference (0 = off, the reference's behaviour): a frame on which a stream has
   * expanded this many hypotheses is settled with the hypotheses closed so far, as if the open set had run dry
   * (beam.py:410-413).  Bounds the work one stream can demand per frame. */
  int32_t max_expansions_per_frame;
} caiman_beam_config_t;

typedef struct caiman_beam caiman_beam_t;

/* pieces: UTF-8 text of every token id (sentencepiece id_to_piece); keywords: phrases (spaces already
 * replaced by U+2581) with their per-symbol weights, or n_keywords = 0.  Returns NULL on error. */
caiman_beam_t* caiman_beam_create(const caiman_beam_config_t* cfg, int32_t n_streams,
                                  const char* const* pieces, int32_t n_pieces,
                                  const char* const* keywords, const double* keyword_weights,
                                  int32_t n_keywords);
void caiman_beam_destroy(caiman_beam_t* h);

/* Start a new utterance on `stream` (drops its beam, frees its state slots). */
int caiman_beam_reset_stream(caiman_beam_t* h, int32_t stream);
/* The next encoder frame of each listed stream is available.  An idle stream starts expanding it at once; a
 * stream still busy with an earlier frame queues it and moves on by itself when that frame closes, so a few
 * slow streams need not hold up a real-time tick (the caller keeps the encoder frames it has pushed until
 * caiman_beam_backlog says they are done).  Streams that have ended (terminal token, silence, symbol budget)
 * ignore the call. */
int caiman_beam_push_frame(caiman_beam_t* h, const int32_t* streams, int32_t n);
/* Pending expansions, one per stream with an open frame.  y_last = -1 and state_in = -1 mark the
 * start-of-sequence step (zero embedding, zero state).  `frame` is the index (count of frames pushed before it)
 * of the encoder frame the hypothesis is to be expanded on.  state_out is a fresh slot the caller must fill
 * with the new prediction state.  Returns the count (<= cap), or -1 on error. */
int64_t caiman_beam_requests(caiman_beam_t* h, int32_t* stream, int32_t* frame, int32_t* y_last,
                             int32_t* state_in, int32_t* state_out, int64_t cap);
/* Answers for exactly the requests returned by the last caiman_beam_requests call, in order:
 * top_scores / top_tokens [n, k] as written by caiman_beam_topk, blank_logp [n] (host pointers). */
int caiman_beam_feed(caiman_beam_t* h, int64_t n, int32_t k, const float* top_scores,
                     const int32_t* top_tokens, const float* blank_logp);
/* No more audio on `stream`: ships what the best hypothesis still holds (one frame after the last). */
int caiman_beam_close_stream(caiman_beam_t* h, int32_t stream);
/* 1 if the stream has ended by itself or was closed. */
int caiman_beam_stream_done(const caiman_beam_t* h, int32_t stream);
/* Frames pushed but not yet finished on `stream`; stream = -1: the maximum over all streams. */
int64_t caiman_beam_backlog(const caiman_beam_t* h, int32_t stream);
/* Frames settled early by max_expansions_per_frame since the object was created. */
int64_t caiman_beam_capped_frames(const caiman_beam_t* h);
/* Number of state slots the device pool must hold (grows; check after caiman_beam_requests). */
int64_t caiman_beam_state_slots(const caiman_beam_t* h);

/* Responses accumulated since the last clear, as two flat arrays that stay valid until the next call
 * on the handle.  ints: a sequence of records
 *     stream, frame_key, kind (0 final | 1 partials | 2 frame closed with neither), start_frame, duration_frames,
 *     n_alternatives, then per alternative: n_tokens, token ids..., frame indices...
 * floats: the confidences of all alternatives in the same order. */
int caiman_beam_responses(caiman_beam_t* h, const int32_t** ints, int64_t* n_ints,
                          const float** floats, int64_t* n_floats);
void caiman_beam_clear_responses(caiman_beam_t* h);

#ifdef __cplusplus
}
#endif
#endif /* CAIMAN_BEAM_H_ */
