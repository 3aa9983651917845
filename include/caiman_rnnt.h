/*
 * caiman_rnnt.h — C-ABI of the MI355X (gfx950) RNN-T hot path.
 *
 * This is the drop-in boundary for the reference's native operator layer
 * (the `.cu` files of `training/lib/csrc`, bound to Python as `rnnt_ext.cuda.{lstm,logsumexp,
 * transducer_loss}`), plus the third-party device ops the reference calls on the
 * same path (apex TransducerJoint / FusedLAMB, DALI log-mel) re-stated as plain
 * C entry points.  Every function:
 *   - takes raw DEVICE pointers, explicit extents and a dtype tag (no torch types),
 *   - launches on the caller's HIP stream (`stream` is a hipStream_t, may be NULL),
 *   - never synchronises, never allocates device memory (caller owns every buffer,
 *     like the reference: training/lib/src/rnnt_ext/custom_lstm/lstm.py:76-80,124),
 *   - returns 0 on success or a non-zero code; `caiman_last_error()` then holds the
 *     message (the reference raises c10::Error via TORCH_CHECK,
 *     training/lib/csrc/transducer_loss.cu:429-448).
 *
 * Reference citations are relative to /root/reference.
 */
#ifndef CAIMAN_RNNT_H_
#define CAIMAN_RNNT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Element types accepted at the boundary — the set MYRTLE_DISPATCH_FLOATING_TYPES
 * dispatches over (training/lib/csrc/myrtle/utility.hpp:40-47). The accumulate
 * type is f64 for f64 and f32 for everything else (at::acc_type<T, true>). */
typedef enum {
  CAIMAN_F64 = 0,
  CAIMAN_F32 = 1,
  CAIMAN_F16 = 2,
  CAIMAN_BF16 = 3
} caiman_dtype_t;

enum {
  CAIMAN_OK = 0,
  CAIMAN_ERR_INVALID = 1, /* argument check failed (TORCH_CHECK equivalent) */
  CAIMAN_ERR_LAUNCH = 2,  /* hipGetLastError() after a launch */
  CAIMAN_ERR_UNSUPPORTED = 3
};

typedef void* caiman_stream_t; /* hipStream_t */

/* Library identity / diagnostics. */
int caiman_abi_version(void);
const char* caiman_last_error(void);
/* 1 when the library was built with device code for gfx950. */
int caiman_built_for_gfx950(void);

/* ------------------------------------------------------------------------- *
 * logsumexp — replaces rnnt_ext.cuda.logsumexp.logsumexp
 *   training/lib/csrc/logsumexp.cu:189-243 (host), :65-105 (kernel)
 * in  : [rows, n], unit stride in the last dim, `row_stride` (elements) >= n
 * out : [rows] of `out_dtype` (= accumulate type when the reference is called
 *        with promote=True, else in_dtype)
 * max_threads is accepted for signature parity; the wave64 kernel picks its own
 * workgroup size.  Rows whose max is not finite return that max (NaN/±inf),
 * training/lib/csrc/logsumexp.cu:91-96.
 * ------------------------------------------------------------------------- */
int caiman_logsumexp(const void* in, int64_t rows, int64_t n, int64_t row_stride,
                     int in_dtype, void* out, int out_dtype, uint32_t max_threads,
                     caiman_stream_t stream);

/* ------------------------------------------------------------------------- *
 * transducer loss forward — replaces rnnt_ext.cuda.transducer_loss.forward
 *   training/lib/csrc/transducer_loss.cu:396-501 (host), :79-264 (kernel)
 * x      : logits, packed [batch_offset[B-1], V] or padded [B, max_f_len, max_g_len, V]
 * denom  : logsumexp of x over V, accumulate type, same leading shape as x
 * label  : [B, max_g_len-1] int32;  f_len, y_len : [B] int32
 * batch_offset : [B] int64 inclusive cumsum of f_len*(y_len+1) (packed only)
 * alpha, beta : [B, max_f_len, max_g_len] accumulate type (out); loss : [B] (out)
 * eos_idx = -1 / star_idx = -2 disable the modifiers
 *   (training/lib/src/rnnt_ext/transducer/loss.py:184-194).
 * ------------------------------------------------------------------------- */
int caiman_transducer_loss_forward(
    const void* x, const void* denom, const int32_t* label, const int32_t* f_len,
    const int32_t* y_len, const int64_t* batch_offset, int64_t batch,
    int64_t max_f_len, int64_t max_g_len, int64_t dict_size, double dp_lam,
    int64_t blank_idx, double eos_lam, int64_t eos_idx, double star_lam,
    int64_t star_idx, int packed, int dtype, void* alpha, void* beta, void* loss,
    caiman_stream_t stream);

/* transducer loss backward (log-softmax backward fused) — replaces
 * rnnt_ext.cuda.transducer_loss.backward
 *   training/lib/csrc/transducer_loss.cu:503-590 (host), :274-394 (kernel)
 * x_grad : same shape/dtype as x (out). Padded layout: don't-care cells are zeroed.
 * total_rows : number of [*, V] rows of x (packed: batch_offset[B-1]; padded:
 *              B*max_f_len*max_g_len) — lets the launch cover exactly the rows. */
int caiman_transducer_loss_backward(
    const void* x, const void* denom, const void* loss_grad, const void* alpha,
    const void* beta, const int32_t* f_len, const int32_t* y_len,
    const int32_t* label, const int64_t* batch_offset, int64_t batch,
    int64_t max_f_len, int64_t max_g_len, int64_t dict_size, int64_t total_rows,
    double dp_lam, int64_t blank_idx, double eos_lam, int64_t eos_idx,
    double star_lam, int64_t star_idx, int packed, int dtype, void* x_grad,
    caiman_stream_t stream);

/* ------------------------------------------------------------------------- *
 * LSTM recurrent passes — replace rnnt_ext.cuda.lstm.lstm_fused_{fwd,bwd}_{soft,hard}
 *   training/lib/csrc/lstm.cu:214-272 / :353-372 (fwd), :274-346 / :377-405 (bwd)
 * R      : [4H, H] recurrent weights, gate row blocks i,f,g,o (lstm.cu:99-102)
 * gates  : [T, B, 4H] in: x·Wᵀ + biases for every step; out: ACTIVATED gates
 * c, y   : [T+1, B, H]; row 0 holds the initial state, rows 1..T are written
 * hard   : 0 = sigmoid/tanh, 1 = hard-sigmoid/hard-tanh (lstm.cu:41-76)
 * All tensors share `dtype`. Math is done in the accumulate type.
 * ------------------------------------------------------------------------- */
int caiman_lstm_fused_fwd(const void* R, void* gates, void* c, void* y, int64_t T,
                          int64_t B, int64_t H, int dtype, int hard,
                          caiman_stream_t stream);

/* delta : upstream gradient w.r.t. y[1..T], [T, B, H] addressed with explicit element
 *         strides (unit stride on H). It is read-only: the reference first copies it to a
 *         contiguous `partials` and accumulates dG[t+1]·R into that copy
 *         (lstm.cu:325-333,394-396); here the sum is formed in registers instead.
 * dG    : [T, B, 4H] out (gradient w.r.t. the PRE-activation gates)
 * dC    : [B, H] scratch in the ACCUMULATE type, zeroed by the callee (lstm.cu:298)
 * Rt    : [H, 4H] scratch (may be NULL: scalar path); the callee fills it with Rᵀ. */
int caiman_lstm_fused_bwd(const void* R, const void* gates, const void* c, const void* delta,
                          int64_t delta_stride_t, int64_t delta_stride_b, void* dG, void* dC,
                          void* Rt, int64_t T, int64_t B, int64_t H, int dtype, int hard,
                          caiman_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CAIMAN_RNNT_H_ */
