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

/* Library identity / diagnostics.  ABI version 4: slot structs of the wave calls carry `hidden` (2) and the backward
 * slot `dbias` (3); the resident-kernel controls were added with 3; 4: caiman_lamb_step drops a step after a resident
 * hand-off timeout (work[5], work[6]) and caiman_lstm_resident_poison carries that decision to the other ranks. */
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

/* The same backward pass, also summing the gradient it writes column by column: the bias gradient of the projection
 * that produced x.  The reference gets it from autograd as `grad_output.sum(0)` for `joint_fc.bias`
 * (training/caiman_asr_train/rnnt/model.py:201,436-439: an nn.Linear) -- one more pass over the whole gradient.
 * colsum_partial : workspace of roundup8(nblk * dict_size) + 8 * total_rows floats, 32-byte aligned, nblk =
 *   ceil(total_rows / rows_per_block).  On return its first [nblk, dict_size] floats hold one partial row per workgroup,
 *   in a fixed order (deterministic): the caller adds the rows up.  (The rest held one 32-byte descriptor per row:
 *   everything that is constant along a row.)  rows_per_block: multiple of 4 (64 is a good value).
 *   Needs x, x_grad 16-byte aligned with dict_size * sizeof(dtype) a multiple of 16, and dict_size <= 16 * 256 * 16 /
 *   sizeof(dtype) (the sums live in registers); CAIMAN_ERR_INVALID otherwise: use the plain call and reduce.  Sums are
 *   over the values as rounded to `dtype`. */
int caiman_transducer_loss_backward_colsum(
    const void* x, const void* denom, const void* loss_grad, const void* alpha,
    const void* beta, const int32_t* f_len, const int32_t* y_len,
    const int32_t* label, const int64_t* batch_offset, int64_t batch,
    int64_t max_f_len, int64_t max_g_len, int64_t dict_size, int64_t total_rows,
    double dp_lam, int64_t blank_idx, double eos_lam, int64_t eos_idx,
    double star_lam, int64_t star_idx, int packed, int dtype, void* x_grad,
    float* colsum_partial, int64_t rows_per_block, caiman_stream_t stream);

/* ------------------------------------------------------------------------- *
 * LSTM recurrent passes — replace rnnt_ext.cuda.lstm.lstm_fused_{fwd,bwd}_{soft,hard}
 *   training/lib/csrc/lstm.cu:214-272 / :353-372 (fwd), :274-346 / :377-405 (bwd)
 * R      : [4H, H] recurrent weights, gate row blocks i,f,g,o (lstm.cu:99-102)
 * gates  : [T, B, 4H] in: x·Wᵀ + biases for every step; out: ACTIVATED gates
 * c, y   : [T+1, B, H]; row 0 holds the initial state, rows 1..T are written
 * hard   : 0 = sigmoid/tanh, 1 = hard-sigmoid/hard-tanh (lstm.cu:41-76)
 * All tensors share `dtype`. Math is done in the accumulate type.
 * ------------------------------------------------------------------------- */
int caiman_lstm_fused_fwd(const void* R, void* gates, void* c, void* y, void* work,
                          int64_t T, int64_t B, int64_t H, int dtype, int hard,
                          caiman_stream_t stream);

/* Scratch the MFMA path needs, in ELEMENTS of `dtype` (forward: backward = 0):
 *   4H*H (weights re-laid fragment-major once per call) + a 2-deep ring of the per-step
 *   operand (h: pad32(B)*H, dG: pad32(B)*4H).  `work` may be NULL: the scalar kernels are
 *   used then (correct for every dtype / size, but not fast).  f32 / f64 never need it. */
int64_t caiman_lstm_workspace_elems(int64_t B, int64_t H, int backward);

/* delta : upstream gradient w.r.t. y[1..T], [T, B, H] addressed with explicit element
 *         strides (unit stride on H). It is read-only: the reference first copies it to a
 *         contiguous `partials` and accumulates dG[t+1]·R into that copy
 *         (lstm.cu:325-333,394-396); here the sum is formed in registers instead.
 * dG    : [T, B, 4H] out (gradient w.r.t. the PRE-activation gates)
 * dC    : [B, H] scratch in the ACCUMULATE type, zeroed by the callee (lstm.cu:298)
 * work  : see caiman_lstm_workspace_elems(B, H, 1); may be NULL. */
int caiman_lstm_fused_bwd(const void* R, const void* gates, const void* c, const void* delta,
                          int64_t delta_stride_t, int64_t delta_stride_b, void* dG, void* dC,
                          void* work, int64_t T, int64_t B, int64_t H, int dtype, int hard,
                          caiman_stream_t stream);

/* ---- multi-layer ("wave") interface of the same kernels ------------------------------------ *
 * One launch advances several independent recurrences (slots): a stack of L layers is run as a
 * pipeline in which layer l works a chunk of timesteps behind layer l-1, so a sequence costs
 * T + (L-1)*chunk dependent kernel boundaries instead of L*T.  f16 / bf16, H % 32 == 0.
 *
 * caiman_lstm_prepare : once per layer and pass. Re-lays R [4H,H] fragment-major into
 *   weights_tiled (4H*H elements), clears the operand ring (2*pad32(B)*H elements forward,
 *   2*pad32(B)*4H backward) and, forward: tiles the initial state h0 [B,H] into ring half 0;
 *   backward: zeroes dC [B,H] f32.
 * forward slot : pointers at the slot's FIRST timestep t0 of this call: gates row t0, c / y row t0
 *   (inputs; outputs go to row t0+1 ...), parity = t0 & 1, nsteps = timesteps to run (<= n_launches).
 * backward slot: pointers at the slot's LAST timestep t_hi of this call (rows walk downwards):
 *   gates / c / delta / dG row t_hi, parity = t_hi & 1, has_next = (t_hi is not the final timestep).
 * Inter-layer dropout is fused: a forward slot with y_masked != NULL also writes
 *   y_masked[row] = y[row] * keep(seed, drop_counter + element) / (1 - drop_p)   (rows parallel to y's OUTPUT rows)
 * which is what the layer above multiplies with W_ih; a backward slot with drop_p > 0 multiplies the incoming
 * delta (gradient w.r.t. the masked copy) by the same factor.  keep() is a counter hash (no mask tensors);
 * drop_counter is the counter of the first element of the slot's first output row (forward) / of row t_hi
 * (backward); caiman_lstm_dropout_mask materialises the factors for tests. */
typedef struct {
  const void* weights_tiled; void* gates; void* c; void* y; void* ring;
  int32_t parity; int32_t nsteps;
  void* y_masked; uint64_t drop_counter; float drop_p;
  int32_t hidden; /* 0: the call's H.  Otherwise this slot's own hidden size (a multiple of 32, <= H): slots of
                     different width can share a launch (prediction network next to the encoder) */
} caiman_lstm_fwd_slot_t;
typedef struct {
  const void* weights_tiled; const void* gates; const void* c; const void* delta;
  int64_t delta_stride_t; int64_t delta_stride_b; void* dG; void* ring; void* dC;
  int32_t parity; int32_t nsteps; int32_t has_next; float drop_p;
  uint64_t drop_counter;
  int32_t hidden; int32_t reserved; /* hidden: as in the forward slot */
  float* dbias; /* NULL, or fp32 [4H] in the call's gate layout: the call ADDS the sum of the dG rows it produces for this
                   slot over its timesteps and batch rows -- the bias gradient of the layer (reference:
                   training/lib/src/rnnt_ext/custom_lstm/lstm.py:57, `dB = dG.sum([0, 1])`, a separate pass over dG).
                   The resident kernel keeps the sums in registers; the per-timestep path adds them with one extra
                   reduction launch per call.  Interleaved gate layout only. */
} caiman_lstm_bwd_slot_t;
/* gate_layout: 0 = the reference's gates / dG layout [B, 4, H] (gate-major, lstm.cu:99-102);
 *              1 = interleaved [B, H, 4] (the 4 gates of a hidden unit adjacent): an internal layout of the
 *                  stack pipeline (the caller permutes the rows of W_ih / biases accordingly) that turns the
 *                  epilogue's 2-byte strided accesses into 8-byte vectors.
 * caiman_lstm_prepare only: + 2 = `ring` (and `dC`) are already zero -- the caller cleared the rings of all layers with one
 *                  memset -- so the call issues no memset of its own. */
int caiman_lstm_prepare(const void* R, const void* h0, void* weights_tiled, void* ring, void* dC,
                        int64_t B, int64_t H, int dtype, int backward, int gate_layout,
                        caiman_stream_t stream);
int caiman_lstm_wave_fwd(const caiman_lstm_fwd_slot_t* slots, int n_slots, int n_launches,
                         int64_t B, int64_t H, int dtype, int hard, int gate_layout,
                         uint64_t seed, caiman_stream_t stream);
int caiman_lstm_wave_bwd(const caiman_lstm_bwd_slot_t* slots, int n_slots, int n_launches,
                         int64_t B, int64_t H, int dtype, int hard, int gate_layout,
                         uint64_t seed, caiman_stream_t stream);
int caiman_lstm_dropout_mask(void* out, int64_t n, uint64_t seed, uint64_t base, float p, int dtype,
                             caiman_stream_t stream);
/* Weight-resident variant of the wave calls (same slots, same results up to fp32 summation order): with mode 1
 * (the default) a wave call whose shapes allow it (interleaved gates, one hidden size, n_slots * H/32 workgroups not more
 * than the device has CUs; B <= 32, or up to 128 in tiles of 32 batch rows for H = 512 / 1024) runs ALL its timesteps in one launch, each workgroup keeping its rows of R in registers
 * and the workgroups of a slot meeting at a device counter once per timestep; other calls keep the per-timestep
 * launches.  Replaces the same time loop (training/lib/csrc/lstm.cu:214-346).  caiman_lstm_resident_mode returns the
 * previous mode; caiman_lstm_resident_failures counts workgroups that timed out waiting (0 in a healthy process; a
 * non-zero value invalidates the results of that launch; from then on the process keeps to the per-timestep launches) and does not synchronise the device;
 * caiman_lstm_resident_launches counts the wave calls served this way. */
int caiman_lstm_resident_mode(int mode);
/* Backward wave calls with H = 512 / 1024: 1 (default) = 2-D split resident kernel (a workgroup keeps 128 columns of R
 * for one quarter of K and gathers a quarter of the dG row; the four K-quarter partial sums meet in a second hand-off
 * per timestep), 0 = the whole-row kernel.  Returns the previous setting. */
int caiman_lstm_resident_bwd_split(int on);
/* Batch-tile kernels (32 < B <= 128, H = 512 / 1024), forward and backward: 1 (default) = the operands of tile-step i + 1
 * arrive by LDS-DMA under the MFMAs of tile-step i (two buffer sets), 0 = the round-2 kernels.  Bit-identical results.
 * Returns the previous setting. */
int caiman_lstm_resident_bt_dma(int on);
/* Mode 2 phase timers of the 2-D split kernel: out8[0..5] = 10 ns ticks {wait for the quarter's producers, gather +
 * MFMA, partial blocks out + drain, wait for the group, partial blocks in + epilogue, drain + barrier}, out8[7] =
 * timesteps, out8[8..14] = "gather + MFMA" split into {DMA issue, stage 0 wait, stage 0 MFMA, stage 1 wait, stage 1 MFMA,
 * later waits, later MFMAs}.  `out8` must hold 16 values.  Synchronises the device and clears the counters. */
int caiman_lstm_resident_profile_bwd2(uint32_t* out8);
int caiman_lstm_resident_failures(void);
/* Overwrite the failure count (0 re-admits the resident kernels after an incident); returns the previous value. */
int caiman_lstm_resident_set_failures(int count);
/* Data-parallel agreement on a hand-off timeout: queued in front of the all-reduce of the gradient slice holding
 * `grad_elem`, writes a NaN there when the failure count has moved past `*seen` (device word: work[5] of
 * caiman_lamb_step's scratch, read as uint32), so that every rank's caiman_lamb_step drops the step.  No-op while
 * no resident launch has been attempted on the device. */
int caiman_lstm_resident_poison(float* grad_elem, const uint32_t* seen, caiman_stream_t stream);
/* Test aid (no reference counterpart): hold `workgroups` CUs for `microseconds` on `stream`, as a collective's kernel
 * would while it waits for the slowest rank.  Lets one GPU rehearse "weight-resident LSTM grids + collectives in one job"
 * (tests/test_gpu_distributed.py); every workgroup leaves by its own clock. */
int caiman_debug_occupy_cus(int workgroups, int microseconds, caiman_stream_t stream);
int64_t caiman_lstm_resident_launches(void);
/* 1 when a multi-timestep BACKWARD wave call with n_slots slots of hidden size H and batch B would be one resident
 * launch on the current device (mode on, n_slots * H/32 <= CUs, and B <= 32 with H/32 in {2,4,8,16,24,32}, or
 * 32 < B <= 128 in tiles of 32 batch rows with H in {512, 1024}; the forward call also takes H = 256 in tiles). */
int caiman_lstm_resident_would_run(int64_t B, int64_t H, int n_slots);
/* Diagnostic: mode 2 = mode 1 plus phase timers in one workgroup (slot 0, first slice).  out10[0..4] forward and
 * out[5..9] backward: 10 ns ticks spent {waiting for the peers' hand-off, bringing the operand row into LDS (backward:
 * including the staged MFMAs), in the MFMA + cell update (backward: the epilogue), draining the stores + barrier},
 * summed over timesteps, then the timestep count.  Synchronises the device and clears the counters. */
int caiman_lstm_resident_profile(uint32_t* out10);
/* 1: resident launches are a flat grid whose workgroup -> layer mapping follows the dispatcher's round-robin over the
 * XCDs, so that the workgroups of a layer share one XCD's L2; 0 (default; measured equal or better): grid (H / 32, layers).
 * Placement only changes the speed
 * (the hand-off protocol does not depend on it).  Returns the previous setting. */
int caiman_lstm_resident_xcd_roles(int on);

/* ------------------------------------------------------------------------- *
 * 16-bit operand images of the LSTM parameters, every layer of a stack in one launch (csrc/lstm_images.hip).  The
 * reference casts its fp32 master weights inside each call under autocast (custom_lstm/lstm.py:51-55, 76-140); the layer
 * pipeline wants several layouts of them per training step.  Inputs: the fp32 parameters in the reference layout (rows
 * [gate][unit]).  Outputs (each may be NULL), `dtype` f16 / bf16:
 *   Wt [K, 4H] K-major, columns [unit][gate];  Wn [4H, K] rows [unit][gate];  bias [4H] = b_ih + b_hh, [unit][gate];
 *   Rf / Rb: the recurrent weights as caiman_lstm_prepare(backward = 0 / 1, gate_layout = 1) tiles them -- pass the image
 *   as `weights_tiled` and R = NULL to caiman_lstm_prepare, which then only initialises the rings.
 * H % 32 == 0, K % 4 == 0, 16-byte aligned parameters.
 * ------------------------------------------------------------------------- */
#define CAIMAN_LSTM_IMAGES_MAX_LAYERS 16
typedef struct {
  const float* W_ih; /* [4H, K] */
  const float* W_hh; /* [4H, H] */
  const float* b_ih; /* [4H] (needed when bias != NULL) */
  const float* b_hh;
  void* Wt;
  void* Wn;
  void* bias;
  void* Rf;
  void* Rb;
  int32_t H, K;
} caiman_lstm_images_t;
int caiman_lstm_weight_images(const caiman_lstm_images_t* layers, int n_layers, int dtype, caiman_stream_t stream);
/* The way back for the gradients: dst[(gate * H + unit) * cols + c] += src[(unit * 4 + gate) * cols + c] for up to 8
 * parameters in one launch.  dst: the fp32 `.grad` of a parameter in the reference layout (rows [gate][unit]); src: the
 * gradient as the layer pipeline produces it (rows [unit][gate]), `dtype` (f16 / bf16) or fp32 when src_fp32 != 0 (the bias
 * sums of the backward kernels); cols = K for a weight, 1 for a bias.  Replaces autograd's AccumulateGrad of
 * `dW, dR, dB, dB` (custom_lstm/lstm.py:119-144) plus this repo's un-permutation. */
#define CAIMAN_LSTM_DELIVER_MAX_ITEMS 8
typedef struct {
  const void* src;
  float* dst;
  int32_t H, cols;
  int32_t src_fp32, slabs; /* slabs > 1 (fp32 sources only): src holds that many partial gradients, 4H * cols floats apart, summed in order */
} caiman_lstm_grad_item_t;
int caiman_lstm_grad_deliver(const caiman_lstm_grad_item_t* items, int n_items, int dtype, caiman_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Grouped input-projection GEMM of the layer-pipelined LSTM stacks (csrc/proj_gemm.hip) — replaces, chunk by chunk,
 *   gates = torch.addmm(bias, x, W_ih.t())          training/lib/src/rnnt_ext/custom_lstm/lstm.py:51-55
 *   dx    = dG @ W_ih  (autograd of the same call)   and the StackTime gather / scatter around the first post_rnn
 *                                                    layer (training/caiman_asr_train/rnnt/model.py:314-342)
 * for all layers that work in one pipeline tick, in ONE launch:  C[m][n] = sum_k A[m][k] * W[n][k] (+ bias[n]).
 * W is row-major [N][K]; A and C rows are addressed as (outer, inner, segment):
 *   A(m, k) at a + (m / a_inner) * a_stride_outer + (m % a_inner) * a_stride_inner + (k / a_kseg) * a_stride_seg + k % a_kseg
 *   C(m, n) at c + (m / c_inner) * c_stride_outer + (m % c_inner) * c_stride_inner + (n / c_nseg) * c_stride_seg + n % c_nseg
 * (strides in elements; a plain row-major matrix: inner = M, stride_inner = leading dimension, kseg = K / nseg = N).
 * bf16 / f16 only; N % 128 == 0, K % 128 == 0, a_kseg % 64 == 0, c_nseg % 16 == 0, 16-byte aligned operand rows:
 * caiman_proj_gemm_supported() tells; caiman_proj_gemm() returns CAIMAN_ERR_INVALID for anything else (the caller then
 * keeps the library GEMM).  Put the problems with the longest K first.  tile: 0 = choose (5), 1 = 256 x 128 tiles with two LDS stages, 2 = 128 x 128,
 * 3 = 256 x 128 with three stages (two in flight), 4 = 256 x 128 with 8 waves, 5 = 128 x 128 with 8 waves,
 * 8 = 128 x 128 with two groups of 8 waves that split K inside the workgroup (long K, few tiles; needs K % 256 == 0, else 5).
 * ------------------------------------------------------------------------- */
#define CAIMAN_PROJ_MAX_PROBLEMS 8
typedef struct {
  const void* a;
  const void* w;
  const void* bias; /* [N] in the operand dtype, or NULL */
  void* c;
  int32_t M, N, K;
  int32_t a_inner, a_kseg, c_inner, c_nseg;
  int64_t a_stride_outer, a_stride_inner, a_stride_seg;
  int64_t c_stride_outer, c_stride_inner, c_stride_seg;
} caiman_proj_problem_t;
int caiman_proj_gemm_supported(const caiman_proj_problem_t* problem, int dtype);
int caiman_proj_gemm(const caiman_proj_problem_t* problems, int n_problems, int dtype, int tile, caiman_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Joint projection with the row log-sum-exp in its epilogue (csrc/joint_gemm.hip) — replaces
 *   logits = self.joint_fc(h)                       training/caiman_asr_train/rnnt/model.py:409-439 (torch.nn.Linear)
 *   denom  = logsumexp(logits)  inside the loss      training/lib/csrc/logsumexp.cu:65-105 (its call site
 *                                                    training/lib/src/rnnt_ext/transducer/loss.py)
 * C [M, N] = A [M, K] · W [N, K]^T + bias [N] (operands and C in `dtype`, bf16 / f16; fp32 accumulation); when `lse` is
 * given, lse[m] = log sum_n exp(C[m][n]) of the STORED (rounded) row, fp32, through `workspace`
 * (caiman_joint_fc_workspace_elems(M, N) floats).  lse == NULL: the plain product (the input gradient dY · W of the same
 * projection is this call with the transposed weight copy as W).  N % 256 == 0, K % 128 == 0, K >= 256; A, W and C
 * 16-byte aligned, bias 8-byte.  `workspace` receives the per-64-column partial pairs as [2][N / 64][M] floats.
 * One persistent workgroup per CU walks the 256 x 256 output tiles (CAIMAN_JOINT_WGS overrides the count).
 * ------------------------------------------------------------------------- */
int64_t caiman_joint_fc_workspace_elems(int64_t M, int64_t N);
int caiman_joint_fc_supported(int64_t M, int64_t N, int64_t K, int dtype);
int caiman_joint_fc_forward(const void* A, const void* W, const void* bias, void* C, float* lse, float* workspace,
                            int64_t M, int64_t N, int64_t K, int dtype, caiman_stream_t stream);
/* Weight gradient of the same projection (csrc/joint_wgrad.hip) — replaces autograd's dW = dY^T . h of torch.nn.Linear
 * (training/caiman_asr_train/rnnt/model.py:409-439): dY [M, N], H [M, K] in `dtype`, the reduction over M split into
 * `slices` consecutive row ranges of `rows_per_slice` rows (a multiple of 128 from caiman_joint_fc_wgrad_plan: pairs of the
 * 8-phase kernel's 64-row tiles; multiples of 32 >= 96 are still accepted and run on the round-3 ring kernel; 0 slices =
 * shape not supported: N, K % 256 == 0, M >= 256, bf16 / f16); slabs [slices][N][K] fp32 receives one partial product per slice
 * (written, not accumulated).  The caller adds the slabs in order.  The rows past slices * rows_per_slice (fewer than 128 per
 * slice) are summed into the LAST slab by the 8-phase kernel (its operand descriptors end with row M - 1: the rows of the
 * last tile pair past the end read as zeros); caiman_wgrad_tn_covers_remainder() = 0 (ring kernel) leaves them to the caller. */
int caiman_joint_fc_wgrad_plan(int64_t M, int64_t N, int64_t K, int dtype, int64_t* rows_per_slice);
int caiman_joint_fc_wgrad(const void* dY, const void* H, float* slabs, int64_t M, int64_t N, int64_t K, int slices,
                          int64_t rows_per_slice, int dtype, caiman_stream_t stream);
/* The same kernel for `batch` products of one shape — the LSTM layers' weight gradients dW = dG^T . x, dR = dG^T . h_prev,
 * which the reference takes as `torch.matmul(dG.t(), x)` per layer (training/lib/src/rnnt_ext/custom_lstm/lstm.py:128-142):
 * operand p at dY + p * stride_y / H + p * stride_h (elements), slabs [batch][slices][N][K]. */
int caiman_wgrad_tn_plan(int64_t M, int64_t N, int64_t K, int batch, int dtype, int64_t* rows_per_slice);
/* microseconds the plan's cost model expects (round quantisation over 256 CUs, per-workgroup fixed cost, slabs); < 0: shape
 * not supported.  For callers that keep the kernel to the calls where it is expected to beat their library product. */
double caiman_wgrad_tn_estimate_us(int64_t M, int64_t N, int64_t K, int batch, int dtype);
/* dst[i] += slabs[0][i] + ... + slabs[slices - 1][i] (n floats per slab, n % 4 == 0, 16-byte aligned): the slabs of a
 * weight-gradient call summed in order into the parameter's fp32 gradient — autograd's AccumulateGrad and the reduction
 * in front of it in one pass. */
int caiman_slab_accumulate(const float* slabs, int slices, int64_t n, float* dst, caiman_stream_t stream);
int caiman_wgrad_tn_covers_remainder(int64_t M, int64_t N, int64_t K, int slices, int64_t rows_per_slice);
int caiman_wgrad_tn(const void* dY, int64_t stride_y, const void* H, int64_t stride_h, float* slabs, int batch, int64_t M,
                    int64_t N, int64_t K, int slices, int64_t rows_per_slice, int dtype, caiman_stream_t stream);
/* two strided groups of products of one shape in one launch (the layers' dR and dW share the gradients and the shape, not the
 * activation buffer): products [0, batch) from (dY, H), products [batch, batch + batch2) from (dY2, H2); plan and estimate
 * with batch + batch2; slabs [batch + batch2][slices][N][K]. */
int caiman_wgrad_tn2(const void* dY, int64_t stride_y, const void* H, int64_t stride_h, int batch, const void* dY2,
                     int64_t stride_y2, const void* H2, int64_t stride_h2, int batch2, float* slabs, int64_t M, int64_t N,
                     int64_t K, int slices, int64_t rows_per_slice, int dtype, caiman_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Transducer joint — replaces apex.contrib.transducer.TransducerJoint (third party, not
 * vendored; call sites training/caiman_asr_train/rnnt/model.py:228-238,425-434; CPU
 * equivalent `torch_transducer_joint` + `relu_drop`, model.py:441-447,224).
 * f : [B, T, H]   g : [B, U, H] (U = max label length + 1)   f_len, g_len : [B] int32
 * out : packed [total_rows, H] (row = batch_offset[b-1] + t*g_len[b] + u) or padded
 *       [B, T, U, H] with -1 in the don't-care cells (training/tests/rnnt/test_model.py:34-64)
 * relu / dropout_p : fused ReLU and inverted dropout (p = 0 disables; `seed` drives this
 *       library's counter-based mask generator).
 * ------------------------------------------------------------------------- */
int caiman_joint_forward(const void* f, const void* g, const int32_t* f_len,
                         const int32_t* g_len, const int64_t* batch_offset, int64_t B,
                         int64_t T, int64_t U, int64_t H, int64_t total_rows, int packed,
                         int relu, double dropout_p, uint64_t seed, int dtype, void* out,
                         caiman_stream_t stream);

/* dh : gradient w.r.t. `out` (same layout); h_out : the forward output (mask source).
 * mask_mode 0: none, 1: relu(+dropout) -> (h_out > 0), 2: dropout only -> (h_out != 0);
 * scale = 1/(1-p).  df : [B, T, H], dg : [B, U, H] (rows beyond f_len / g_len zeroed). */
int caiman_joint_backward(const void* dh, const void* h_out, const int32_t* f_len,
                          const int32_t* g_len, const int64_t* batch_offset, int64_t B,
                          int64_t T, int64_t U, int64_t H, int packed, int mask_mode,
                          double scale, int dtype, void* df, void* dg,
                          caiman_stream_t stream);

/* ------------------------------------------------------------------------- *
 * LAMB + EMA step over a flat fp32 arena — replaces apex.optimizers.FusedLAMB (third party;
 * call site training/caiman_asr_train/train_utils/build_optimizer.py:11-32), the finite-
 * gradient skip of training/caiman_asr_train/train_utils/optimizer.py:31-57 and the EMA loop
 * of training/caiman_asr_train/train.py:58-64.
 * p, g, m, v, ema : device arenas of equal length (ema may be NULL); tensors are laid out
 *   back to back at 16-byte aligned offsets and cut into chunks (<= 2^31 elements each):
 *   chunk_start[c] (element offset), chunk_len[c], chunk_tensor[c]; tensor_first_chunk[t]
 *   (n_tensors + 1 entries), tensor_group[t].  These tables live on the DEVICE.
 * group_lr / group_wd : HOST arrays, n_groups <= 16 entries (passed by value to the kernels).
 * inv_grad_scale : 1/loss_scale (1 for bf16).  Non-finite gradient norm => p, m, v and the
 *   device step counter are left untouched (EMA still advances, as in the reference).  The same happens when a
 *   weight-resident LSTM launch of this device has timed out at a hand-off since the previous call
 *   (caiman_lstm_resident_failures moved): such a launch leaves stale but FINITE rows behind, so the decision is
 *   taken on the device from the failure word itself; work[6] = 1 for a step dropped for that reason.
 * zero_grad : also clear g in the last pass (g otherwise holds the LAMB update on return).
 * work : device scratch of 8 + 2*n_chunks + n_tensors floats; work[0] = gradient norm,
 *        work[2] = 1 if the step was applied, work[5] (uint32) = failure count seen, work[6] as above.
 * ------------------------------------------------------------------------- */
int caiman_lamb_step(float* p, float* g, float* m, float* v, float* ema,
                     const int64_t* chunk_start, const int32_t* chunk_len,
                     const int32_t* chunk_tensor, int64_t n_chunks,
                     const int32_t* tensor_group, const int64_t* tensor_first_chunk,
                     int64_t n_tensors, const float* group_lr, const float* group_wd,
                     int n_groups, float beta1, float beta2, float eps, float max_grad_norm,
                     float ema_decay, float inv_grad_scale, int bias_correction,
                     int grad_averaging, int zero_grad, float* work, int32_t* step_counter,
                     caiman_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Log-mel frontend — replaces the DALI operator chain of the reference (third party, not
 * vendored): construction training/caiman_asr_train/data/dali/pipeline.py:260-315, graph tail
 * :439-462; base config training/configs/base-8703sp.yaml:37-49.
 *   [0]*initial_pad ++ samples -> + dither*N(0,1) -> y[n] = x[n] - preemph*x[n-1] (clamped border)
 *   -> frames of win_len every hop (no centring) x window -> zero-pad to nfft -> |FFT|^2
 *   -> nmel triangular filters -> ln(max(., log_floor)) ; frames past the utterance are 0.
 * audio : [B, max_samples] f32, audio_len : [B]; out : [B, nmel, max_frames] f32,
 * out_len[b] = (audio_len[b] + initial_pad - win_len) / hop + 1.
 * window [win_len], tw_cos/tw_sin [nfft/2] (cos / sin of 2*pi*k/nfft), mel_w [nmel, nfft/2+1]
 * dense weights with [mel_lo, mel_hi) the non-zero bin range of each filter: DEVICE tables
 * built once by the caller (caiman_asr_amd/data/frontend.py).
 * ------------------------------------------------------------------------- */
int caiman_logmel_forward(const float* audio, const int32_t* audio_len, int64_t B,
                          int64_t max_samples, int win_len, int hop, int nfft, int nmel,
                          int initial_pad, float preemph, float dither, uint64_t seed,
                          float log_floor, const float* window, const float* tw_cos,
                          const float* tw_sin, const float* mel_w, const int32_t* mel_lo,
                          const int32_t* mel_hi, float* out, int32_t* out_len,
                          int64_t max_frames, caiman_stream_t stream);

/* In-place per-feature normalisation over the valid frames, blended with dataset statistics:
 *   ratio*(x - ds_mean)/ds_std + (1 - ratio)*(x - mean_utt)/std_utt   (population std),
 * training/caiman_asr_train/data/dali/mel_normalization.py:85-118.  ds_* may be NULL when
 * ratio == 0.  Frames >= len[b] are zeroed. */
int caiman_mel_normalize(float* x, const int32_t* len, int64_t B, int nmel, int64_t T,
                         const float* ds_mean, const float* ds_std, float ratio,
                         caiman_stream_t stream);

/* Gradient of an embedding table into its fp32 `.grad` — replaces autograd's embedding_dense_backward + AccumulateGrad for the
 * prediction network's `torch.nn.Embedding` (training/caiman_asr_train/rnnt/model.py, `self.prediction["embed"]`):
 * grad[v][:] += sum over n with tokens[n] == v of dy[n][:], positions in ascending order (fixed summation order).
 * tokens [n] int64; dy [n][E] in `dtype` (CAIMAN_F32 / BF16 / F16), rows contiguous; grad [V][E] f32; tokens outside
 * [0, V) contribute nothing. */
int caiman_embedding_grad(const int64_t* tokens, int64_t n, const void* dy, int dtype, int64_t V, int64_t E, float* grad,
                          caiman_stream_t stream);

/* The (h, c) rows of every utterance's last valid step, all layers of a stack, in one launch — replaces the two
 * advanced-indexing selections of training/caiman_asr_train/train_utils/rsp.py:108-130 (`get_last_nonpadded_states`; with
 * back = 1 the prediction network's next-to-last state, rsp.py:132-205).  h, c: [L][T][B] rows of row_bytes bytes each
 * (rows of a step contiguous; layer and step strides in bytes, so views that skip the initial-state row qualify);  lens [B]
 * (lens_kind 0: int32, 1: int64);  step picked = lens[b] - 1 - back, a negative value counting from the end as in Python
 * indexing;  h_out, c_out: [L][B] rows, contiguous.  B <= 2^31 - 1, L <= 65535. */
int caiman_lstm_last_states(const void* h, const void* c, int64_t L, int64_t T, int64_t B, int64_t row_bytes,
                            int64_t h_stride_l, int64_t h_stride_t, int64_t c_stride_l, int64_t c_stride_t,
                            const void* lens, int lens_kind, int back, void* h_out, void* c_out, caiman_stream_t stream);

/* SpecAugment mask geometry in one launch — the arithmetic of training/caiman_asr_train/data/features.py:60-101 (widths
 * U[min, max], starts U[0, extent - width], adaptive counts / widths as fractions of the utterance length) applied to
 * uniform draws the caller made: rnd [B][2 nf + 2 nt] in [0, 1), columns = draws for fw | f0 | tw | t0;  lens [B] frames per
 * utterance (lens_kind 0: int32, 1: int64, 2: float);  freq_span = max_freq - min_freq + 1;  time_masks: a count (>= 1,
 * nt = that count) or a fraction of the length in (0, 1) (nt = round(T * fraction) + 1 slots, the ones past an utterance's
 * count get width 0);  max_time likewise a width or a fraction.  out: fw [B][nf], f0 [B][nf], tw [B][nt], t0 [B][nt] f32,
 * one after the other — the arrays caiman_specaug_splice takes. */
int caiman_specaug_geometry(const float* rnd, const void* lens, int lens_kind, int64_t B, int64_t F, int64_t T, int nf,
                            float min_freq, float freq_span, float time_masks, int nt, float min_time, float max_time,
                            float* out, caiman_stream_t stream);

/* SpecAugment masks applied + frame splicing + PermuteAudio in one pass — replaces the tail of the reference's feature
 * processors (training/caiman_asr_train/data/features.py:34-115 `SpecAugment.calculate_features`' masked_fill, :118-139
 * `stack_subsample_frames`, :160-162 `PermuteAudio`): x [B, F, T] f32;  f0 / fw [B, nf], t0 / tw [B, nt]: start and width
 * of every frequency / time mask of utterance b (width 0: no mask; the caller draws them);  out [T_out, B, F * stacking] f32
 * with out[t1][b][n * F + f] = masked x[b][f][t1 * subsampling + n], zero past T;  T_out <= ceil(T / subsampling). */
int caiman_specaug_splice(const float* x, int64_t B, int64_t F, int64_t T, const float* f0, const float* fw, int nf,
                          const float* t0, const float* tw, int nt, int stacking, int subsampling, int64_t T_out,
                          float* out, caiman_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CAIMAN_RNNT_H_ */
