/*
 * rnnt_oracle.c — CPU restatement of the reference's native RNN-T operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under caiman_asr_amd/ may link, import or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker.
 *
 * Every function is a sequential, double-precision restatement of one CUDA
 * operator of /root/reference/training/lib/csrc (file:line cited per function).
 * The reference kernels cannot be built here (training/lib/setup.py:10-11 needs
 * CUDA_HOME; there is no CUDA toolchain in this image) and the reference has no
 * CPU implementation of the transducer loss or of logsumexp
 * (training/caiman_asr_train/args/val.py:143-149), so this oracle is pinned by
 *   - brute-force enumeration of all monotone alignments (tests/test_oracle_pin.py),
 *   - torch.logsumexp / torch.nn.LSTM equality, the same differential checks the
 *     reference's own tests use (training/lib/tests/...), and
 *   - finite-difference gradient checks over the reference's modifier grid
 *     (training/lib/tests/transducer/test_loss.py:208-260).
 *
 * Build: see oracle/Makefile (gcc -O2 -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- training/lib/csrc/transducer_loss.cu:46-62 ------------------------- */
static double lse2(double a, double b) {
  return (a >= b) ? a + log1p(exp(b - a)) : b + log1p(exp(a - b));
}
static double frac_penalty(double lam, double t, double T) { return lam * ((T - 1) / 2 - t); }
static double sub_or_nan(double num, double den) { return isfinite(den) ? num - den : NAN; }

/* ---- training/lib/csrc/logsumexp.cu:65-105 ------------------------------ */
static double nan_max(double x, double y) { return isnan(x) ? x : (x > y ? x : y); }

void oracle_logsumexp(const double* in, int64_t rows, int64_t n, int64_t stride, double* out) {
  for (int64_t r = 0; r < rows; ++r) {
    const double* row = in + r * stride;
    double m = -INFINITY;
    for (int64_t i = 0; i < n; ++i) m = nan_max(m, row[i]);
    if (!isfinite(m)) { /* :91-96 */
      out[r] = m;
      continue;
    }
    double s = 0;
    for (int64_t i = 0; i < n; ++i) s += exp(row[i] - m);
    out[r] = m + log(s);
  }
}

/* ---- lattice helpers: transducer_loss.cu:101-173 ------------------------ */
typedef struct {
  const double* x;
  const double* denom;
  const int32_t* label; /* this utterance's labels */
  int64_t stride, V, blank, eos_idx, star_idx, T;
  double dp_lam, eos_lam, star_lam;
} cell_ctx;

static double lsm(const cell_ctx* c, int64_t t, int64_t u, int64_t k) { /* :114-118 */
  return sub_or_nan(c->x[(t * c->stride + u) * c->V + k], c->denom[t * c->stride + u]);
}
static double log_null(const cell_ctx* c, int64_t t, int64_t u) { /* :120-142 */
  double v = lsm(c, t, u, c->blank);
  if (u == 0) return v;
  if (c->label[u - 1] == c->star_idx) return c->star_lam;
  return v;
}
static double log_emit(const cell_ctx* c, int64_t t, int64_t u) { /* :144-173 */
  double dp = frac_penalty(c->dp_lam, (double)t, (double)c->T);
  if (c->label[u] == c->star_idx) return dp;
  double v = lsm(c, t, u, c->label[u]) + dp;
  if (c->label[u] == c->eos_idx) return v + frac_penalty(c->eos_lam, (double)t, (double)c->T);
  return v;
}

/* ---- forward: transducer_loss.cu:175-263 -------------------------------- *
 * alpha/beta are [B, max_flen, max_glen]; only the valid region is written.  */
void oracle_transducer_forward(const double* x, const double* denom, const int32_t* label,
                               const int32_t* f_len, const int32_t* y_len,
                               const int64_t* batch_offset, int64_t B, int64_t max_flen,
                               int64_t max_glen, int64_t V, double dp_lam, int64_t blank,
                               double eos_lam, int64_t eos_idx, double star_lam, int64_t star_idx,
                               int packed, double* alpha, double* beta, double* loss) {
  for (int64_t b = 0; b < B; ++b) {
    const int64_t T = f_len[b], U = y_len[b] + 1;
    const int64_t off = packed ? (b == 0 ? 0 : batch_offset[b - 1]) : b * max_flen * max_glen;
    cell_ctx c = {x + off * V, denom + off, label + b * (max_glen - 1), packed ? U : max_glen,
                  V, blank, eos_idx, star_idx, T, dp_lam, eos_lam, star_lam};
    double* a = alpha + b * max_flen * max_glen;
    double* be = beta + b * max_flen * max_glen;
    a[0] = 0;
    for (int64_t t = 0; t < T; ++t)
      for (int64_t u = 0; u < U; ++u) {
        if (t == 0 && u == 0) continue;
        if (u == 0)
          a[t * max_glen] = a[(t - 1) * max_glen] + log_null(&c, t - 1, 0);
        else if (t == 0)
          a[u] = a[u - 1] + log_emit(&c, 0, u - 1);
        else
          a[t * max_glen + u] = lse2(a[(t - 1) * max_glen + u] + log_null(&c, t - 1, u),
                                     a[t * max_glen + u - 1] + log_emit(&c, t, u - 1));
      }
    be[(T - 1) * max_glen + U - 1] = log_null(&c, T - 1, U - 1);
    for (int64_t t = T - 1; t >= 0; --t)
      for (int64_t u = U - 1; u >= 0; --u) {
        if (t == T - 1 && u == U - 1) continue;
        if (u == U - 1)
          be[t * max_glen + u] = be[(t + 1) * max_glen + u] + log_null(&c, t, u);
        else if (t == T - 1)
          be[t * max_glen + u] = be[t * max_glen + u + 1] + log_emit(&c, t, u);
        else
          be[t * max_glen + u] = lse2(be[(t + 1) * max_glen + u] + log_null(&c, t, u),
                                      be[t * max_glen + u + 1] + log_emit(&c, t, u));
      }
    loss[b] = -be[0];
  }
}

/* ---- backward: transducer_loss.cu:297-394 ------------------------------- *
 * x_grad has the shape of x; padded layout zero-fills the don't-care cells.   */
void oracle_transducer_backward(const double* x, const double* denom, const double* loss_grad,
                                const double* alpha, const double* beta, const int32_t* f_len,
                                const int32_t* y_len, const int32_t* label,
                                const int64_t* batch_offset, int64_t B, int64_t max_flen,
                                int64_t max_glen, int64_t V, double dp_lam, int64_t blank,
                                double eos_lam, int64_t eos_idx, double star_lam,
                                int64_t star_idx, int packed, double* x_grad) {
  for (int64_t b = 0; b < B; ++b) {
    const int64_t T = f_len[b], U = y_len[b] + 1;
    const int64_t off = packed ? (b == 0 ? 0 : batch_offset[b - 1]) : b * max_flen * max_glen;
    const int64_t stride = packed ? U : max_glen;
    const double* a = alpha + b * max_flen * max_glen;
    const double* be = beta + b * max_flen * max_glen;
    const int32_t* lab = label + b * (max_glen - 1);
    const int64_t tmax = packed ? T : max_flen, umax = packed ? U : max_glen;
    for (int64_t t = 0; t < tmax; ++t)
      for (int64_t u = 0; u < umax; ++u) {
        double* gx = x_grad + (off + t * stride + u) * V;
        if (!(t < T && u < U)) {
          for (int64_t h = 0; h < V; ++h) gx[h] = 0; /* :388-393 */
          continue;
        }
        const double* rx = x + (off + t * stride + u) * V;
        const double den = denom[off + t * stride + u];
        const double common = log(loss_grad[b]) + a[t * max_glen + u] - be[0];
        const double beta_TU = be[t * max_glen + u];
        const int32_t labU = (u == 0) ? -1 : lab[u - 1];
        double beta_Tp1U = 0, beta_TUp1 = 0;
        int32_t labUp1 = -4;
        if (t != T - 1) beta_Tp1U = be[(t + 1) * max_glen + u];
        if (u != U - 1) {
          beta_TUp1 = be[t * max_glen + u + 1] + frac_penalty(dp_lam, (double)t, (double)T);
          labUp1 = lab[u];
          if (labUp1 == eos_idx) beta_TUp1 += frac_penalty(eos_lam, (double)t, (double)T);
        }
        for (int64_t h = 0; h < V; ++h) {
          const double grad = common + sub_or_nan(rx[h], den);
          double g = exp(grad + beta_TU);
          if (u != U - 1) {
            if (labUp1 == star_idx || h == labUp1) g -= exp(grad + beta_TUp1);
          }
          if (h == blank || labU == star_idx) {
            const double star_pen = (labU == star_idx) ? star_lam : 0;
            if (t == T - 1 && u == U - 1)
              g -= exp(grad + star_pen);
            else if (t != T - 1)
              g -= exp(grad + beta_Tp1U + star_pen);
          }
          gx[h] = g;
        }
      }
  }
}

/* ---- LSTM cell: training/lib/csrc/lstm.cu:22-76 ------------------------- */
static double clampd(double z, double lo, double hi) { return fmax(lo, fmin(z, hi)); }
static double act_sigm(double z, int hard) { return hard ? clampd(0.5 + z / 8.0, 0, 1) : 1.0 / (1.0 + exp(-z)); }
static double act_tanh(double z, int hard) { return hard ? clampd(z, -1, 1) : tanh(z); }
static double sigm_prime(double a, int hard) { return hard ? ((a == 0 || a == 1) ? 0 : 0.125) : (1 - a) * a; }
static double tanh_prime(double a, int hard) { return hard ? ((a == -1 || a == 1) ? 0 : 1) : 1 - a * a; }

/* forward: lstm.cu:214-272 (time loop) + :85-135 (pointwise).
 * R [4H,H]; gates [T,B,4H] in/out (activated on return); c,y [T+1,B,H], row 0 = initial. */
void oracle_lstm_fwd(const double* R, double* gates, double* c, double* y, int64_t T, int64_t B,
                     int64_t H, int hard) {
  for (int64_t t = 0; t < T; ++t) {
    double* g = gates + t * B * 4 * H;
    const double* yp = y + t * B * H;
    const double* cp = c + t * B * H;
    double* yn = y + (t + 1) * B * H;
    double* cn = c + (t + 1) * B * H;
    for (int64_t b = 0; b < B; ++b) {
      for (int64_t r = 0; r < 4 * H; ++r) { /* gates[t] += y[t] @ R^T, :265 */
        double s = 0;
        for (int64_t k = 0; k < H; ++k) s += yp[b * H + k] * R[r * H + k];
        g[b * 4 * H + r] += s;
      }
      for (int64_t n = 0; n < H; ++n) {
        double* gi = &g[b * 4 * H + 0 * H + n];
        double* gf = &g[b * 4 * H + 1 * H + n];
        double* gg = &g[b * 4 * H + 2 * H + n];
        double* go = &g[b * 4 * H + 3 * H + n];
        const double i = act_sigm(*gi, hard), f = act_sigm(*gf, hard);
        const double gv = act_tanh(*gg, hard), o = act_sigm(*go, hard);
        const double cv = i * gv + f * cp[b * H + n];
        *gi = i; *gf = f; *gg = gv; *go = o; /* :127-130 */
        cn[b * H + n] = cv;
        yn[b * H + n] = o * act_tanh(cv, hard);
      }
    }
  }
}

/* backward: lstm.cu:274-346 (reverse loop) + :137-212 (pointwise).
 * partials [T,B,H] = copy of upstream delta, modified in place; dG [T,B,4H] out. */
void oracle_lstm_bwd(const double* R, const double* gates, const double* c, double* partials,
                     double* dG, int64_t T, int64_t B, int64_t H, int hard) {
  double* dC = (double*)calloc((size_t)(B * H), sizeof(double));
  for (int64_t t = T - 1; t >= 0; --t) {
    double* p = partials + t * B * H;
    if (t < T - 1) { /* partials[t] += dG[t+1] @ R, :325-333 */
      const double* dgn = dG + (t + 1) * B * 4 * H;
      for (int64_t b = 0; b < B; ++b)
        for (int64_t n = 0; n < H; ++n) {
          double s = 0;
          for (int64_t r = 0; r < 4 * H; ++r) s += dgn[b * 4 * H + r] * R[r * H + n];
          p[b * H + n] += s;
        }
    }
    const double* g = gates + t * B * 4 * H;
    const double* cprev = c + t * B * H;
    const double* ccur = c + (t + 1) * B * H;
    double* dg = dG + t * B * 4 * H;
    for (int64_t b = 0; b < B; ++b)
      for (int64_t n = 0; n < H; ++n) {
        const double i = g[b * 4 * H + n], f = g[b * 4 * H + H + n];
        const double gv = g[b * 4 * H + 2 * H + n], o = g[b * 4 * H + 3 * H + n];
        const double dy = p[b * H + n];
        const double c_tanh = act_tanh(ccur[b * H + n], hard);
        const double dO = dy * c_tanh * sigm_prime(o, hard);
        const double dc = dy * o * tanh_prime(c_tanh, hard) + dC[b * H + n];
        dg[b * 4 * H + n] = dc * gv * sigm_prime(i, hard);             /* dI */
        dg[b * 4 * H + H + n] = dc * cprev[b * H + n] * sigm_prime(f, hard); /* dF */
        dg[b * 4 * H + 2 * H + n] = dc * i * tanh_prime(gv, hard);     /* dG */
        dg[b * 4 * H + 3 * H + n] = dO;
        dC[b * H + n] = dc * f;
      }
  }
  free(dC);
}
