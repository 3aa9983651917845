// RNN-T transducer loss (alpha/beta lattice + fused log-softmax backward) for gfx950.
//
// Arithmetic contract: training/lib/csrc/transducer_loss.cu
//   :46-62   log_sum_exp, frac_penalty, sub_or_nan
//   :101-264 forward (alpha / beta anti-diagonal recursion, loss = -beta[0,0])
//   :297-394 fused backward (softmax backward + lattice gradient, delay / EOS / star terms)
// restated in SURVEY.md Appendix A.3.  The kernels below are a new design:
//   forward : one workgroup per (direction, utterance); lane <-> u; the previous
//             anti-diagonal lives in registers + a double-buffered LDS line (the
//             reference round-trips alpha/beta through global memory every step);
//             the two gathered logits + denominators of the NEXT diagonals are
//             prefetched PF steps ahead so the dependent chain is ALU + LDS only.
//   backward: one wave64 per lattice cell row (V logits); 16-byte coalesced
//             loads/stores, no LDS, no barrier; launch covers exactly the packed
//             rows (the reference launches a (Umax+1, Tmax, B) grid and early-outs).
#include "common.h"

namespace caiman {
namespace {

template <typename A>
__device__ __forceinline__ A lse2(A a, A b) {
  // transducer_loss.cu:46-52
  return (a >= b) ? a + log1p(exp(b - a)) : b + log1p(exp(a - b));
}
template <typename A>
__device__ __forceinline__ A frac_penalty(A lam, A t, A T) {
  return lam * ((T - 1) / 2 - t);  // transducer_loss.cu:54-57
}
template <typename A>
__device__ __forceinline__ A sub_or_nan(A num, A den) {
  return isfinite(den) ? num - den : static_cast<A>(NAN);  // transducer_loss.cu:59-62
}

struct LossParams {
  const int32_t* label;
  const int32_t* f_len;
  const int32_t* y_len;
  const int64_t* batch_offset;
  int64_t max_flen, max_glen, V, blank, eos_idx, star_idx;
  double dp_lam, eos_lam, star_lam;
  int packed;
  int batch;
};

// Prefetch distance (anti-diagonals) of the forward kernel = the length of a group of diagonals whose results are stored
// together.  What it takes for the prefetch to exist at all (found in the ISA: the first version waited `vmcnt(0)` right
// behind every load, 1.2 us per diagonal = the latency of a gather from the 5 GB logits): (1) the ring keeps RAW loaded
// values and converts at use (a conversion at load time needs the data at load time); (2) the loads are unconditional,
// from clamped addresses (a load under a per-lane condition is a branch with a full wait at its join); (3) no store
// inside a group: loads and stores share `vmcnt` and complete out of order with each other, so with a store pending the
// compiler answers every wait for a load with `vmcnt(0)` -- a lane keeps the group's results in registers and stores them
// behind the group's last diagonal (one exposed load latency per kPF diagonals instead of one per diagonal).
constexpr int kPF = 8;

// ---------------------------------------------------------------------------
// forward: grid (2, B); blockIdx.x = 0 -> alpha, 1 -> beta. blockDim = NT >= U'.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void loss_fwd_kernel(const T* __restrict__ x,
                                                        const acc_t<T>* __restrict__ denom,
                                                        LossParams p, acc_t<T>* __restrict__ alpha,
                                                        acc_t<T>* __restrict__ beta,
                                                        acc_t<T>* __restrict__ loss) {
  using A = acc_t<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  A* line = reinterpret_cast<A*>(smem_raw);  // [2][blockDim.x + 2]

  const int b = blockIdx.y;
  const int u = threadIdx.x;
  const int NT = blockDim.x;
  const int64_t Tn = p.f_len[b];
  const int64_t Un = p.y_len[b] + 1;  // U' = labels + SOS
  const int32_t* lab = p.label + (int64_t)b * (p.max_glen - 1);
  const int64_t off = p.packed ? (b == 0 ? 0 : p.batch_offset[b - 1]) : (int64_t)b * p.max_flen * p.max_glen;
  const int64_t stride = p.packed ? Un : p.max_glen;
  // An utterance without cells (f_len == 0) owns no block of the packed buffers: its lanes are all inactive, but the
  // prefetch ring below loads "cell 0 of the utterance" unconditionally -- point it at cell 0 of the whole buffer, which
  // always exists (the last utterance of a packed batch would otherwise read past the logits and the denominators).
  const bool has_cells = Tn > 0 && Un > 0;
  const T* xb = has_cells ? x + off * p.V : x;
  const A* db = has_cells ? denom + off : denom;
  const A dp_lam = (A)p.dp_lam, eos_lam = (A)p.eos_lam, star_lam = (A)p.star_lam;
  const A Tf = (A)Tn;

  const bool active = u < Un;
  // per-lane label facts
  const int32_t lab_u = (active && u < Un - 1) ? lab[u] : -3;     // label emitted when leaving (.,u) upward
  const int32_t lab_um1 = (active && u > 0) ? lab[u - 1] : -3;    // label that brought us to row u
  const bool row_is_star = (u > 0) && (lab_um1 == p.star_idx);    // null(t,u) = star_lam
  A* buf0 = line;
  A* buf1 = line + (NT + 2);

  if (blockIdx.x == 0) {
    // ------------------------------ alpha ----------------------------------
    A* my_alpha = alpha + (int64_t)b * p.max_flen * p.max_glen;
    // cell (t,u) on diagonal s = t + u needs null(t-1,u) and emit(t,u-1).
    // ring slot k holds the raw operands for diagonal s with (s % kPF == k).
    T xn[kPF], xe[kPF];
    A dn[kPF], de[kPF];
    const bool emit_is_star = (u > 0) && (lab_um1 == p.star_idx);  // emit(t,u-1) with label[u-1]==star
    const int64_t xe_col = (u > 0 && !emit_is_star && active) ? lab_um1 : 0;
    auto issue = [&](int64_t s, int k) {
      const int64_t t = s - u;
      const bool in = active && t >= 0 && t < Tn;
      const int64_t cn = (in && t > 0) ? (t - 1) * stride + u : 0;        // cell 0 of the utterance: always there
      const int64_t ce = (in && u > 0) ? t * stride + (u - 1) : 0;
      xn[k] = xb[cn * p.V + p.blank];
      dn[k] = db[cn];
      de[k] = db[ce];
      xe[k] = xb[ce * p.V + xe_col];
    };
    const int64_t nsteps = Tn + Un - 1;  // diagonals 0 .. nsteps-1
#pragma unroll
    for (int k = 0; k < kPF; ++k) issue(1 + k, (1 + k) % kPF);

    A mine = 0;  // alpha(t,u) of the previous diagonal for this lane (t-1,u)
    A hist[kPF];
    if (u == 0) my_alpha[0] = 0;
    buf0[u + 1] = (u == 0) ? (A)0 : (A)0;  // diagonal 0: only (0,0) is valid
    __syncthreads();

    for (int64_t s0 = 1; s0 < nsteps; s0 += kPF) {
      unsigned stored = 0;
#pragma unroll
      for (int k = 0; k < kPF; ++k) {
        const int64_t s = s0 + k;
        if (s >= nsteps) break;
        const int slot = (1 + k) % kPF;  // s % kPF because s0 ≡ 1 (mod kPF)
        A* prev = ((s - 1) & 1) ? buf1 : buf0;
        A* cur = (s & 1) ? buf1 : buf0;
        const int64_t t = s - u;
        const bool valid = active && t >= 0 && t < Tn;
        A val = mine;
        if (valid) {
          const A left = prev[u];  // alpha(t, u-1) (slot u holds lane u-1's value)
          A a_null = 0, a_emit = 0;
          if (t > 0) {
            const A lp = sub_or_nan<A>(static_cast<A>(xn[slot]), dn[slot]);
            const A nul = (u == 0) ? lp : (row_is_star ? star_lam : lp);
            a_null = mine + nul;
          }
          if (u > 0) {
            const A dp = frac_penalty<A>(dp_lam, (A)t, Tf);
            A em;
            if (emit_is_star) {
              em = dp;
            } else {
              em = sub_or_nan<A>(static_cast<A>(xe[slot]), de[slot]) + dp;
              if (lab_um1 == p.eos_idx) em += frac_penalty<A>(eos_lam, (A)t, Tf);
            }
            a_emit = left + em;
          }
          val = (u == 0) ? a_null : (t == 0 ? a_emit : lse2<A>(a_null, a_emit));
          hist[k] = val;
          stored |= 1u << k;
        }
        mine = val;
        cur[u + 1] = val;
        issue(s + kPF, slot);
        __syncthreads();
      }
#pragma unroll
      for (int k = 0; k < kPF; ++k)
        if ((stored >> k) & 1u) my_alpha[(s0 + k - u) * p.max_glen + u] = hist[k];
    }
  } else {
    // ------------------------------ beta -----------------------------------
    A* my_beta = beta + (int64_t)b * p.max_flen * p.max_glen;
    // cell (t,u) on diagonal s = t + u needs null(t,u) and emit(t,u) (same cell).
    T xn[kPF], xe[kPF];
    A dd[kPF];
    const bool emit_is_star = active && (u < Un - 1) && (lab_u == p.star_idx);
    const int64_t xe_col = (active && u < Un - 1 && !emit_is_star) ? lab_u : 0;
    auto issue = [&](int64_t s, int k) {
      const int64_t t = s - u;
      const bool in = active && s >= 0 && t >= 0 && t < Tn;
      const int64_t c = in ? t * stride + u : 0;                          // cell 0 of the utterance: always there
      dd[k] = db[c];
      xn[k] = xb[c * p.V + p.blank];
      xe[k] = xb[c * p.V + xe_col];
    };
    const int64_t top = Tn + Un - 2;  // diagonal of the terminal cell
    // Walk diagonals top, top-1, ..., 0. Use j = top - s as the ascending counter
    // so ring slots are static: slot(j) = j % kPF.
#pragma unroll
    for (int k = 0; k < kPF; ++k) issue(top - k, k);

    A mine = 0;  // beta(t+1, u) for this lane
    A hist[kPF];
    for (int64_t j0 = 0; j0 <= top; j0 += kPF) {
      unsigned stored = 0;
#pragma unroll
      for (int k = 0; k < kPF; ++k) {
        const int64_t j = j0 + k;
        if (j > top) break;
        const int64_t s = top - j;
        A* prev = ((j + 1) & 1) ? buf1 : buf0;  // written at step j-1
        A* cur = (j & 1) ? buf1 : buf0;
        const int64_t t = s - u;
        const bool valid = active && t >= 0 && t < Tn;
        A val = mine;
        if (valid) {
          const A lp_blank = sub_or_nan<A>(static_cast<A>(xn[k]), dd[k]);
          const A nul = (u == 0) ? lp_blank : (row_is_star ? star_lam : lp_blank);
          if (t == Tn - 1 && u == Un - 1) {
            val = nul;  // transducer_loss.cu:228
          } else {
            A em = 0;
            if (u < Un - 1) {
              const A dp = frac_penalty<A>(dp_lam, (A)t, Tf);
              if (emit_is_star) {
                em = dp;
              } else {
                em = sub_or_nan<A>(static_cast<A>(xe[k]), dd[k]) + dp;
                if (lab_u == p.eos_idx) em += frac_penalty<A>(eos_lam, (A)t, Tf);
              }
            }
            if (u == Un - 1) {
              val = mine + nul;
            } else {
              const A up = prev[u + 2];  // beta(t, u+1) from lane u+1, previous step
              val = (t == Tn - 1) ? up + em : lse2<A>(mine + nul, up + em);
            }
          }
          hist[k] = val;
          stored |= 1u << k;
        }
        mine = val;
        cur[u + 1] = val;
        issue(s - kPF, k);
        __syncthreads();
      }
#pragma unroll
      for (int k = 0; k < kPF; ++k)
        if ((stored >> k) & 1u) my_beta[(top - (j0 + k) - u) * p.max_glen + u] = hist[k];
    }
    if (u == 0) loss[b] = -mine;  // lane 0 finishes on cell (0,0): transducer_loss.cu:260-262
  }
}

// ---------------------------------------------------------------------------
// backward: one wave per [*, V] row.
// ---------------------------------------------------------------------------
template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) vecT {
  T v[VEC];
};

template <typename T, int VEC>
__global__ __launch_bounds__(256) void loss_bwd_kernel(
    const T* __restrict__ x, const acc_t<T>* __restrict__ denom,
    const acc_t<T>* __restrict__ loss_grad, const acc_t<T>* __restrict__ alpha,
    const acc_t<T>* __restrict__ beta, LossParams p, int64_t total_rows, T* __restrict__ x_grad) {
  using A = acc_t<T>;
  using Vt = vecT<T, VEC>;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  if (row >= total_rows) return;

  // ---- locate (b, t, u) for this row -------------------------------------------
  int b;
  int64_t local;
  int64_t Un_stride;
  if (p.packed) {
    // smallest b with batch_offset[b] > row (batch_offset is an inclusive cumsum)
    int lo = 0, hi = p.batch - 1;  // the last utterance needs no comparison
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (p.batch_offset[mid] > row) hi = mid; else lo = mid + 1;
    }
    b = lo;
    local = row - (b == 0 ? 0 : p.batch_offset[b - 1]);
    Un_stride = p.y_len[b] + 1;
  } else {
    const int64_t per = p.max_flen * p.max_glen;
    b = (int)(row / per);
    local = row - (int64_t)b * per;
    Un_stride = p.max_glen;
  }
  const int64_t Tn = p.f_len[b];
  const int64_t Un = p.y_len[b] + 1;
  const int64_t t = local / Un_stride;
  const int64_t u = local - t * Un_stride;
  T* gx = x_grad + row * p.V;
  const int64_t nfull = p.V / VEC;

  if (t >= Tn || u >= Un) {
    // padded layout: zero the don't-care region (transducer_loss.cu:388-393)
    Vt z;
#pragma unroll
    for (int j = 0; j < VEC; ++j) z.v[j] = static_cast<T>(0);
    for (int64_t c = lane; c < nfull; c += kWave) *reinterpret_cast<Vt*>(gx + c * VEC) = z;
    for (int64_t h = nfull * VEC + lane; h < p.V; h += kWave) gx[h] = static_cast<T>(0);
    return;
  }

  const T* rx = x + row * p.V;
  const A* my_alpha = alpha + (int64_t)b * p.max_flen * p.max_glen;
  const A* my_beta = beta + (int64_t)b * p.max_flen * p.max_glen;
  const int32_t* lab = p.label + (int64_t)b * (p.max_glen - 1);
  const A dp_lam = (A)p.dp_lam, eos_lam = (A)p.eos_lam, star_lam = (A)p.star_lam;
  const A Tf = (A)Tn;

  // wave-uniform scalars (every lane loads the same addresses: broadcast)
  const A den = denom[row];
  const bool den_ok = isfinite(den);
  const A common = log(loss_grad[b]) + my_alpha[t * p.max_glen + u] - my_beta[0];
  const A beta_TU = my_beta[t * p.max_glen + u];
  const int32_t labU = (u == 0) ? -1 : lab[u - 1];
  const bool not_top = (u != Un - 1);
  const bool last_t = (t == Tn - 1);
  A beta_TUp1 = 0, beta_Tp1U = 0;
  int32_t labUp1 = -4;
  if (!last_t) beta_Tp1U = my_beta[(t + 1) * p.max_glen + u];
  if (not_top) {
    beta_TUp1 = my_beta[t * p.max_glen + u + 1] + frac_penalty<A>(dp_lam, (A)t, Tf);
    labUp1 = lab[u];
    if (labUp1 == p.eos_idx) beta_TUp1 += frac_penalty<A>(eos_lam, (A)t, Tf);
  }
  const bool up_all = not_top && (labUp1 == p.star_idx);
  const bool right_all = (labU == p.star_idx);
  const A star_pen = right_all ? star_lam : (A)0;
  // right contribution exists for the terminal cell or any non-last frame
  const bool right_term = last_t && !not_top;
  const bool right_any = right_term || !last_t;
  const A right_add = right_term ? star_pen : beta_Tp1U + star_pen;

  // The result is rounded to T: for the 16-bit types the hardware exponential (v_exp_f32, ~1 ulp f32) is exact
  // after rounding; f32 / f64 keep the libm exp.  The bulk of the row takes one exp per element: only the
  // label column, the blank column and star rows subtract a second / third term.
  auto fexp = [](A v) -> A {
    if constexpr (sizeof(T) == 2) return __expf(v);
    else return exp(v);
  };
  const A base = common + beta_TU - den;  // grad + beta_TU = x + base
  auto one = [&](A xv, int64_t h) -> A {
    if (!den_ok) return static_cast<A>(NAN);
    A g = fexp(xv + base);
    if (not_top && (up_all || h == labUp1)) g -= fexp(xv + (common - den + beta_TUp1));
    if (right_any && (right_all || h == p.blank)) g -= fexp(xv + (common - den + right_add));
    return g;
  };

  // 4 independent 16-byte loads in flight per lane before the first exp (HBM-bound streaming kernel)
  constexpr int UN = 4;
  for (int64_t c0 = lane; c0 < nfull; c0 += UN * kWave) {
    Vt v[UN];
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const int64_t c = c0 + (int64_t)k * kWave;
      if (c < nfull) {
        if constexpr (sizeof(Vt) == 16) {   // streamed once: keep it out of the caches' way
          using u4 = __attribute__((ext_vector_type(4))) unsigned;
          const u4 raw = __builtin_nontemporal_load(reinterpret_cast<const u4*>(rx + c * VEC));
          __builtin_memcpy(&v[k], &raw, 16);
        } else {
          v[k] = *reinterpret_cast<const Vt*>(rx + c * VEC);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const int64_t c = c0 + (int64_t)k * kWave;
      if (c < nfull) {
        Vt o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.v[j] = static_cast<T>(one(static_cast<A>(v[k].v[j]), c * VEC + j));
        if constexpr (sizeof(Vt) == 16) {
          using u4 = __attribute__((ext_vector_type(4))) unsigned;
          u4 raw;
          __builtin_memcpy(&raw, &o, 16);
          __builtin_nontemporal_store(raw, reinterpret_cast<u4*>(gx + c * VEC));
        } else {
          *reinterpret_cast<Vt*>(gx + c * VEC) = o;
        }
      }
    }
  }
  for (int64_t h = nfull * VEC + lane; h < p.V; h += kWave) gx[h] = static_cast<T>(one(static_cast<A>(rx[h]), h));
}

// ---------------------------------------------------------------------------
// backward + column sums of the gradient: the bias gradient of the projection that produced x, which the reference
// obtains from a separate pass over the whole gradient (autograd's grad_output.sum(0) for joint_fc.bias; 5.3 GB /
// 0.93 ms per step at B = 32).
//   1. loss_row_desc_kernel: one thread per row reduces everything that is constant along a row (utterance lookup,
//      lattice scalars, penalties, label / blank / star rules) to a 32-byte descriptor: the gradient of an element is
//      exp(x + base) - [up rule](exp(x + up_off)) - [right rule](exp(x + right_off)).
//   2. loss_bwd_colsum_kernel: a workgroup of 4 waves takes `rows_per_block` consecutive rows; each wave owns a quarter
//      of the 16-byte column chunks and walks all of the workgroup's rows on its own (no barriers), so a lane meets the
//      same <= KACC chunks in every row and keeps their sums in registers (40 floats at V = 8704).  One partial row [V]
//      per workgroup is left for the caller to add up, in a fixed order.
// Tried first: one wave per row summing into an LDS array with ds_add_f32 (7x slower: 16-way bank conflicts on top of
// the atomic rate); then the per-row scalar code of the plain kernel inside the column-quarter loop (226 VGPRs, two
// waves per SIMD, a full vmcnt(0) drain per row: 2.2x slower than the plain kernel).  The sums are taken over the values
// as rounded to T, as the separate pass would see them.
// ---------------------------------------------------------------------------
template <typename A>
struct alignas(32) RowDesc {
  A base, up_off, right_off;
  int32_t lab_up;       // -1: no "up" term, -2: every column (star), else the label column
  int32_t right_mode;   // 0: no "right" term, 1: blank column only, 2: every column (star), 3: don't-care row (zeros)
};
static_assert(sizeof(RowDesc<float>) == 32 && sizeof(RowDesc<double>) == 32, "descriptor size");

template <typename T>
__global__ __launch_bounds__(256) void loss_row_desc_kernel(const acc_t<T>* __restrict__ denom,
                                                            const acc_t<T>* __restrict__ loss_grad,
                                                            const acc_t<T>* __restrict__ alpha,
                                                            const acc_t<T>* __restrict__ beta, LossParams p,
                                                            int64_t total_rows, RowDesc<acc_t<T>>* __restrict__ desc) {
  using A = acc_t<T>;
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= total_rows) return;
  int b;
  int64_t local, Un_stride;
  if (p.packed) {
    int lo = 0, hi = p.batch - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (p.batch_offset[mid] > row) hi = mid; else lo = mid + 1;
    }
    b = lo;
    local = row - (b == 0 ? 0 : p.batch_offset[b - 1]);
    Un_stride = p.y_len[b] + 1;
  } else {
    const int64_t per = p.max_flen * p.max_glen;
    b = (int)(row / per);
    local = row - (int64_t)b * per;
    Un_stride = p.max_glen;
  }
  const int64_t Tn = p.f_len[b];
  const int64_t Un = p.y_len[b] + 1;
  const int64_t t = local / Un_stride;
  const int64_t u = local - t * Un_stride;
  RowDesc<A> d;
  d.base = d.up_off = d.right_off = (A)0;
  d.lab_up = -1;
  d.right_mode = 3;
  if (t < Tn && u < Un) {
    // same arithmetic, in the same order, as loss_bwd_kernel (transducer_loss.cu:297-387)
    const A* my_alpha = alpha + (int64_t)b * p.max_flen * p.max_glen;
    const A* my_beta = beta + (int64_t)b * p.max_flen * p.max_glen;
    const int32_t* lab = p.label + (int64_t)b * (p.max_glen - 1);
    const A dp_lam = (A)p.dp_lam, eos_lam = (A)p.eos_lam, star_lam = (A)p.star_lam;
    const A Tf = (A)Tn;
    const A den = denom[row];
    const A common = log(loss_grad[b]) + my_alpha[t * p.max_glen + u] - my_beta[0];
    const A beta_TU = my_beta[t * p.max_glen + u];
    const int32_t labU = (u == 0) ? -1 : lab[u - 1];
    const bool not_top = (u != Un - 1);
    const bool last_t = (t == Tn - 1);
    A beta_TUp1 = 0, beta_Tp1U = 0;
    int32_t labUp1 = -4;
    if (!last_t) beta_Tp1U = my_beta[(t + 1) * p.max_glen + u];
    if (not_top) {
      beta_TUp1 = my_beta[t * p.max_glen + u + 1] + frac_penalty<A>(dp_lam, (A)t, Tf);
      labUp1 = lab[u];
      if (labUp1 == p.eos_idx) beta_TUp1 += frac_penalty<A>(eos_lam, (A)t, Tf);
    }
    const bool up_all = not_top && (labUp1 == p.star_idx);
    const bool right_all = (labU == p.star_idx);
    const A star_pen = right_all ? star_lam : (A)0;
    const bool right_term = last_t && !not_top;
    const bool right_any = right_term || !last_t;
    const A right_add = right_term ? star_pen : beta_Tp1U + star_pen;
    const A nan = static_cast<A>(NAN);
    const bool den_ok = isfinite(den);
    d.base = den_ok ? common + beta_TU - den : nan;          // a non-finite normaliser makes the whole row NaN
    d.up_off = common - den + beta_TUp1;
    d.right_off = common - den + right_add;
    d.lab_up = !not_top ? -1 : (up_all ? -2 : labUp1);
    d.right_mode = !right_any ? 0 : (right_all ? 2 : 1);
  }
  desc[row] = d;
}

template <typename T, int VEC, int KACC, int NW>
__global__ __launch_bounds__(64 * NW) void loss_bwd_colsum_kernel(const T* __restrict__ x,
                                                              const RowDesc<acc_t<T>>* __restrict__ desc, int64_t V,
                                                              int64_t blank, int64_t total_rows, T* __restrict__ x_grad,
                                                              int rows_per_block, float* __restrict__ colsum_partial) {
  using A = acc_t<T>;
  using Vt = vecT<T, VEC>;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave, nwave = NW;
  // 32-bit column arithmetic throughout (V < 2^31): 64-bit per-element invariants cost dozens of VGPRs
  const int nchunk = (int)(V / VEC);                   // the launch guarantees V % VEC == 0
  const int quarter = (nchunk + nwave - 1) / nwave;    // and KACC * 64 >= quarter (each wave's share of the chunks)
  const int cbeg = wave * quarter;
  const int cend = cbeg + quarter < nchunk ? cbeg + quarter : nchunk;
  const int blank_col = (int)blank;
  const int64_t row_begin = (int64_t)blockIdx.x * rows_per_block;
  const int64_t row_end = row_begin + rows_per_block < total_rows ? row_begin + rows_per_block : total_rows;
  float acc[KACC][VEC];
#pragma unroll
  for (int k = 0; k < KACC; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[k][j] = 0.f;
  auto fexp = [](A v) -> A {
    if constexpr (sizeof(T) == 2) return __expf(v);
    else return exp(v);
  };
  const int c0 = cbeg + lane;
  for (int64_t row = row_begin; row < row_end; ++row) {
    const RowDesc<A> d = desc[row];     // wave-uniform
    const T* rx = x + row * V + (int64_t)c0 * VEC;
    T* gx = x_grad + row * V + (int64_t)c0 * VEC;
    if (d.right_mode == 3) {   // padded layout: zero the don't-care region (transducer_loss.cu:388-393)
      Vt z;
#pragma unroll
      for (int j = 0; j < VEC; ++j) z.v[j] = static_cast<T>(0);
#pragma unroll
      for (int k = 0; k < KACC; ++k)
        if (c0 + k * kWave < cend) *reinterpret_cast<Vt*>(gx + (int64_t)k * kWave * VEC) = z;
      continue;
    }
    Vt v[KACC];
#pragma unroll
    for (int k = 0; k < KACC; ++k) {
      if (c0 + k * kWave < cend) {
        if constexpr (sizeof(Vt) == 16) {
          using u4 = __attribute__((ext_vector_type(4))) unsigned;
          const u4 raw = __builtin_nontemporal_load(reinterpret_cast<const u4*>(rx + (int64_t)k * kWave * VEC));
          __builtin_memcpy(&v[k], &raw, 16);
        } else {
          v[k] = *reinterpret_cast<const Vt*>(rx + (int64_t)k * kWave * VEC);
        }
      }
    }
    const bool up_any = d.lab_up != -1, up_all = d.lab_up == -2;
    const bool right_any = d.right_mode != 0, right_all = d.right_mode == 2;
#pragma unroll
    for (int k = 0; k < KACC; ++k) {
      if (c0 + k * kWave < cend) {
        // Every element takes exp(x + base); the second / third term touches ONE column of the row (the label, the
        // blank) unless the row is a star row.  Whether that column lies in the 64 chunks this wave-instruction covers
        // is wave-uniform: a scalar branch around a block of selects, instead of per-element control flow.
        const int w0 = (cbeg + k * kWave) * VEC, w1 = w0 + kWave * VEC;   // columns of chunk row k
        // element j of this lane's chunk is column h0 + j: compare j with (column - h0), one 32-bit value per lane
        // (64-bit column numbers for all KACC x VEC elements cost 80 VGPRs as loop invariants)
        const int h0 = (c0 + k * kWave) * VEC;
        const int rel_up = d.lab_up - h0, rel_blank = blank_col - h0;
        A g[VEC], xv[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          xv[j] = static_cast<A>(v[k].v[j]);
          g[j] = fexp(xv[j] + d.base);
        }
        if (up_any && (up_all || (d.lab_up >= w0 && d.lab_up < w1))) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const A e = fexp(xv[j] + d.up_off);
            g[j] -= (up_all || rel_up == j) ? e : (A)0;
          }
        }
        if (right_any && (right_all || (blank_col >= w0 && blank_col < w1))) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const A e = fexp(xv[j] + d.right_off);
            g[j] -= (right_all || rel_blank == j) ? e : (A)0;
          }
        }
        Vt o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          o.v[j] = static_cast<T>(g[j]);
          acc[k][j] += static_cast<float>(o.v[j]);
        }
        if constexpr (sizeof(Vt) == 16) {
          using u4 = __attribute__((ext_vector_type(4))) unsigned;
          u4 raw;
          __builtin_memcpy(&raw, &o, 16);
          __builtin_nontemporal_store(raw, reinterpret_cast<u4*>(gx + (int64_t)k * kWave * VEC));
        } else {
          *reinterpret_cast<Vt*>(gx + (int64_t)k * kWave * VEC) = o;
        }
      }
      // one chunk at a time: left to itself the scheduler interleaves the exps of all KACC chunks (251 VGPRs, one
      // wave per SIMD at KACC = 5)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (row_begin < row_end) {
    float* out = colsum_partial + (int64_t)blockIdx.x * V + (int64_t)c0 * VEC;
#pragma unroll
    for (int k = 0; k < KACC; ++k) {
      if (c0 + k * kWave < cend) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) out[(int64_t)k * kWave * VEC + j] = acc[k][j];
      }
    }
  }
}

int check_common(int64_t batch, int64_t max_f_len, int64_t max_g_len, int64_t V, int64_t blank,
                 int64_t eos_idx, int64_t star_idx) {
  CAIMAN_CHECK(batch >= 1 && max_f_len >= 1 && max_g_len >= 1 && V >= 1, "transducer_loss: bad extents");
  // transducer_loss.cu:429-448
  CAIMAN_CHECK(blank >= 0 && blank < V, "Expected blank index to be in the range of 0 to %lld, but got %lld",
               (long long)(V - 1), (long long)blank);
  CAIMAN_CHECK(eos_idx < V, "Expected eos index to be less than %lld, but got %lld", (long long)(V - 1),
               (long long)eos_idx);
  CAIMAN_CHECK(star_idx < V, "Expected star index to be less than %lld, but got %lld", (long long)(V - 1),
               (long long)star_idx);
  return CAIMAN_OK;
}

}  // namespace
}  // namespace caiman

extern "C" int caiman_transducer_loss_forward(
    const void* x, const void* denom, const int32_t* label, const int32_t* f_len, const int32_t* y_len,
    const int64_t* batch_offset, int64_t batch, int64_t max_f_len, int64_t max_g_len, int64_t dict_size,
    double dp_lam, int64_t blank_idx, double eos_lam, int64_t eos_idx, double star_lam, int64_t star_idx,
    int packed, int dtype, void* alpha, void* beta, void* loss, caiman_stream_t stream) {
  using namespace caiman;
  if (int e = check_common(batch, max_f_len, max_g_len, dict_size, blank_idx, eos_idx, star_idx)) return e;
  CAIMAN_CHECK(x && denom && label && f_len && y_len && alpha && beta && loss, "transducer_loss_forward: null pointer");
  CAIMAN_CHECK(!packed || batch_offset, "transducer_loss_forward: packed input needs batch_offset");
  CAIMAN_CHECK(max_g_len <= 1024, "transducer_loss_forward: max_g_len %lld > 1024 not supported",
               (long long)max_g_len);
  CAIMAN_CHECK(batch <= 65535, "transducer_loss_forward: batch too large for one launch");
  LossParams p{label, f_len, y_len, batch_offset, max_f_len, max_g_len, dict_size, blank_idx,
               eos_idx, star_idx, dp_lam, eos_lam, star_lam, packed, (int)batch};
  const int nt = (int)(((max_g_len + kWave - 1) / kWave) * kWave);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return CAIMAN_DISPATCH(dtype, "transducer_loss_forward", [&]() -> int {
    using A = acc_t<scalar_t>;
    const size_t shm = 2 * (size_t)(nt + 2) * sizeof(A);
    hipLaunchKernelGGL((loss_fwd_kernel<scalar_t>), dim3(2, (unsigned)batch), dim3(nt), shm, s,
                       static_cast<const scalar_t*>(x), static_cast<const A*>(denom), p,
                       static_cast<A*>(alpha), static_cast<A*>(beta), static_cast<A*>(loss));
    return check_launch("caiman_transducer_loss_forward");
  });
}

namespace caiman {
namespace {
int loss_backward_impl(const void* x, const void* denom, const void* loss_grad, const void* alpha, const void* beta,
                       const int32_t* f_len, const int32_t* y_len, const int32_t* label, const int64_t* batch_offset,
                       int64_t batch, int64_t max_f_len, int64_t max_g_len, int64_t dict_size, int64_t total_rows,
                       double dp_lam, int64_t blank_idx, double eos_lam, int64_t eos_idx, double star_lam, int64_t star_idx,
                       int packed, int dtype, void* x_grad, float* colsum_partial, int64_t rows_per_block,
                       caiman_stream_t stream) {
  if (int e = check_common(batch, max_f_len, max_g_len, dict_size, blank_idx, eos_idx, star_idx)) return e;
  CAIMAN_CHECK(x && denom && loss_grad && alpha && beta && label && f_len && y_len && x_grad,
               "transducer_loss_backward: null pointer");
  CAIMAN_CHECK(!packed || batch_offset, "transducer_loss_backward: packed input needs batch_offset");
  CAIMAN_CHECK(total_rows >= 0, "transducer_loss_backward: negative row count");
  CAIMAN_CHECK(packed || total_rows == batch * max_f_len * max_g_len,
               "transducer_loss_backward: padded input must have B*T*U rows");
  CAIMAN_CHECK(batch <= 65535, "transducer_loss_backward: batch too large for one launch");
  CAIMAN_CHECK(rows_per_block >= 4 && rows_per_block % 4 == 0 && rows_per_block <= 4096,
               "transducer_loss_backward: rows_per_block must be a multiple of 4 in [4, 4096]");
  if (total_rows == 0) return CAIMAN_OK;
  LossParams p{label, f_len, y_len, batch_offset, max_f_len, max_g_len, dict_size, blank_idx,
               eos_idx, star_idx, dp_lam, eos_lam, star_lam, packed, (int)batch};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t nblk = (total_rows + rows_per_block - 1) / rows_per_block;
  CAIMAN_CHECK(nblk < ((int64_t)1 << 31), "transducer_loss_backward: too many rows for one launch");
  return CAIMAN_DISPATCH(dtype, "transducer_loss_backward", [&]() -> int {
    using A = acc_t<scalar_t>;
    constexpr int VEC = 16 / sizeof(scalar_t);
    const bool aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                         (reinterpret_cast<uintptr_t>(x_grad) % 16 == 0) &&
                         ((dict_size * (int64_t)sizeof(scalar_t)) % 16 == 0);
    const dim3 grid((unsigned)nblk);
    auto xs = static_cast<const scalar_t*>(x);
    auto gs = static_cast<scalar_t*>(x_grad);
    using AA = const A*;
    if (colsum_partial) {
      // a lane keeps the sums of its column chunks in registers: chunks per lane = ceil(ceil(V / VEC / waves) / 64 lanes).
      // Four waves per workgroup up to 5 chunks per lane (V <= 10 240 at 2 bytes); eight waves past that, which keeps
      // 17 408 columns (large-196M) at 5 chunks per lane and doubles the widest row the fused sums take.  At 17 408 the
      // two forms measure the same (4.9 ms for 21 GB of logits + gradient, 4.6 TB/s, as the four-wave kernel at 8704).
      CAIMAN_CHECK(aligned, "transducer_loss_backward_colsum: x, x_grad and the rows must be 16-byte aligned");
      const int64_t per_lane4 = ((dict_size / VEC + 3) / 4 + kWave - 1) / kWave;
      const bool wide = per_lane4 > 5;
      const int64_t per_lane = wide ? ((dict_size / VEC + 7) / 8 + kWave - 1) / kWave : per_lane4;
      CAIMAN_CHECK(per_lane <= 16, "transducer_loss_backward_colsum: dict_size %lld too large for the fused column sums",
                   (long long)dict_size);
      // descriptors behind the partial sums: [nblk, V] floats (rounded up to a multiple of 8), then total_rows * 32 bytes
      const int64_t lead = (nblk * dict_size + 7) / 8 * 8;
      auto* desc = reinterpret_cast<RowDesc<A>*>(colsum_partial + lead);
      CAIMAN_CHECK(reinterpret_cast<uintptr_t>(desc) % 32 == 0, "transducer_loss_backward_colsum: workspace must be 32-byte aligned");
      hipLaunchKernelGGL((loss_row_desc_kernel<scalar_t>), dim3((unsigned)((total_rows + 255) / 256)), dim3(256), 0, s,
                         (AA)denom, (AA)loss_grad, (AA)alpha, (AA)beta, p, total_rows, desc);
      auto go = [&](auto k_tag, auto w_tag) {
        constexpr int K_ = decltype(k_tag)::value, W_ = decltype(w_tag)::value;
        hipLaunchKernelGGL((loss_bwd_colsum_kernel<scalar_t, VEC, K_, W_>), grid, dim3(64 * W_), 0, s, xs, desc, dict_size,
                           blank_idx, total_rows, gs, (int)rows_per_block, colsum_partial);
      };
      using w4 = std::integral_constant<int, 4>;
      using w8 = std::integral_constant<int, 8>;
      if (!wide) go(std::integral_constant<int, 5>{}, w4{});
      else if (per_lane <= 5) go(std::integral_constant<int, 5>{}, w8{});
      else if (per_lane <= 9) go(std::integral_constant<int, 9>{}, w8{});
      else go(std::integral_constant<int, 16>{}, w8{});
    } else if (aligned) {
      hipLaunchKernelGGL((loss_bwd_kernel<scalar_t, VEC>), grid, dim3(256), 0, s, xs, (AA)denom, (AA)loss_grad, (AA)alpha,
                         (AA)beta, p, total_rows, gs);
    } else {
      hipLaunchKernelGGL((loss_bwd_kernel<scalar_t, 1>), grid, dim3(256), 0, s, xs, (AA)denom, (AA)loss_grad, (AA)alpha,
                         (AA)beta, p, total_rows, gs);
    }
    return check_launch("caiman_transducer_loss_backward");
  });
}
}  // namespace
}  // namespace caiman

extern "C" int caiman_transducer_loss_backward(
    const void* x, const void* denom, const void* loss_grad, const void* alpha, const void* beta,
    const int32_t* f_len, const int32_t* y_len, const int32_t* label, const int64_t* batch_offset,
    int64_t batch, int64_t max_f_len, int64_t max_g_len, int64_t dict_size, int64_t total_rows,
    double dp_lam, int64_t blank_idx, double eos_lam, int64_t eos_idx, double star_lam, int64_t star_idx,
    int packed, int dtype, void* x_grad, caiman_stream_t stream) {
  return caiman::loss_backward_impl(x, denom, loss_grad, alpha, beta, f_len, y_len, label, batch_offset, batch, max_f_len,
                                    max_g_len, dict_size, total_rows, dp_lam, blank_idx, eos_lam, eos_idx, star_lam,
                                    star_idx, packed, dtype, x_grad, nullptr, 4, stream);
}

extern "C" int caiman_transducer_loss_backward_colsum(
    const void* x, const void* denom, const void* loss_grad, const void* alpha, const void* beta,
    const int32_t* f_len, const int32_t* y_len, const int32_t* label, const int64_t* batch_offset,
    int64_t batch, int64_t max_f_len, int64_t max_g_len, int64_t dict_size, int64_t total_rows,
    double dp_lam, int64_t blank_idx, double eos_lam, int64_t eos_idx, double star_lam, int64_t star_idx,
    int packed, int dtype, void* x_grad, float* colsum_partial, int64_t rows_per_block, caiman_stream_t stream) {
  CAIMAN_CHECK(colsum_partial, "transducer_loss_backward_colsum: null colsum_partial");
  return caiman::loss_backward_impl(x, denom, loss_grad, alpha, beta, f_len, y_len, label, batch_offset, batch, max_f_len,
                                    max_g_len, dict_size, total_rows, dp_lam, blank_idx, eos_lam, eos_idx, star_lam,
                                    star_idx, packed, dtype, x_grad, colsum_partial, rows_per_block, stream);
}
