// Memory-bound glue of one beam expansion round (include/caiman_beam.h, part 1b).
//
// A round evaluates, for n pending hypotheses, one step of the prediction network (embedding -> L LSTM layers
// -> joint_pred) and the joint (relu(f + g) -> joint_fc) -- training/caiman_asr_train/rnnt/beam.py:564-612
// (`_batched_decode`, `_collate`) and rnnt/model.py:344-439.  With one timestep per call the LSTM is a plain
// [n, I+H] x [I+H, 4H] product, which belongs to the BLAS library; what is left are gathers, the cell
// non-linearity and the scatter of the new states, written here so that a round is ~11 launches instead of the
// ~60 the module-by-module path needs:
//   caiman_beam_gather_inputs : X0[i] = [ embed[y_i] (0 for start-of-sequence) | h_pool[0][slot_in_i] ]
//   caiman_beam_lstm_cell     : gates -> (c, h) written to the state pool at slot_out_i, and the next layer's
//                               input row [ h | h_pool[l+1][slot_in_i] ]
//   caiman_beam_joint_act     : A[i] = relu(f_rows[row_i] + g[i])
// States live in pools laid out [layers, slots, hidden]; pool row 0 is the all-zero start state and slot s of
// the search object is pool row s + 1.
#include "common.h"
#include "../../include/caiman_beam.h"

namespace caiman {
namespace {

constexpr int kRowThreads = 256;

template <typename T>
__global__ __launch_bounds__(kRowThreads) void gather_inputs_kernel(const T* __restrict__ embed, int64_t E,
                                                                   const T* __restrict__ h_pool, int64_t H,
                                                                   const int32_t* __restrict__ y,
                                                                   const int32_t* __restrict__ slot_in,
                                                                   T* __restrict__ X, int64_t ldx) {
  const int64_t i = blockIdx.x;
  const int32_t tok = y[i];
  const T* e = embed + (int64_t)(tok < 0 ? 0 : tok) * E;
  const T* h = h_pool + (int64_t)(slot_in[i] + 1) * H;
  T* x = X + i * ldx;
  for (int64_t j = threadIdx.x; j < E; j += kRowThreads) x[j] = tok < 0 ? (T)0.f : e[j];
  for (int64_t j = threadIdx.x; j < H; j += kRowThreads) x[E + j] = h[j];
}

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

// gates [n, 4H] in the order i, f, g, o (training/lib/csrc/lstm.cu:99-102)
template <typename T>
__global__ __launch_bounds__(kRowThreads) void lstm_cell_kernel(const T* __restrict__ gates, int64_t H,
                                                               float* __restrict__ c_pool_l, T* __restrict__ h_pool_l,
                                                               const T* __restrict__ h_pool_next,
                                                               const int32_t* __restrict__ slot_in,
                                                               const int32_t* __restrict__ slot_out,
                                                               T* __restrict__ X_next, int64_t ldx) {
  const int64_t i = blockIdx.x;
  const T* g4 = gates + i * 4 * H;
  const int64_t rin = (int64_t)(slot_in[i] + 1) * H, rout = (int64_t)(slot_out[i] + 1) * H;
  T* xn = X_next + i * ldx;
  for (int64_t j = threadIdx.x; j < H; j += kRowThreads) {
    const float gi = sigmoidf_((float)g4[j]), gf = sigmoidf_((float)g4[H + j]);
    const float gg = tanhf((float)g4[2 * H + j]), go = sigmoidf_((float)g4[3 * H + j]);
    const float c = gi * gg + gf * c_pool_l[rin + j];
    const T h = (T)(go * tanhf(c));
    c_pool_l[rout + j] = c;
    h_pool_l[rout + j] = h;
    xn[j] = h;
    if (h_pool_next) xn[H + j] = h_pool_next[rin + j];
  }
}

template <typename T>
__global__ __launch_bounds__(kRowThreads) void joint_act_kernel(const T* __restrict__ f_rows, const int64_t* __restrict__ row,
                                                               const T* __restrict__ g, int64_t Hj, T* __restrict__ A) {
  const int64_t i = blockIdx.x;
  const T* f = f_rows + row[i] * Hj;
  for (int64_t j = threadIdx.x; j < Hj; j += kRowThreads) {
    const float v = (float)f[j] + (float)g[i * Hj + j];
    A[i * Hj + j] = (T)(v > 0.f ? v : 0.f);
  }
}

}  // namespace
}  // namespace caiman

extern "C" int caiman_beam_gather_inputs(const void* embed, int64_t embed_dim, const void* h_pool_l0, int64_t hidden,
                                         const int32_t* y_last, const int32_t* slot_in, int64_t n, void* X, int64_t ldx,
                                         int dtype, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(n >= 0 && embed_dim >= 1 && hidden >= 1 && ldx >= embed_dim + hidden && n <= 0x7fffffffLL,
               "beam_gather_inputs: bad extents");
  if (n == 0) return CAIMAN_OK;
  CAIMAN_CHECK(embed && h_pool_l0 && y_last && slot_in && X, "beam_gather_inputs: null pointer");
  return CAIMAN_DISPATCH(dtype, "beam_gather_inputs", [&]() -> int {
    hipLaunchKernelGGL((gather_inputs_kernel<scalar_t>), dim3((unsigned)n), dim3(kRowThreads), 0,
                       static_cast<hipStream_t>(stream), static_cast<const scalar_t*>(embed), embed_dim,
                       static_cast<const scalar_t*>(h_pool_l0), hidden, y_last, slot_in, static_cast<scalar_t*>(X), ldx);
    return check_launch("caiman_beam_gather_inputs");
  });
}

extern "C" int caiman_beam_lstm_cell(const void* gates, int64_t hidden, float* c_pool_l, void* h_pool_l,
                                     const void* h_pool_next, const int32_t* slot_in, const int32_t* slot_out, int64_t n,
                                     void* X_next, int64_t ldx, int dtype, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(n >= 0 && hidden >= 1 && ldx >= (h_pool_next ? 2 * hidden : hidden) && n <= 0x7fffffffLL,
               "beam_lstm_cell: bad extents");
  if (n == 0) return CAIMAN_OK;
  CAIMAN_CHECK(gates && c_pool_l && h_pool_l && slot_in && slot_out && X_next, "beam_lstm_cell: null pointer");
  return CAIMAN_DISPATCH(dtype, "beam_lstm_cell", [&]() -> int {
    hipLaunchKernelGGL((lstm_cell_kernel<scalar_t>), dim3((unsigned)n), dim3(kRowThreads), 0,
                       static_cast<hipStream_t>(stream), static_cast<const scalar_t*>(gates), hidden, c_pool_l,
                       static_cast<scalar_t*>(h_pool_l), static_cast<const scalar_t*>(h_pool_next), slot_in, slot_out,
                       static_cast<scalar_t*>(X_next), ldx);
    return check_launch("caiman_beam_lstm_cell");
  });
}

extern "C" int caiman_beam_joint_act(const void* f_rows, const int64_t* row, const void* g, int64_t n, int64_t joint_dim,
                                     void* A, int dtype, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(n >= 0 && joint_dim >= 1 && n <= 0x7fffffffLL, "beam_joint_act: bad extents");
  if (n == 0) return CAIMAN_OK;
  CAIMAN_CHECK(f_rows && row && g && A, "beam_joint_act: null pointer");
  return CAIMAN_DISPATCH(dtype, "beam_joint_act", [&]() -> int {
    hipLaunchKernelGGL((joint_act_kernel<scalar_t>), dim3((unsigned)n), dim3(kRowThreads), 0,
                       static_cast<hipStream_t>(stream), static_cast<const scalar_t*>(f_rows), row,
                       static_cast<const scalar_t*>(g), joint_dim, static_cast<scalar_t*>(A));
    return check_launch("caiman_beam_joint_act");
  });
}
