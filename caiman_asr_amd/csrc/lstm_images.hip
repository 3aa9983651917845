// 16-bit operand images of the LSTM parameters, all layers of a stack in ONE launch.
//
// The reference keeps fp32 master weights and lets autocast cast them inside every `torch.addmm` / kernel call
// (training/lib/src/rnnt_ext/custom_lstm/lstm.py:51-55, 76-140).  The layer pipeline here needs, per layer and training step:
//   Wt   [K, 4H]   input weights, K-major, gate columns in [unit][gate] order  (whole-sequence input GEMM, backward chunk GEMMs)
//   Wn   [4H, K]   input weights, rows in [unit][gate] order                   (forward operand of the grouped projection kernel)
//   bias [4H]      b_ih + b_hh in [unit][gate] order
//   Rf             recurrent weights in the forward kernels' fragment-major image  (what caiman_lstm_prepare(backward = 0) tiles)
//   Rb             recurrent weights in the backward kernels' image, interleaved gate layout (caiman_lstm_prepare(backward = 1))
// which the first version produced with 7 launches per layer (cast, permute + cast, transpose + cast, add, two tiling
// kernels): ~60 launches and 0.7 ms of a 29 ms training step for data that one pass over the fp32 parameters yields.
// One workgroup = one 32 x 32 tile of one matrix: 8 hidden units x 4 gates (the rows g*H + u of the reference layout)
// by 32 columns, read as fp32 in 128-byte row segments, held in LDS in [unit][gate] row order, written out in every
// image that wants it in 64-byte (Wn, Wt) or 1 KB (Rf, Rb) contiguous pieces.
#include "common.h"

namespace caiman {
namespace {

constexpr int kImgMax = CAIMAN_LSTM_IMAGES_MAX_LAYERS;
struct ImgBatch {
  caiman_lstm_images_t l[kImgMax];
  int job_begin[2 * kImgMax + 1];   // job ranges: [2i] = W tiles of layer i, [2i+1] = R tiles of layer i
  int n;
};

template <typename T>
__global__ __launch_bounds__(256) void lstm_images_kernel(ImgBatch ib) {
  __shared__ float t[32][33];   // [kk = unit_local * 4 + gate][column]
  const int bid = blockIdx.x, tid = threadIdx.x;
  int seg = 0;
#pragma unroll
  for (int i = 1; i < 2 * kImgMax; ++i)
    if (i < 2 * ib.n && bid >= ib.job_begin[i]) seg = i;
  const caiman_lstm_images_t& L = ib.l[seg >> 1];
  const bool is_r = seg & 1;
  const int H = L.H, K = is_r ? L.H : L.K;
  const float* __restrict__ src = is_r ? L.W_hh : L.W_ih;
  const int job = bid - ib.job_begin[seg];
  const int ncb = (K + 31) / 32;
  const int ub = job / ncb, cb = job - ub * ncb;   // 8-unit block, 32-column block
  const int u0 = ub * 8, c0 = cb * 32;

  {   // load: thread -> (reference row g*H + u0 + ul, 4 columns)
    const int rr = tid >> 3, c4 = (tid & 7) * 4;
    const int g = rr >> 3, ul = rr & 7;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 + c4 < K) v = *reinterpret_cast<const float4*>(src + (int64_t)(g * H + u0 + ul) * K + c0 + c4);   // K % 4 == 0
    const int kk = ul * 4 + g;
    t[kk][c4] = v.x; t[kk][c4 + 1] = v.y; t[kk][c4 + 2] = v.z; t[kk][c4 + 3] = v.w;
  }
  __syncthreads();
  using v4 = __attribute__((ext_vector_type(4))) T;
  const int a = tid >> 3, b4 = (tid & 7) * 4;   // 32 x (8 x 4) decomposition of the 32 x 32 outputs
  if (!is_r) {
    if (L.Wn && c0 + b4 < K) {   // row u0*4 + a of the [unit][gate] order, columns c0 + b4 ..
      v4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = static_cast<T>(t[a][b4 + j]);
      *reinterpret_cast<v4*>(static_cast<T*>(L.Wn) + (int64_t)(u0 * 4 + a) * K + c0 + b4) = o;
    }
    if (L.Wt && c0 + a < K) {    // row c0 + a of the transposed image, columns u0*4 + b4 ..
      v4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = static_cast<T>(t[b4 + j][a]);
      *reinterpret_cast<v4*>(static_cast<T*>(L.Wt) + (int64_t)(c0 + a) * (4 * H) + u0 * 4 + b4) = o;
    }
    if (cb == 0 && L.bias && tid < 32) {   // this unit block's 32 bias entries
      const int g = tid & 3, ul = tid >> 2;
      static_cast<T*>(L.bias)[u0 * 4 + tid] = static_cast<T>(L.b_ih[g * H + u0 + ul] + L.b_hh[g * H + u0 + ul]);
    }
  } else {
    const int half = tid >> 7, n = (tid >> 3) & 15;
    if (L.Rf) {
      // csrc/lstm.hip tile_R_fwd_kernel: Rf[((blk*nk + s)*16 + n)*32 + kk'] = R[(gate(n)*H + blk*4 + unit(n))*H + s*32 + kk'],
      // n = gate*4 + unit: two 4-unit blocks per tile, s = cb
      const int blk = u0 / 4 + half, g = n >> 2, ul = half * 4 + (n & 3);
      v4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = static_cast<T>(t[ul * 4 + g][b4 + j]);
      *reinterpret_cast<v4*>(static_cast<T*>(L.Rf) + (((int64_t)blk * (H >> 5) + cb) * 16 + n) * 32 + b4) = o;
    }
    if (L.Rb) {
      // tile_Rt_bwd_kernel<IL>: Rb[((blk*nk4 + s)*16 + n)*32 + kk] = R[row(s*32 + kk)*H + blk*16 + n], K index kk = unit*4 + gate:
      // s = ub (8 units x 4 gates = 32 K indices), two 16-column blocks per tile
      const int blk = c0 / 16 + half;
      v4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = static_cast<T>(t[b4 + j][half * 16 + n]);
      *reinterpret_cast<v4*>(static_cast<T*>(L.Rb) + (((int64_t)blk * ((4 * H) >> 5) + ub) * 16 + n) * 32 + b4) = o;
    }
  }
}

// The way back: a layer's parameter gradients leave the pipeline with rows in [unit][gate] order (16-bit GEMM outputs
// for the weights, fp32 row sums for the biases) and are ADDED into the fp32 `.grad` views of the optimiser's arena in the
// reference layout (rows [gate][unit]): un-permute + widen + accumulate for the four parameters of a layer in one launch
// (was one torch kernel per parameter: 50 launches, 0.35 ms per step).
constexpr int kDeliverMax = CAIMAN_LSTM_DELIVER_MAX_ITEMS;
struct DeliverBatch {
  caiman_lstm_grad_item_t it[kDeliverMax];
  int n;
};

template <typename T>
__global__ __launch_bounds__(256) void lstm_grad_deliver_kernel(DeliverBatch db) {
  const caiman_lstm_grad_item_t& I = db.it[blockIdx.y];
  const int H = I.H, cols = I.cols;
  float* __restrict__ dst = I.dst;
  if (cols % 4 == 0) {
    const int c4n = cols / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)4 * H * c4n) return;
    const int r = (int)(i / c4n), c = (int)(i - (int64_t)r * c4n) * 4;     // r = unit * 4 + gate
    const int64_t d = (int64_t)((r & 3) * H + (r >> 2)) * cols + c;
    float4 a = *reinterpret_cast<const float4*>(dst + d);
    if (I.src_fp32) {
      // partial products of the weight-gradient GEMM (slabs of 4H x cols floats, one behind the other): summed in order here
      const float* sp = static_cast<const float*>(I.src) + (int64_t)r * cols + c;
      float4 v = *reinterpret_cast<const float4*>(sp);
      for (int sl = 1; sl < I.slabs; ++sl) {
        const float4 w = *reinterpret_cast<const float4*>(sp + (int64_t)sl * 4 * H * cols);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    } else {
      using v4 = __attribute__((ext_vector_type(4))) T;
      const v4 v = *reinterpret_cast<const v4*>(static_cast<const T*>(I.src) + (int64_t)r * cols + c);
      a.x += static_cast<float>(v[0]); a.y += static_cast<float>(v[1]); a.z += static_cast<float>(v[2]); a.w += static_cast<float>(v[3]);
    }
    *reinterpret_cast<float4*>(dst + d) = a;
  } else {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)4 * H * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    const int64_t d = (int64_t)((r & 3) * H + (r >> 2)) * cols + c;
    float v = I.src_fp32 ? static_cast<const float*>(I.src)[i] : static_cast<float>(static_cast<const T*>(I.src)[i]);
    if (I.src_fp32)
      for (int sl = 1; sl < I.slabs; ++sl) v += static_cast<const float*>(I.src)[(int64_t)sl * 4 * H * cols + i];
    dst[d] += v;
  }
}

// h / c rows of each utterance's last valid step, every layer, one launch (rsp.py:108-130 there picks them with two
// advanced-indexing operations per stack).  Rows are moved as 4-byte words (row bytes % 4 == 0 is checked by the caller).
struct LastStates {
  const char* src[2];
  char* dst[2];
  int64_t stride_l[2], stride_t[2];   // bytes
  const void* lens;
  int lens_kind, back;
  int64_t T, B, row_bytes;
};
__global__ __launch_bounds__(256) void lstm_last_states_kernel(LastStates a) {
  const int b = blockIdx.x, l = blockIdx.y, which = blockIdx.z;
  int64_t t = (a.lens_kind == 0 ? (int64_t)static_cast<const int32_t*>(a.lens)[b] : static_cast<const int64_t*>(a.lens)[b]) - 1 - a.back;
  if (t < 0) t += a.T;                      // Python indexing: -1 is the last step
  t = t < 0 ? 0 : (t >= a.T ? a.T - 1 : t);   // out of range is an indexing error in torch; never read outside the tensor
  const uint32_t* s = reinterpret_cast<const uint32_t*>(a.src[which] + l * a.stride_l[which] + t * a.stride_t[which] + b * a.row_bytes);
  uint32_t* d = reinterpret_cast<uint32_t*>(a.dst[which] + ((int64_t)l * a.B + b) * a.row_bytes);
  for (int64_t i = threadIdx.x; i < a.row_bytes / 4; i += blockDim.x) d[i] = s[i];
}

}  // namespace
}  // namespace caiman

extern "C" int caiman_lstm_grad_deliver(const caiman_lstm_grad_item_t* items, int n_items, int dtype, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(items && n_items >= 1 && n_items <= kDeliverMax, "caiman_lstm_grad_deliver: 1 .. %d items per call", kDeliverMax);
  CAIMAN_CHECK(dtype == CAIMAN_F16 || dtype == CAIMAN_BF16, "caiman_lstm_grad_deliver: 16-bit sources are f16 / bf16");
  DeliverBatch db;
  db.n = n_items;
  int64_t most = 0;
  for (int i = 0; i < n_items; ++i) {
    const caiman_lstm_grad_item_t& I = items[i];
    CAIMAN_CHECK(I.src && I.dst && I.H >= 1 && I.cols >= 1, "caiman_lstm_grad_deliver: item %d: null pointer or empty extent", i);
    CAIMAN_CHECK(I.slabs >= 0 && I.slabs <= 64 && (I.slabs <= 1 || I.src_fp32), "caiman_lstm_grad_deliver: item %d: 0 .. 64 slabs, fp32 sources only", i);
    CAIMAN_CHECK(I.cols % 4 != 0 || (((reinterpret_cast<uintptr_t>(I.dst) & 15u) == 0) &&
                                     ((reinterpret_cast<uintptr_t>(I.src) & (I.src_fp32 ? 15u : 7u)) == 0)),
                 "caiman_lstm_grad_deliver: item %d: misaligned pointer", i);
    db.it[i] = I;
    const int64_t work = I.cols % 4 == 0 ? (int64_t)4 * I.H * (I.cols / 4) : (int64_t)4 * I.H * I.cols;
    most = work > most ? work : most;
  }
  for (int i = n_items; i < kDeliverMax; ++i) db.it[i] = items[0];
  const dim3 grid((unsigned)((most + 255) / 256), (unsigned)n_items);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == CAIMAN_BF16) hipLaunchKernelGGL((lstm_grad_deliver_kernel<bf16_t>), grid, dim3(256), 0, s, db);
  else hipLaunchKernelGGL((lstm_grad_deliver_kernel<f16_t>), grid, dim3(256), 0, s, db);
  return check_launch("lstm gradient delivery");
}

extern "C" int caiman_lstm_weight_images(const caiman_lstm_images_t* layers, int n_layers, int dtype, caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(layers && n_layers >= 1 && n_layers <= kImgMax, "caiman_lstm_weight_images: 1 .. %d layers per call", kImgMax);
  CAIMAN_CHECK(dtype == CAIMAN_F16 || dtype == CAIMAN_BF16, "caiman_lstm_weight_images: f16 / bf16 images only");
  ImgBatch ib;
  ib.n = n_layers;
  int jobs = 0;
  for (int i = 0; i < n_layers; ++i) {
    const caiman_lstm_images_t& L = layers[i];
    CAIMAN_CHECK(L.H >= 32 && L.H % 32 == 0 && L.K >= 4 && L.K % 4 == 0, "caiman_lstm_weight_images: layer %d: H %% 32 == 0, K %% 4 == 0", i);
    CAIMAN_CHECK(L.W_ih && L.W_hh, "caiman_lstm_weight_images: layer %d: null parameter", i);
    CAIMAN_CHECK(!L.bias || (L.b_ih && L.b_hh), "caiman_lstm_weight_images: layer %d: bias image without biases", i);
    CAIMAN_CHECK(((reinterpret_cast<uintptr_t>(L.W_ih) | reinterpret_cast<uintptr_t>(L.W_hh)) & 15u) == 0 &&
                 ((reinterpret_cast<uintptr_t>(L.Wt) | reinterpret_cast<uintptr_t>(L.Wn) | reinterpret_cast<uintptr_t>(L.Rf) |
                   reinterpret_cast<uintptr_t>(L.Rb)) & 7u) == 0, "caiman_lstm_weight_images: layer %d: misaligned pointer", i);
    ib.l[i] = L;
    ib.job_begin[2 * i] = jobs;
    jobs += (L.Wt || L.Wn || L.bias) ? (L.H / 8) * ((L.K + 31) / 32) : 0;
    ib.job_begin[2 * i + 1] = jobs;
    jobs += (L.Rf || L.Rb) ? (L.H / 8) * (L.H / 32) : 0;
  }
  for (int i = 2 * n_layers; i <= 2 * kImgMax; ++i) ib.job_begin[i] = jobs;
  for (int i = n_layers; i < kImgMax; ++i) ib.l[i] = layers[0];
  if (jobs == 0) return CAIMAN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == CAIMAN_BF16) hipLaunchKernelGGL((lstm_images_kernel<bf16_t>), dim3((unsigned)jobs), dim3(256), 0, s, ib);
  else hipLaunchKernelGGL((lstm_images_kernel<f16_t>), dim3((unsigned)jobs), dim3(256), 0, s, ib);
  return check_launch("lstm weight images");
}

extern "C" int caiman_lstm_last_states(const void* h, const void* c, int64_t L, int64_t T, int64_t B, int64_t row_bytes,
                                       int64_t h_stride_l, int64_t h_stride_t, int64_t c_stride_l, int64_t c_stride_t,
                                       const void* lens, int lens_kind, int back, void* h_out, void* c_out,
                                       caiman_stream_t stream) {
  using namespace caiman;
  CAIMAN_CHECK(h && c && h_out && c_out && lens, "caiman_lstm_last_states: null pointer");
  CAIMAN_CHECK(L >= 1 && L <= 65535 && T >= 1 && B >= 1 && row_bytes >= 4 && row_bytes % 4 == 0, "caiman_lstm_last_states: bad extents");
  CAIMAN_CHECK(lens_kind == 0 || lens_kind == 1, "caiman_lstm_last_states: lens_kind 0 (int32) or 1 (int64)");
  CAIMAN_CHECK(back >= 0, "caiman_lstm_last_states: negative look-back");
  CAIMAN_CHECK(((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(h_out) |
                 reinterpret_cast<uintptr_t>(c_out) | (uintptr_t)h_stride_l | (uintptr_t)h_stride_t | (uintptr_t)c_stride_l |
                 (uintptr_t)c_stride_t) & 3u) == 0, "caiman_lstm_last_states: pointers and strides must be multiples of 4 bytes");
  LastStates a;
  a.src[0] = static_cast<const char*>(h); a.src[1] = static_cast<const char*>(c);
  a.dst[0] = static_cast<char*>(h_out); a.dst[1] = static_cast<char*>(c_out);
  a.stride_l[0] = h_stride_l; a.stride_t[0] = h_stride_t; a.stride_l[1] = c_stride_l; a.stride_t[1] = c_stride_t;
  a.lens = lens; a.lens_kind = lens_kind; a.back = back; a.T = T; a.B = B; a.row_bytes = row_bytes;
  hipLaunchKernelGGL(lstm_last_states_kernel, dim3((unsigned)B, (unsigned)L, 2u), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch("lstm last states");
}
