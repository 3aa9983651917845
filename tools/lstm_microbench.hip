// Ablation micro-benchmark for the per-timestep LSTM forward kernel (not part of the product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lstm_microbench.hip -o gpurun_out/lstm_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using bf16 = __bf16;
using frag = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE bits: 1 = load R, 2 = load h, 4 = epilogue loads/stores, 8 = nontemporal R loads
template <int MODE, int NK>
__global__ __launch_bounds__(256) void fwd_step(const bf16* __restrict__ R, bf16* __restrict__ g,
                                                const bf16* __restrict__ c0, bf16* __restrict__ c1,
                                                const bf16* __restrict__ y0, bf16* __restrict__ y1, int B, int H) {
  __shared__ float tile[4][2][16][17];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;
  const int j0 = blockIdx.x * 4, m0 = blockIdx.y * 32;
  const int eb = tid >> 2, eu = tid & 3;
  const int be = m0 + eb, ne = j0 + eu;
  const bool ep = (tid < 128) && (be < B) && (MODE & 4);
  const int64_t gb = (int64_t)be * 4 * H + ne;
  float pre[4] = {0.f, 0.f, 0.f, 0.f};
  float cprev = 0.f;
  if (ep) {
#pragma unroll
    for (int q = 0; q < 4; ++q) pre[q] = (float)g[gb + (int64_t)q * H];
    cprev = (float)c0[(int64_t)be * H + ne];
  }
  const int nk_total = H >> 5;
  const bf16* Rrow = R + (int64_t)((r >> 2) * H + j0 + (r & 3)) * H + 8 * kq;
  const bf16* A0 = y0 + (int64_t)(m0 + r) * H + 8 * kq;
  const bf16* A1 = y0 + (int64_t)(m0 + 16 + r) * H + 8 * kq;
  frag zero;
#pragma unroll
  for (int q = 0; q < 8; ++q) zero[q] = (bf16)0.f;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  frag bf[NK], a0[NK], a1[NK];
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    const int s = (MODE & 32) ? (wave * NK + i) : (MODE & 16) ? (2 * (wave + 4 * (i >> 1)) + (i & 1)) : wave + 4 * i;
    if (MODE & 128) bf[i] = *reinterpret_cast<const frag*>(R + (((int64_t)blockIdx.x * nk_total + s) * 16 + r) * 32 + 8 * kq);
    else if (MODE & 8) bf[i] = (MODE & 1) ? __builtin_nontemporal_load(reinterpret_cast<const frag*>(Rrow + 32 * s)) : zero;
    else bf[i] = (MODE & 1) ? *reinterpret_cast<const frag*>(Rrow + 32 * s) : zero;
    if (MODE & 64) {
      const bf16* T0 = y0 + ((int64_t)s * 32 + r) * 32 + 8 * kq;
      a0[i] = *reinterpret_cast<const frag*>(T0);
      a1[i] = *reinterpret_cast<const frag*>(T0 + 16 * 32);
    } else {
      a0[i] = (MODE & 2) ? *reinterpret_cast<const frag*>(A0 + 32 * s) : zero;
      a1[i] = (MODE & 2) ? *reinterpret_cast<const frag*>(A1 + 32 * s) : zero;
    }
  }
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], bf[i], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], bf[i], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    tile[wave][0][kq * 4 + q][r] = acc0[q];
    tile[wave][1][kq * 4 + q][r] = acc1[q];
  }
  __syncthreads();
  if (!(MODE & 4)) { if (tid == 0 && tile[0][0][0][0] == 123.f) y1[0] = (bf16)1.f; return; }
  if (!ep) return;
  const int mt = eb >> 4, rr = eb & 15;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = q * 4 + eu;
    pre[q] += tile[0][mt][rr][col] + tile[1][mt][rr][col] + tile[2][mt][rr][col] + tile[3][mt][rr][col];
  }
  const float i = 1.f / (1.f + __expf(-pre[0])), f = 1.f / (1.f + __expf(-pre[1]));
  const float gg = tanhf(pre[2]), o = 1.f / (1.f + __expf(-pre[3]));
  const float c = i * gg + f * cprev;
  g[gb] = (bf16)i; g[gb + H] = (bf16)f; g[gb + 2 * (int64_t)H] = (bf16)gg; g[gb + 3 * (int64_t)H] = (bf16)o;
  c1[(int64_t)be * H + ne] = (bf16)c;
  y1[(int64_t)be * H + ne] = (bf16)(o * tanhf(c));
}

__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }

template <int MODE>
float run(const bf16* R, bf16* g, bf16* c, bf16* y, int T, int B, int H, int nstreams, hipStream_t* st) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int64_t go = (int64_t)B * 4 * H, so = (int64_t)B * H;
  auto launch = [&](int reps) {
    for (int rep = 0; rep < reps; ++rep)
      for (int t = 0; t < T; ++t)
        for (int s = 0; s < nstreams; ++s) {
          bf16* gs = g + (int64_t)s * T * go; bf16* cs = c + (int64_t)s * (T + 1) * so; bf16* ys = y + (int64_t)s * (T + 1) * so;
          hipLaunchKernelGGL((fwd_step<MODE, 8>), dim3(H / 4, (B + 31) / 32), dim3(256), 0, st[s], R + (int64_t)s * 4 * H * H, gs + go * t,
                             cs + so * t, cs + so * (t + 1), ys + so * t, ys + so * (t + 1), B, H);
        }
  };
  launch(1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, st[0]));
  launch(3);
  for (int s = 1; s < nstreams; ++s) { hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, st[s])); CK(hipStreamWaitEvent(st[0], ev, 0)); }
  CK(hipEventRecord(e1, st[0]));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / (3.f * T);
}

template <int MODE>
float run_graph(const bf16* R, bf16* g, bf16* c, bf16* y, int T, int B, int H, hipStream_t st) {
  const int64_t go = (int64_t)B * 4 * H, so = (int64_t)B * H;
  hipGraph_t graph; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int t = 0; t < T; ++t)
    hipLaunchKernelGGL((fwd_step<MODE, 8>), dim3(H / 4, (B + 31) / 32), dim3(256), 0, st, R, g + go * t, c + so * t, c + so * (t + 1),
                       y + so * t, y + so * (t + 1), B, H);
  CK(hipStreamEndCapture(st, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st));
  CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / (3.f * T);
}

__global__ void empty_kernel2(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
float run_graph_empty(int T, hipStream_t st) {
  hipGraph_t graph; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int t = 0; t < T; ++t) hipLaunchKernelGGL(empty_kernel2, dim3(256), dim3(256), 0, st, (float*)nullptr);
  CK(hipStreamEndCapture(st, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st));
  CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / (3.f * T);
}

int main(int argc, char** argv) {
  const int T = 400, B = argc > 1 ? atoi(argv[1]) : 32, H = 1024, NS = 4;
  bf16 *R, *g, *c, *y;
  CK(hipMalloc(&R, (size_t)NS * 4 * H * H * 2));
  CK(hipMalloc(&g, (size_t)NS * T * B * 4 * H * 2));
  CK(hipMalloc(&c, (size_t)NS * (T + 1) * B * H * 2));
  CK(hipMalloc(&y, (size_t)NS * (T + 1) * B * H * 2));
  CK(hipMemset(R, 0, (size_t)NS * 4 * H * H * 2)); CK(hipMemset(g, 0, (size_t)NS * T * B * 4 * H * 2));
  CK(hipMemset(c, 0, (size_t)NS * (T + 1) * B * H * 2)); CK(hipMemset(y, 0, (size_t)NS * (T + 1) * B * H * 2));
  hipStream_t st[NS];
  for (int i = 0; i < NS; ++i) CK(hipStreamCreate(&st[i]));
  // empty kernel boundary
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st[0], (float*)nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, st[0]));
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st[0], (float*)nullptr);
    CK(hipEventRecord(e1, st[0])); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty 256x256 kernel           : %.2f us/launch\n", ms * 1e3f / 2000);
  }
  printf("B=%d H=%d T=%d  (us per timestep)\n", B, H, T);
  printf("full (R+h+epilogue)            : %.2f\n", run<7>(R, g, c, y, T, B, H, 1, st));
  printf("no R loads                     : %.2f\n", run<6>(R, g, c, y, T, B, H, 1, st));
  printf("no h loads                     : %.2f\n", run<5>(R, g, c, y, T, B, H, 1, st));
  printf("no operand loads (epilogue only): %.2f\n", run<4>(R, g, c, y, T, B, H, 1, st));
  printf("operands only, no epilogue     : %.2f\n", run<3>(R, g, c, y, T, B, H, 1, st));
  printf("MFMA+LDS only (no global)      : %.2f\n", run<0>(R, g, c, y, T, B, H, 1, st));
  printf("full, TILED h layout           : %.2f\n", run<71>(R, g, c, y, T, B, H, 1, st));
  printf("full, TILED h and R            : %.2f\n", run<199>(R, g, c, y, T, B, H, 1, st));
  printf("TILED h and R, no epilogue     : %.2f\n", run<195>(R, g, c, y, T, B, H, 1, st));
  printf("full, paired k-steps           : %.2f\n", run<23>(R, g, c, y, T, B, H, 1, st));
  printf("full, contiguous k per wave    : %.2f\n", run<39>(R, g, c, y, T, B, H, 1, st));
  printf("GRAPH: empty kernels           : %.2f\n", run_graph_empty(T, st[0]));
  printf("GRAPH: full TILED h and R      : %.2f\n", run_graph<199>(R, g, c, y, T, B, H, st[0]));
  printf("GRAPH: TILED, no epilogue      : %.2f\n", run_graph<195>(R, g, c, y, T, B, H, st[0]));
  printf("GRAPH: MFMA+LDS only           : %.2f\n", run_graph<0>(R, g, c, y, T, B, H, st[0]));
  printf("full, nontemporal R            : %.2f\n", run<15>(R, g, c, y, T, B, H, 1, st));
  printf("full, 2 independent streams    : %.2f per step-pair\n", run<7>(R, g, c, y, T, B, H, 2, st));
  printf("full, 4 independent streams    : %.2f per step-quad\n", run<7>(R, g, c, y, T, B, H, 4, st));
  return 0;
}
