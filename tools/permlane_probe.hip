// Prints what v_permlane32_swap / v_permlane16_swap do to two registers holding (register id, lane) on gfx950.
// hipcc --offload-arch=gfx950 -O2 tools/permlane_probe.hip -o tools/permlane_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  unsigned a = 0x000 + l, b = 0x100 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"swap32 first ", "swap32 second", "swap16 first ", "swap16 second"};
  for (int i = 0; i < 4; ++i) {
    printf("%s:", names[i]);
    for (int g = 0; g < 4; ++g) printf("  lanes %2d-%2d <- reg %c lanes %2u..", g * 16, g * 16 + 15, (h[i * 64 + g * 16] >> 8) ? 'b' : 'a', h[i * 64 + g * 16] & 0xff);
    printf("\n");
  }
  return 0;
}
