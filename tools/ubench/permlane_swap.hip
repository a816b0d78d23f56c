// What v_permlane32_swap_b32 / v_permlane16_swap_b32 (gfx950) do, printed: hipcc --offload-arch=gfx950 -O2 -o permlane_swap permlane_swap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *out) {
  const unsigned lane = threadIdx.x;
  unsigned a = 1000 + lane, b = 2000 + lane;
  u2v r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane] = r.x; out[64 + lane] = r.y;
  u2v q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + lane] = q.x; out[192 + lane] = q.y;
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[4] = {"permlane32_swap(a, b).x", "permlane32_swap(a, b).y", "permlane16_swap(a, b).x", "permlane16_swap(a, b).y"};
  for (int v = 0; v < 4; v++) {
    printf("%s: lanes 0,15,16,31,32,47,48,63 ->", names[v]);
    const int ls[8] = {0, 15, 16, 31, 32, 47, 48, 63};
    for (int i = 0; i < 8; i++) printf(" %u", h[64 * v + ls[i]]);
    printf("\n");
  }
  return 0;
}
