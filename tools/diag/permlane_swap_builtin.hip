// hipcc 7.2 / gfx950: the SECOND result of __builtin_amdgcn_permlane32_swap / _permlane16_swap is taken from the FIRST result's register.
// Self-checking: the builtin form against the same swaps spelled as asm statements.   hipcc --offload-arch=gfx950 -O3 -o permlane_swap_builtin permlane_swap_builtin.hip
// (device assembly of k_builtin: `v_permlane32_swap_b32 v1, v2` is followed by `v_add_f32 v2, v1, v1` and `v_add_f32 v1, 5.0, v1` - the
//  second addition should read v2.)  Found in round 5 while replacing stage A's LDS exchange by the two swaps (fmd_kernels.inc, decimate_mfma).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__global__ void k_builtin(float *p) {
  float a = p[threadIdx.x] * 3.0f, b = p[threadIdx.x + 64] + 1.0f;
  u2v r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  float c = __builtin_bit_cast(float, r.x) * 2.0f, d = __builtin_bit_cast(float, r.y) + 5.0f;
  u2v q = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d), false, false);
  p[threadIdx.x] = __builtin_bit_cast(float, q.x) * 7.0f; p[threadIdx.x + 64] = __builtin_bit_cast(float, q.y) - 2.0f;
}
__global__ void k_asm(float *p) {
  float a = p[threadIdx.x] * 3.0f, b = p[threadIdx.x + 64] + 1.0f;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  float c = a * 2.0f, d = b + 5.0f;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
  p[threadIdx.x] = c * 7.0f; p[threadIdx.x + 64] = d - 2.0f;
}
int main() {
  float h[128], g1[128], g2[128], *d;
  for (int i = 0; i < 128; i++) h[i] = (float)(i + 1);
  (void)hipMalloc(&d, sizeof(h));
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_builtin, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(g1, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_asm, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(g2, d, sizeof(h), hipMemcpyDeviceToHost);
  /* host model: permlane32_swap(a, b): x = [a.lo32, b.lo32], y = [a.hi32, b.hi32]; permlane16_swap: x rows = [a.r0, b.r0, a.r2, b.r2], y rows = [a.r1, b.r1, a.r3, b.r3] */
  float a[64], b[64], x[64], y[64], c[64], e[64], want[128];
  for (int l = 0; l < 64; l++) { a[l] = h[l] * 3.0f; b[l] = h[l + 64] + 1.0f; }
  for (int l = 0; l < 64; l++) { x[l] = l < 32 ? a[l] : b[l - 32]; y[l] = l < 32 ? a[l + 32] : b[l]; }
  for (int l = 0; l < 64; l++) { c[l] = x[l] * 2.0f; e[l] = y[l] + 5.0f; }
  for (int l = 0; l < 64; l++) {
    const int r = l >> 4, i = l & 15;
    const float qx = (r & 1) ? e[16 * (r - 1) + i] : c[l], qy = (r & 1) ? e[l] : c[16 * (r + 1) + i];
    want[l] = qx * 7.0f; want[l + 64] = qy - 2.0f;
  }
  int bad1 = 0, bad2 = 0;
  for (int i = 0; i < 128; i++) { bad1 += g1[i] != want[i]; bad2 += g2[i] != want[i]; }
  printf("builtin form: %d of 128 results wrong; asm form: %d of 128 wrong\n", bad1, bad2);
  return bad2 != 0;
}
