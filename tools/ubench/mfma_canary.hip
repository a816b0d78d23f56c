// Do the registers of one wave survive while other waves of the same SIMD run bf16 MFMA bursts?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_canary mfma_canary.hip && ./mfma_canary
// Every wave alternates a phase of NB back-to-back v_mfma_f32_16x16x32_bf16 (or i8 16x16x64) with a VALU-only phase in
// which 48 "canary" registers - values derived from the lane id - are recomputed in place by an involution
// (x -> c - x twice) and finally compared with what they must be.  Waves start out of phase.  Mismatches are counted per
// (register, lane group).  mode 0: bf16 bursts, 1: i8 bursts, 2: no MFMA (control).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 3) void k(unsigned *bad, int iters, int nb) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float can[48];
#pragma unroll
  for (int i = 0; i < 48; i++) can[i] = (float)(lane * 64 + i);
  b8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane ^ j)); }
  i4 ai = {lane, 3, 5, 7}, bi = {wave, 1, 2, 3};
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  i4 d0 = {0, 0, 0, 0}, d1 = d0;
  unsigned errs = 0;
  // out of phase: wave w of block b idles first
  for (int s = 0; s < (int)((wave * 7 + blockIdx.x * 3) % 13) * 20; s++) __builtin_amdgcn_s_sleep(1);
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
      for (int m = 0; m < nb; m += 4) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
      }
    } else if (MODE == 1) {
      for (int m = 0; m < nb; m += 2) {
        d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ai, bi, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ai, bi, d1, 0, 0, 0);
      }
    }
    // VALU phase on the canaries: two involutions, kept in registers
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int i = 0; i < 48; i++) can[i] = 100000.0f - can[i];
#pragma unroll
      for (int i = 0; i < 48; i += 8) asm volatile("" : "+v"(can[i]), "+v"(can[i + 1]), "+v"(can[i + 2]), "+v"(can[i + 3]), "+v"(can[i + 4]), "+v"(can[i + 5]), "+v"(can[i + 6]), "+v"(can[i + 7]));
    }
#pragma unroll
    for (int i = 0; i < 48; i++)
      if (can[i] != (float)(lane * 64 + i)) { errs++; can[i] = (float)(lane * 64 + i); atomicAdd(&bad[1 + (i % 48) * 4 + (lane >> 4)], 1u); }
  }
  float sink = c0[0] + c1[1] + c2[2] + c3[3] + (float)(d0[0] + d1[1]);
  if (sink == 1234.5f) lds[threadIdx.x] = sink;
  if (errs) atomicAdd(&bad[0], errs);
}

template <int MODE>
void run(const char *name, unsigned *d_bad, int iters, int nb) {
  CHECK(hipMemset(d_bad, 0, 4 * 256));
  CHECK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 50 * 1024));
  hipLaunchKernelGGL((k<MODE>), dim3(768), dim3(256), 50 * 1024, 0, d_bad, iters, nb);   // 3 workgroups of 4 waves per CU: 3 waves per SIMD
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned> h(256);
  CHECK(hipMemcpy(h.data(), d_bad, 4 * 256, hipMemcpyDeviceToHost));
  printf("%-6s iters %d, %d MFMAs per burst: canary mismatches %u", name, iters, nb, h[0]);
  if (h[0]) {
    printf("  by lane group:");
    for (int g = 0; g < 4; g++) { unsigned s = 0; for (int i = 0; i < 48; i++) s += h[1 + i * 4 + g]; printf(" %u", s); }
  }
  printf("\n");
}

int main() {
  unsigned *d_bad; CHECK(hipMalloc(&d_bad, 4 * 256));
  run<2>("none", d_bad, 2000, 0);
  run<1>("i8", d_bad, 2000, 48);
  run<0>("bf16", d_bad, 2000, 48);
  run<0>("bf16", d_bad, 2000, 144);
  return 0;
}
