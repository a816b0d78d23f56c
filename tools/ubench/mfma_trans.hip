// Does v_rcp_f32 (quarter-rate "trans" VALU op) or a packed / DPP VALU op of one wave give wrong results while other waves
// of the same SIMD run dense 4-pass MFMAs?  Each wave alternates an MFMA burst with a VALU phase that folds the bits of
// rcp / packed-fma / DPP results of lane-dependent inputs into a checksum; the checksums of a run WITHOUT MFMAs are the
// reference.   hipcc --offload-arch=gfx950 -O3 -o mfma_trans mfma_trans.hip && ./mfma_trans
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0 bf16 16x16x32, 1 i8 16x16x64, 2 none
__global__ __launch_bounds__(256, 3) void k(unsigned *sum_out, int iters, int nb) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  b8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane ^ j)); }
  i4 ai = {lane, 3, 5, 7}, bi = {wave, 1, 2, 3};
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  i4 d0 = {0, 0, 0, 0}, d1 = d0;
  unsigned s_rcp = 0, s_pk = 0, s_dpp = 0;
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
    // VALU phase: 16 rcp, 16 packed fma with op_sel-free operands, 8 DPP row_shr adds, on lane- and iteration-dependent inputs
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = (float)(lane * 17 + i * 3 + 1) + 0.25f * (float)(it & 7);
#pragma unroll
    for (int i = 0; i < 16; i++) { const float r = __builtin_amdgcn_rcpf(x[i]); s_rcp = s_rcp * 31u + __builtin_bit_cast(unsigned, r); }
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const f2 p = __builtin_elementwise_fma(f2{x[i], x[i + 1]}, f2{1.5f, -0.75f}, f2{x[i + 1], x[i]});
      s_pk = s_pk * 31u + __builtin_bit_cast(unsigned, p.x) + 7u * __builtin_bit_cast(unsigned, p.y);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[i]), 0x111, 0xf, 0xf, true));   // row_shr:1
      s_dpp = s_dpp * 31u + __builtin_bit_cast(unsigned, t + x[i]);
    }
  }
  float sink = c0[0] + c1[1] + c2[2] + c3[3] + (float)(d0[0] + d1[1]);
  if (sink == 1234.5f) lds[threadIdx.x] = sink;
  const size_t gi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  sum_out[gi * 3] = s_rcp; sum_out[gi * 3 + 1] = s_pk; sum_out[gi * 3 + 2] = s_dpp;
}

template <int MODE>
std::vector<unsigned> run(unsigned *d, int iters, int nb, size_t n) {
  CHECK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 50 * 1024));
  hipLaunchKernelGGL((k<MODE>), dim3(768), dim3(256), 50 * 1024, 0, d, iters, nb);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned> h(n);
  CHECK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
  return h;
}

int main() {
  const size_t n = (size_t)768 * 256 * 3;
  unsigned *d; CHECK(hipMalloc(&d, n * 4));
  const int iters = 1500, nb = 48;
  auto ref = run<2>(d, iters, nb, n);
  const char *names[2] = {"bf16 16x16x32", "i8 16x16x64"};
  for (int mode = 0; mode < 2; mode++) {
    auto got = mode == 0 ? run<0>(d, iters, nb, n) : run<1>(d, iters, nb, n);
    size_t bad[3] = {0, 0, 0}, lanes48[3] = {0, 0, 0};
    for (size_t i = 0; i < n; i++)
      if (got[i] != ref[i]) { bad[i % 3]++; if (((i / 3) & 63) >= 48) lanes48[i % 3]++; }
    printf("%-14s threads whose checksum differs from the MFMA-free run: rcp %zu (lanes 48-63: %zu), packed fma %zu (%zu), dpp %zu (%zu) of %zu\n",
           names[mode], bad[0], lanes48[0], bad[1], lanes48[1], bad[2], lanes48[2], n / 3);
  }
  auto again = run<2>(d, iters, nb, n);
  size_t b2 = 0; for (size_t i = 0; i < n; i++) b2 += again[i] != ref[i];
  printf("control (MFMA-free run repeated): %zu differing checksums\n", b2);
  return 0;
}
