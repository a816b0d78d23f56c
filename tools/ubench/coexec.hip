// Do matrix-pipe and vector-ALU instructions of two waves on ONE SIMD overlap on gfx950, by MFMA type?
//   hipcc --offload-arch=gfx950 -O3 -o coexec coexec.hip && ./coexec
// One 512-thread workgroup per CU (LDS-limited): waves w and w + 4 share a SIMD.  Waves 0..3 run NM MFMAs of the chosen
// type (four independent accumulators), waves 4..7 run NV VALU instructions (v_fma_f32 or v_pk_fma_f32, 16 independent
// chains).  Times: MFMA role alone, VALU role alone, both.  both == max(alone) -> separate pipes; both == sum -> shared.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MT, int VT, int PM = 0, int PV = 0, bool SWAP = false>   // PM / PV: s_setprio of the MFMA / VALU waves; SWAP: MFMA role in the younger waves 4..7
// MT: 0 none, 1 f32 16x16x4, 2 bf16 16x16x32, 3 i8 16x16x64, 4 f16 16x16x32, 5 f32 32x32x2, 6 bf16 16x16x16, 7 i8 16x16x32 (64-bit operands), 8 fp8 16x16x32 ; VT: 0 none, 1 v_fma_f32, 2 v_pk_fma_f32
__global__ __launch_bounds__(512) void k(float *out, int nm, int nv, float a, float b) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  const bool mrole = SWAP ? wave >= 4 : wave < 4;
  if (mrole) __builtin_amdgcn_s_setprio(PM); else __builtin_amdgcn_s_setprio(PV);
  if (mrole) {
    if constexpr (MT == 1) {
      f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
      const float x = threadIdx.x * 1e-3f, y = a;
      for (int i = 0; i < nm; i += 4) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c3, 0, 0, 0);
      }
      r = c0[0] + c1[1] + c2[2] + c3[3];
    } else if constexpr (MT == 5) {
      typedef float f16v __attribute__((ext_vector_type(16)));
      f16v c0 = {}, c1 = {};
      const float x = threadIdx.x * 1e-3f, y = a;
      for (int i = 0; i < nm; i += 2) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c1, 0, 0, 0);
      }
      r = c0[0] + c1[1];
    } else if constexpr (MT == 2) {
      f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
      b8 x, y;
      for (int j = 0; j < 8; j++) { x[j] = (__bf16)(threadIdx.x * 1e-3f + j); y[j] = (__bf16)(a + j); }
      for (int i = 0; i < nm; i += 4) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c3, 0, 0, 0);
      }
      r = c0[0] + c1[1] + c2[2] + c3[3];
    } else if constexpr (MT == 4) {
      f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
      h8 x, y;
      for (int j = 0; j < 8; j++) { x[j] = (_Float16)(threadIdx.x * 1e-3f + j); y[j] = (_Float16)(a + j); }
      for (int i = 0; i < nm; i += 4) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c3, 0, 0, 0);
      }
      r = c0[0] + c1[1] + c2[2] + c3[3];
    } else if constexpr (MT == 6) {
      typedef short s4v __attribute__((ext_vector_type(4)));
      f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
      s4v x = {(short)threadIdx.x, 3, 5, 7}, y = {(short)(a * 100), 1, 2, 3};
      for (int i = 0; i < nm; i += 4) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, c3, 0, 0, 0);
      }
      r = c0[0] + c1[1] + c2[2] + c3[3];
    } else if constexpr (MT == 7 || MT == 8) {      // 64-bit operands: the older int8 16x16x32 and fp8 16x16x32
      i4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
      f4 e0 = {0, 0, 0, 0}, e1 = e0, e2 = e0, e3 = e0;
      const long x = ((long)threadIdx.x << 32) | 0x03050709, y = ((long)(int)(a * 100) << 32) | 0x01020304;
      for (int i = 0; i < nm; i += 4) {
        if constexpr (MT == 7) {
          c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(x, y, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(x, y, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(x, y, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(x, y, c3, 0, 0, 0);
        } else {
          e0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(x, y, e0, 0, 0, 0); e1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(x, y, e1, 0, 0, 0);
          e2 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(x, y, e2, 0, 0, 0); e3 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(x, y, e3, 0, 0, 0);
        }
      }
      r = (float)(c0[0] + c1[1] + c2[2] + c3[3]) + e0[0] + e1[1] + e2[2] + e3[3];
    } else if constexpr (MT == 3) {
      i4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
      i4 x = {(int)threadIdx.x, 3, 5, 7}, y = {(int)(a * 100), 1, 2, 3};
      for (int i = 0; i < nm; i += 4) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, c3, 0, 0, 0);
      }
      r = (float)(c0[0] + c1[1] + c2[2] + c3[3]);
    }
  } else {
    if constexpr (VT == 1) {
      float acc[16];
      for (int i = 0; i < 16; i++) acc[i] = threadIdx.x + i;
      for (int it = 0; it < nv; it += 16) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = __builtin_fmaf(acc[i], a, b);
      }
      for (int i = 0; i < 16; i++) r += acc[i];
    } else if constexpr (VT == 2) {
      f2 acc[16];
      for (int i = 0; i < 16; i++) acc[i] = f2{(float)threadIdx.x + i, (float)i};
      const f2 av = {a, a}, bv = {b, b};
      for (int it = 0; it < nv; it += 16) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
      }
      for (int i = 0; i < 16; i++) r += acc[i].x + acc[i].y;
    }
  }
  if (r == 12345.678f) lds[threadIdx.x] = r;   // keep the LDS allocation and the result alive
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MT, int VT, int PM = 0, int PV = 0, bool SWAP = false>
float run(float *d, int nm, int nv) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute((const void *)k<MT, VT, PM, PV, SWAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k<MT, VT, PM, PV, SWAP>), dim3(256), dim3(512), 100 * 1024, 0, d, nm, nv, 0.999f, 0.001f);
  CHECK(hipEventRecord(e0));
  for (int w = 0; w < 5; w++) hipLaunchKernelGGL((k<MT, VT, PM, PV, SWAP>), dim3(256), dim3(512), 100 * 1024, 0, d, nm, nv, 0.999f, 0.001f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5;
}

template <int MT>
void pair(const char *name, float *d, int nm, double mfma_cyc_expected) {
  const int nv = 1 << 20;
  const float tm = run<MT, 0>(d, nm, 0);
  const float tv1 = run<0, 1>(d, 0, nv), tv2 = run<0, 2>(d, 0, nv);
  const float b1 = run<MT, 1>(d, nm, nv), b2 = run<MT, 2>(d, nm, nv);
  printf("%-14s MFMA alone %.3f ms (%d instr, ~%.1f cyc each at 2.4 GHz) | v_fma alone %.3f, both %.3f (sum %.3f) | v_pk_fma alone %.3f, both %.3f (sum %.3f)\n",
         name, tm, nm, tm * 2.4e6 / nm, tv1, b1, tm + tv1, tv2, b2, tm + tv2);
  (void)mfma_cyc_expected;
  printf("   %-11s v_fma + MFMA: VALU waves at prio 3: %.3f | MFMA waves at prio 3: %.3f | MFMA in the younger waves: %.3f, ... and VALU prio 3: %.3f, ... and MFMA prio 3: %.3f\n", name,
         run<MT, 1, 0, 3>(d, nm, nv), run<MT, 1, 3, 0>(d, nm, nv), run<MT, 1, 0, 0, true>(d, nm, nv), run<MT, 1, 0, 3, true>(d, nm, nv), run<MT, 1, 3, 0, true>(d, nm, nv));
  printf("   %-11s v_pk_fma + MFMA: VALU waves at prio 3: %.3f | MFMA waves at prio 3: %.3f | MFMA in the younger waves: %.3f\n", name,
         run<MT, 2, 0, 3>(d, nm, nv), run<MT, 2, 3, 0>(d, nm, nv), run<MT, 2, 0, 0, true>(d, nm, nv));
}

int main() {
  float *d; CHECK(hipMalloc(&d, 256 * 512 * 4));
  pair<1>("f32 16x16x4", d, 1 << 16, 32);
  pair<5>("f32 32x32x2", d, 1 << 15, 64);
  pair<2>("bf16 16x16x32", d, 1 << 17, 16);
  pair<6>("bf16 16x16x16", d, 1 << 17, 8);
  pair<4>("f16 16x16x32", d, 1 << 17, 16);
  pair<3>("i8 16x16x64", d, 1 << 17, 16);
  pair<7>("i8 16x16x32", d, 1 << 17, 16);
  pair<8>("fp8 16x16x32", d, 1 << 17, 16);
  return 0;
}
