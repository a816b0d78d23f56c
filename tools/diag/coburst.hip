// A neighbour that only issues matrix instructions: one small wave per SIMD (no LDS, < 56 VGPRs), started on its own
// stream BEFORE the product's launches so that it shares every SIMD with the fused kernel's three workers.  Used by
// tools/diag/coburst.py to answer: does a dense MFMA burst of ANOTHER wave change the results of the (VALU + sparse
// i8-MFMA) product kernels?  (The bf16 / int8 stage-C experiments of round 3 lost determinism exactly when several
// waves of a SIMD ran such bursts.)
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libcoburst.so coburst.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i16v __attribute__((ext_vector_type(16)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// kind 0: v_mfma_f32_16x16x32_bf16   1: v_mfma_i32_16x16x64_i8   2: v_mfma_f32_16x16x4_f32   3: v_fma_f32 only (control)
//      4: v_mfma_f32_32x32x16_bf16   5: v_mfma_i32_32x32x32_i8   6: v_mfma_f32_16x16x16_bf16 (64-bit operands)
//      7: v_mfma_f32_16x16x32_f16    8: v_mfma_f32_16x16x32_fp8_fp8 (64-bit operands)   9: kind 0 with s_nop 7 between the MFMAs (half duty)
//      10: v_mfma_i32_16x16x32_i8 (the older int8 opcode, 64-bit operands)
template <int KIND>
__global__ void __launch_bounds__(256) burst(volatile int *stop, float *sink, int prio, int max_loops) {
  const int lane = threadIdx.x & 63;
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  if (prio == 3) __builtin_amdgcn_s_setprio(3);
  f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
  i4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
  bf8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  i4 ia = {lane, lane * 3, lane * 5, lane * 7}, ib = {lane * 11, lane * 13, lane * 17, lane * 19};
  float fa = 0.001f * lane, fb = 0.5f;
  f16v e0 = {}, e1 = {};
  i16v g0 = {}, g1 = {};
  h8 ha, hb;
  for (int i = 0; i < 8; i++) { ha[i] = (_Float16)(0.001f * (lane + i)); hb[i] = (_Float16)(0.002f * (lane - i)); }
  for (int loop = 0; loop < max_loops; loop++) {        /* bounded: about 10 us per loop, so 500 000 loops end by themselves after seconds */
    for (int it = 0; it < 256; it++) {
      if constexpr (KIND == 0) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
      } else if constexpr (KIND == 1) {
        d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, d3, 0, 0, 0);
      } else if constexpr (KIND == 2) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c3, 0, 0, 0);
      } else if constexpr (KIND == 4) {
        e0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, e0, 0, 0, 0);
        e1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, e1, 0, 0, 0);
      } else if constexpr (KIND == 5) {
        g0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, g0, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, g1, 0, 0, 0);
      } else if constexpr (KIND == 6) {
        const s4 sa = {(short)ia.x, (short)ia.y, (short)ia.z, (short)ia.w}, sb = {(short)ib.x, (short)ib.y, (short)ib.z, (short)ib.w};
        c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, c3, 0, 0, 0);
      } else if constexpr (KIND == 7) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c3, 0, 0, 0);
      } else if constexpr (KIND == 8) {
        const long la = ((long)ia.x << 32) | (unsigned)ia.y, lb = ((long)ib.x << 32) | (unsigned)ib.y;
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, c3, 0, 0, 0);
      } else if constexpr (KIND == 10) {
        const long la = ((long)ia.x << 32) | (unsigned)ia.y, lb = ((long)ib.x << 32) | (unsigned)ib.y;
        d0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, d3, 0, 0, 0);
      } else if constexpr (KIND == 9) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1));
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0); asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1));
      } else {
        c0 = __builtin_elementwise_fma(c0, f4{fb, fb, fb, fb}, f4{fa, fa, fa, fa});
        c1 = __builtin_elementwise_fma(c1, f4{fb, fb, fb, fb}, f4{fa, fa, fa, fa});
      }
    }
    if (*stop) break;
  }
  const f4 c = c0 + c1 + c2 + c3;
  const i4 d = d0 + d1 + d2 + d3;
  float es = 0.f; int gs = 0;
  for (int i = 0; i < 16; i++) { es += e0[i] + e1[i]; gs += g0[i] + g1[i]; }
  sink[blockIdx.x * 256 + threadIdx.x] = c.x + c.y + c.z + c.w + (float)(d.x + d.y + d.z + d.w) + es + (float)gs;
}

static hipStream_t g_stream;
static int *g_stop;          // pinned host memory, read by the kernel through the zero-copy mapping
static float *g_sink;
static int g_blocks;

extern "C" int coburst_start(int kind, int blocks, int prio) {
  if (!g_stop) {
    if (hipHostMalloc((void **)&g_stop, 64, hipHostMallocMapped) != hipSuccess) return -1;
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return -2;
  }
  if (g_sink && blocks > g_blocks) { (void)hipFree(g_sink); g_sink = nullptr; }
  if (!g_sink) { if (hipMalloc((void **)&g_sink, (size_t)blocks * 256 * sizeof(float)) != hipSuccess) return -3; g_blocks = blocks; }
  *g_stop = 0;
  int *dstop = nullptr;
  if (hipHostGetDevicePointer((void **)&dstop, g_stop, 0) != hipSuccess) return -4;
  switch (kind) {
    case 0: hipLaunchKernelGGL(burst<0>, dim3(blocks), dim3(256), 0, g_stream, dstop, g_sink, prio, 500000); break;
    case 1: hipLaunchKernelGGL(burst<1>, dim3(blocks), dim3(256), 0, g_stream, dstop, g_sink, prio, 500000); break;
    case 2: hipLaunchKernelGGL(burst<2>, dim3(blocks), dim3(256), 0, g_stream, dstop, g_sink, prio, 500000); break;
    case 4: hipLaunchKernelGGL(burst<4>, dim3(blocks), dim3(256), 0, g_stream, dstop, g_sink, prio, 500000); break;
    case 5: hipLaunchKernelGGL(burst<5>, dim3(blocks), dim3(256), 0, g_stream, dstop, g_sink, prio, 500000); break;
    case 6: hipLaunchKernelGGL(burst<6>, dim3(blocks), dim3(256), 0, g_stream, dstop, g_sink, prio, 500000); break;
    case 7: hipLaunchKernelGGL(burst<7>, dim3(blocks), dim3(256), 0, g_stream, dstop, g_sink, prio, 500000); break;
    case 8: hipLaunchKernelGGL(burst<8>, dim3(blocks), dim3(256), 0, g_stream, dstop, g_sink, prio, 500000); break;
    case 10: hipLaunchKernelGGL(burst<10>, dim3(blocks), dim3(256), 0, g_stream, dstop, g_sink, prio, 500000); break;
    case 9: hipLaunchKernelGGL(burst<9>, dim3(blocks), dim3(256), 0, g_stream, dstop, g_sink, prio, 500000); break;
    default: hipLaunchKernelGGL(burst<3>, dim3(blocks), dim3(256), 0, g_stream, dstop, g_sink, prio, 500000); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : -5;
}

extern "C" int coburst_stop(void) {
  if (!g_stop) return -1;
  *g_stop = 1;
  return hipStreamSynchronize(g_stream) == hipSuccess ? 0 : -2;
}
