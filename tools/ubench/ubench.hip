// Micro-benchmarks answering design questions for the FM kernels (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

// 16 independent accumulators, plain v_fma_f32
__global__ void k_fma(float *out, float a, float b) {
  float acc[16];
  for (int i = 0; i < 16; i++) acc[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_fmaf(acc[i], a, b);
  }
  float s = 0; for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// packed: 8 x v_pk_fma_f32 = 16 FMAs
__global__ void k_pkfma(float *out, float a, float b) {
  f2 acc[8];
  for (int i = 0; i < 8; i++) acc[i] = f2{(float)threadIdx.x + i, (float)i};
  f2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
  }
  float s = 0; for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// add + fma mix like the FIR inner loop: p = x + y; acc = fma(p, t, acc)
__global__ void k_addfma(float *out, float a, float b) {
  float acc[8], x[8];
  for (int i = 0; i < 8; i++) { acc[i] = threadIdx.x + i; x[i] = i * a; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) { float p = x[i] + acc[(i + 1) & 7]; acc[i] = __builtin_fmaf(p, a, acc[i]); }
  }
  float s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// v_cvt_f32_ubyte throughput
__global__ void k_cvt(float *out, unsigned w) {
  float acc[8]; unsigned x = w + threadIdx.x;
  for (int i = 0; i < 8; i++) acc[i] = i;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) { acc[i] += (float)((x >> (8 * (i & 3))) & 0xff); }
    x = x * 1664525u + 1013904223u;
  }
  float s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the MPX inner pattern: one pair-sum feeds three FMAs whose tap is a scalar (SGPR) operand
template <int NACC>
__global__ void k_mpx(float *out, float t0, float t1, float t2, float t3, float t4, float t5) {
  float am[NACC], ap[NACC], as[NACC], x[NACC + 4];
  for (int i = 0; i < NACC; i++) { am[i] = threadIdx.x + i; ap[i] = i; as[i] = -i; }
  for (int i = 0; i < NACC + 4; i++) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < ITERS / 2; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) {
      float p = x[i] + x[i + 3];
      am[i] = __builtin_fmaf(p, t0, am[i]); ap[i] = __builtin_fmaf(p, t1, ap[i]); as[i] = __builtin_fmaf(p, t2, as[i]);
      float q = x[i + 1] + x[i + 4];
      am[i] = __builtin_fmaf(q, t3, am[i]); ap[i] = __builtin_fmaf(q, t4, ap[i]); as[i] = __builtin_fmaf(q, t5, as[i]);
    }
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
  }
  float s = 0; for (int i = 0; i < NACC; i++) s += am[i] + ap[i] + as[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// same arithmetic with the iteration loop unrolled UNR times: straight-line code of UNR x 64
// instructions, to see what instruction fetch costs when the body is long
template <int UNR>
__global__ void k_mpx_long(float *out, float t0, float t1, float t2, float t3, float t4, float t5) {
  constexpr int NACC = 8;
  float am[NACC], ap[NACC], as[NACC], x[NACC + 4];
  for (int i = 0; i < NACC; i++) { am[i] = threadIdx.x + i; ap[i] = i; as[i] = -i; }
  for (int i = 0; i < NACC + 4; i++) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < ITERS / 2 / UNR; it++) {
#pragma unroll
    for (int u = 0; u < UNR; u++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) {
        float p = x[i] + x[i + 3];
        am[i] = __builtin_fmaf(p, t0, am[i]); ap[i] = __builtin_fmaf(p, t1, ap[i]); as[i] = __builtin_fmaf(p, t2, as[i]);
        float q = x[i + 1] + x[i + 4];
        am[i] = __builtin_fmaf(q, t3, am[i]); ap[i] = __builtin_fmaf(q, t4, ap[i]); as[i] = __builtin_fmaf(q, t5, as[i]);
      }
      asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    }
  }
  float s = 0; for (int i = 0; i < NACC; i++) s += am[i] + ap[i] + as[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// VOP3 (8-byte) encodings of the same arithmetic with the iteration loop unrolled UNR times: straight-line code of UNR x 64
// instructions, to see what instruction fetch costs when the body is long
template <int UNR>
__global__ void k_mpx_long_vop3(float *out, float t0, float t1, float t2, float t3, float t4, float t5) {
  constexpr int NACC = 8;
  float am[NACC], ap[NACC], as[NACC], x[NACC + 4];
  for (int i = 0; i < NACC; i++) { am[i] = threadIdx.x + i; ap[i] = i; as[i] = -i; }
  for (int i = 0; i < NACC + 4; i++) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < ITERS / 2 / UNR; it++) {
#pragma unroll
    for (int u = 0; u < UNR; u++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) {
        float p = x[i] + x[i + 3];
        am[i] = __builtin_fmaf(__builtin_fabsf(p), t0, am[i]); ap[i] = __builtin_fmaf(__builtin_fabsf(p), t1, ap[i]); as[i] = __builtin_fmaf(__builtin_fabsf(p), t2, as[i]);
        float q = x[i + 1] + x[i + 4];
        am[i] = __builtin_fmaf(__builtin_fabsf(q), t3, am[i]); ap[i] = __builtin_fmaf(__builtin_fabsf(q), t4, ap[i]); as[i] = __builtin_fmaf(__builtin_fabsf(q), t5, as[i]);
      }
      asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    }
  }
  float s = 0; for (int i = 0; i < NACC; i++) s += am[i] + ap[i] + as[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// v_dot4_i32_i8 with a scalar tap operand: is the integer dot product full rate?
__global__ void k_dot4(int *out, int t0, int t1, int t2) {
  int acc[16], x[8];
  for (int i = 0; i < 16; i++) acc[i] = threadIdx.x + i;
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 0x01010101 + i * 0x00030507;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_sdot4(x[i & 7], (i % 3 == 0) ? t0 : ((i % 3 == 1) ? t1 : t2), acc[i], false);
    asm volatile("" : "+v"(x[0]), "+v"(x[1]));
  }
  int s = 0; for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_perm(int *out, int sel) {
  int acc[16], x[8];
  for (int i = 0; i < 16; i++) acc[i] = threadIdx.x + i;
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 0x01010101 + i * 0x00030507;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_perm(acc[i], x[i & 7], sel);
    asm volatile("" : "+v"(x[0]), "+v"(x[1]));
  }
  int s = 0; for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the MPX pattern with packed math: pairs of outputs share a tap (scalar, broadcast to both halves)
__global__ void k_mpx_pk(float *out, float t0, float t1, float t2, float t3, float t4, float t5) {
  f2 am[4], ap[4], as[4], x[6];
  for (int i = 0; i < 4; i++) { am[i] = f2{(float)threadIdx.x + i, 1.f}; ap[i] = f2{(float)i, 2.f}; as[i] = f2{-(float)i, 3.f}; }
  for (int i = 0; i < 6; i++) x[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
  const f2 T0 = {t0, t0}, T1 = {t1, t1}, T2 = {t2, t2}, T3 = {t3, t3}, T4 = {t4, t4}, T5 = {t5, t5};
  for (int it = 0; it < ITERS / 2; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      f2 p = x[i] + x[i + 1];
      am[i] = __builtin_elementwise_fma(p, T0, am[i]); ap[i] = __builtin_elementwise_fma(p, T1, ap[i]); as[i] = __builtin_elementwise_fma(p, T2, as[i]);
      f2 q = x[i + 1] + x[i + 2];
      am[i] = __builtin_elementwise_fma(q, T3, am[i]); ap[i] = __builtin_elementwise_fma(q, T4, ap[i]); as[i] = __builtin_elementwise_fma(q, T5, as[i]);
    }
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
  }
  f2 s = {0.f, 0.f}; for (int i = 0; i < 4; i++) s += am[i] + ap[i] + as[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
// signed-byte conversion (v_cvt_f32_i32 with an SDWA sign-extended byte select) against v_cvt_f32_ubyteN
__global__ void k_cvt_s8(float *out, unsigned w) {
  float acc[8]; unsigned x = w + threadIdx.x;
  for (int i = 0; i < 8; i++) acc[i] = i;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) { acc[i] += (float)(signed char)((x >> (8 * (i & 3))) & 0xff); }
    x = x * 1664525u + 1013904223u;
  }
  float s = 0; for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// plain VOP2 ops with two VGPR sources
__global__ void k_add2(float *out, float a) {
  float acc[16], x[16];
  for (int i = 0; i < 16; i++) { acc[i] = threadIdx.x + i; x[i] = i * a; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = acc[i] + x[(i + 5) & 15];
    asm volatile("" : "+v"(x[0]), "+v"(x[1]));
  }
  float s = 0; for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// fma with three VGPR sources
__global__ void k_fma3(float *out, float a) {
  float acc[16], x[16], y[16];
  for (int i = 0; i < 16; i++) { acc[i] = threadIdx.x + i; x[i] = i * a; y[i] = 1.f + i * a; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = __builtin_fmaf(x[(i + 5) & 15], y[(i + 3) & 15], acc[i]);
    asm volatile("" : "+v"(x[0]), "+v"(y[1]));
  }
  float s = 0; for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__shared__ f4 lds[4096];
// LDS reads: mode 0 = b128 distinct contiguous, 1 = b128 broadcast (same address), 2 = b64 contiguous,
//            3 = b32 contiguous, 4 = b96-like broadcast (use 3 of 4), 5 = b64 broadcast
template <int MODE>
__global__ void k_lds(float *out, int stride) {
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = f4{(float)i, 1.f, 2.f, 3.f};
  __syncthreads();
  float acc = 0;
  const float *lf = (const float *)lds;
  const f2 *l2 = (const f2 *)lds;
  int base = (MODE == 0 || MODE == 2 || MODE == 3) ? threadIdx.x : 0;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      int idx = (base + (it & 15) * 64 + i * stride) & 2047;
      if (MODE == 0 || MODE == 1) { f4 v = lds[idx]; acc += v.x + v.w; }
      if (MODE == 4) { f4 v = lds[idx]; acc += v.x + v.z; }
      if (MODE == 2 || MODE == 5) { f2 v = l2[idx]; acc += v.x + v.y; }
      if (MODE == 3) { acc += lf[idx]; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename F>
float time_kernel(F launch, int reps = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  float *out; CHECK(hipMalloc(&out, 256 * 1024 * 4 * sizeof(float)));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  const int cus = prop.multiProcessorCount;
  /* settle the clocks first: after idle the part needs ~30 ms of work before its rates are steady */
  for (int i = 0; i < 300; i++) hipLaunchKernelGGL(k_fma, dim3(cus), dim3(768), 0, 0, out, 1.0001f, 0.5f);
  (void)hipDeviceSynchronize();
  for (int wpsimd : {1, 2, 4}) {
    int threads = 64 * 4 * wpsimd;  // one block per CU
    if (threads > 1024) continue;
    double lane_ops = (double)cus * threads * ITERS * 16;
    float t1 = time_kernel([&] { hipLaunchKernelGGL(k_fma, dim3(cus), dim3(threads), 0, 0, out, 1.0001f, 0.5f); });
    float t2 = time_kernel([&] { hipLaunchKernelGGL(k_pkfma, dim3(cus), dim3(threads), 0, 0, out, 1.0001f, 0.5f); });
    float t3 = time_kernel([&] { hipLaunchKernelGGL(k_addfma, dim3(cus), dim3(threads), 0, 0, out, 1.0001f, 0.5f); });
    float t4 = time_kernel([&] { hipLaunchKernelGGL(k_cvt, dim3(cus), dim3(threads), 0, 0, out, 12345u); });
    printf("waves/SIMD %d: fma %.3f ms = %.1f Tlane-FMA/s | pk_fma %.3f ms = %.1f Tlane-FMA/s | add+fma %.3f ms = %.1f Tlane-op/s | cvt+add %.3f ms = %.1f Tlane-op/s\n",
           wpsimd, t1, lane_ops / t1 / 1e9, t2, lane_ops / t2 / 1e9, t3, lane_ops / t3 / 1e9, t4, lane_ops / t4 / 1e9);
  }
  for (int wpsimd : {1, 2, 3, 4}) {
    int threads = 64 * 4 * wpsimd;
    double ops8 = (double)cus * threads * (ITERS / 2) * 8 * 8, ops16 = (double)cus * threads * ITERS * 16;
    float m8 = time_kernel([&] { hipLaunchKernelGGL(k_mpx<8>, dim3(cus), dim3(threads), 0, 0, out, 1.0001f, 0.5f, 0.25f, 0.3f, 0.7f, 0.9f); });
    float a2 = time_kernel([&] { hipLaunchKernelGGL(k_add2, dim3(cus), dim3(threads), 0, 0, out, 1.0001f); });
    float f3 = time_kernel([&] { hipLaunchKernelGGL(k_fma3, dim3(cus), dim3(threads), 0, 0, out, 1.0001f); });
    float l4 = time_kernel([&] { hipLaunchKernelGGL(k_mpx_long<4>, dim3(cus), dim3(threads), 0, 0, out, 1.0001f, 0.5f, 0.25f, 0.3f, 0.7f, 0.9f); });
    float l32 = time_kernel([&] { hipLaunchKernelGGL(k_mpx_long<32>, dim3(cus), dim3(threads), 0, 0, out, 1.0001f, 0.5f, 0.25f, 0.3f, 0.7f, 0.9f); });
    float l128 = time_kernel([&] { hipLaunchKernelGGL(k_mpx_long<128>, dim3(cus), dim3(threads), 0, 0, out, 1.0001f, 0.5f, 0.25f, 0.3f, 0.7f, 0.9f); });
    float v4 = time_kernel([&] { hipLaunchKernelGGL(k_mpx_long_vop3<4>, dim3(cus), dim3(threads), 0, 0, out, 1.0001f, 0.5f, 0.25f, 0.3f, 0.7f, 0.9f); });
    float v128 = time_kernel([&] { hipLaunchKernelGGL(k_mpx_long_vop3<128>, dim3(cus), dim3(threads), 0, 0, out, 1.0001f, 0.5f, 0.25f, 0.3f, 0.7f, 0.9f); });
    float d4 = time_kernel([&] { hipLaunchKernelGGL(k_dot4, dim3(cus), dim3(threads), 0, 0, (int *)out, 0x01020304, 0x7f80fe01, 0x10203040); });
    float pm = time_kernel([&] { hipLaunchKernelGGL(k_perm, dim3(cus), dim3(threads), 0, 0, (int *)out, 0x07020500); });
    printf("waves/SIMD %d: v_dot4_i32_i8 %.1f T lane-instr/s | v_perm_b32 %.1f\n", wpsimd, ops16 / d4 / 1e9, ops16 / pm / 1e9);
    float mpk = time_kernel([&] { hipLaunchKernelGGL(k_mpx_pk, dim3(cus), dim3(threads), 0, 0, out, 1.0001f, 0.5f, 0.25f, 0.3f, 0.7f, 0.9f); });
    printf("waves/SIMD %d: mpx pattern, packed: %.1f T lane-op/s (same arithmetic as the plain pattern below)\n", wpsimd, ops8 / mpk / 1e9);
    float cu = time_kernel([&] { hipLaunchKernelGGL(k_cvt, dim3(cus), dim3(threads), 0, 0, out, 12345u); });
    float cs = time_kernel([&] { hipLaunchKernelGGL(k_cvt_s8, dim3(cus), dim3(threads), 0, 0, out, 12345u); });
    printf("waves/SIMD %d: cvt ubyte + add %.1f | cvt signed byte (sdwa) + add %.1f T lane-instr/s\n", wpsimd, ops16 / cu / 1e9, ops16 / cs / 1e9);
    printf("waves/SIMD %d: 8-byte encodings x4 %.1f x128 %.1f T lane-instr/s\n", wpsimd, ops8 / v4 / 1e9, ops8 / v128 / 1e9);
    printf("waves/SIMD %d: mpx pattern %.3f ms = %.1f T lane-instr/s | add vgpr,vgpr %.1f | fma vgpr,vgpr,vgpr %.1f | straight-line x4 %.1f x32 %.1f x128 %.1f\n", wpsimd, m8,
           ops8 / m8 / 1e9, ops16 / a2 / 1e9, ops16 / f3 / 1e9, ops8 / l4 / 1e9, ops8 / l32 / 1e9, ops8 / l128 / 1e9);
  }
  for (int wpsimd : {1, 2, 4}) {
    int threads = 64 * 4 * wpsimd;
    double reads = (double)cus * threads * ITERS * 8;
    float a0 = time_kernel([&] { hipLaunchKernelGGL(k_lds<0>, dim3(cus), dim3(threads), 0, 0, out, 0); });
    float a1 = time_kernel([&] { hipLaunchKernelGGL(k_lds<1>, dim3(cus), dim3(threads), 0, 0, out, 1); });
    float a2 = time_kernel([&] { hipLaunchKernelGGL(k_lds<2>, dim3(cus), dim3(threads), 0, 0, out, 0); });
    float a3 = time_kernel([&] { hipLaunchKernelGGL(k_lds<3>, dim3(cus), dim3(threads), 0, 0, out, 0); });
    float a4 = time_kernel([&] { hipLaunchKernelGGL(k_lds<4>, dim3(cus), dim3(threads), 0, 0, out, 1); });
    float a5 = time_kernel([&] { hipLaunchKernelGGL(k_lds<5>, dim3(cus), dim3(threads), 0, 0, out, 1); });
    // cycles per wave-instruction per CU at 2.4 GHz
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / (reads / 64 / cus); };
    printf("waves/SIMD %d LDS cycles per wave-instr (per CU @2.4GHz): b128 %.2f | b128 bcast %.2f | b64 %.2f | b32 %.2f | b96 bcast %.2f | b64 bcast %.2f\n",
           wpsimd, cyc(a0), cyc(a1), cyc(a2), cyc(a3), cyc(a4), cyc(a5));
  }
  hipFree(out);
  return 0;
}
