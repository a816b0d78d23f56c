// Standalone reproducer (gfx950 / MI355X, ROCm 7.2): a packed-fp32 instruction whose LOW lane combines the LOW half of
// one register pair with the HIGH half of a later operand's pair - `v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]`:
// d.lo = a.lo + b.hi - gets d.lo of lanes 48-63 WRONG (computed as if b.hi were 0) while another wave of the same SIMD
// issues v_mfma_f32_16x16x32_bf16 back to back.  The same sum with the swapped operand first (op_sel:[1,0]
// op_sel_hi:[0,1], operands exchanged) or as two v_add_f32 is always right.  Found in round 4 of this repository through the
// fused FM-demodulation kernels (profiles/archive/r04_pk_opsel_hazard.md; the cut-out of their stage A is tools/diag/repro/).
//   hipcc --offload-arch=gfx950 -O3 -o pk_opsel_repro pk_opsel_repro.hip && ./pk_opsel_repro [launches] [iterations]
// Every victim wave checks itself: the result of the form under test against the same sum from two plain v_add_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s, line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ __launch_bounds__(256) void neighbour(volatile int *stop, float *sink, int max_loops) {   // one wave per SIMD, MFMAs only
  const int lane = threadIdx.x & 63;
  f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
  bf8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  for (int loop = 0; loop < max_loops && !*stop; loop++)
    for (int it = 0; it < 256; it++) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
  const f4 c = c0 + c1 + c2 + c3;
  sink[blockIdx.x * 256 + threadIdx.x] = c.x + c.y + c.z + c.w;
}

// FORM 0: v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]   (the failing one)     FORM 1: v_pk_add_f32 d, b, a op_sel:[1,0] op_sel_hi:[0,1]
// `a` is produced by the instruction right before (a v_pk_fma_f32), as in the kernels where this was found.
template <int FORM>
__global__ __launch_bounds__(256) void victim(unsigned *bad_by_lane, float *first_bad, int iters, float k0, float k1) {
  const int lane = threadIdx.x & 63;
  f2 a = {0.25f + 0.001f * lane, -0.5f + 0.002f * lane}, b = {0.125f - 0.003f * lane, 0.75f + 0.001f * lane};
  const f2 k = {k0, k1};
  unsigned bad = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      f2 d; float e0, e1;
      if constexpr (FORM == 0)
        asm volatile("v_pk_fma_f32 %0, %0, %3, %2 op_sel_hi:[1,0,1]\n\t"          // a = a * k.lo + b   (fresh producer of a)
                     "v_pk_add_f32 %1, %0, %2 op_sel:[0,1] op_sel_hi:[1,0]"        // d = (a.lo + b.hi, a.hi + b.lo)
                     : "+v"(a), "=&v"(d) : "v"(b), "s"(k));
      else
        asm volatile("v_pk_fma_f32 %0, %0, %3, %2 op_sel_hi:[1,0,1]\n\t"
                     "v_pk_add_f32 %1, %2, %0 op_sel:[1,0] op_sel_hi:[0,1]"        // the swapped operand first: same sums
                     : "+v"(a), "=&v"(d) : "v"(b), "s"(k));
      asm volatile("s_nop 3\n\tv_add_f32 %0, %2, %5\n\tv_add_f32 %1, %3, %4" : "=&v"(e0), "=&v"(e1) : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));
      if (__float_as_uint(d.x) != __float_as_uint(e0) || __float_as_uint(d.y) != __float_as_uint(e1)) {
        if (!bad) { first_bad[4 * (blockIdx.x * 256 + threadIdx.x)] = d.x; first_bad[4 * (blockIdx.x * 256 + threadIdx.x) + 1] = e0;
                    first_bad[4 * (blockIdx.x * 256 + threadIdx.x) + 2] = a.x; first_bad[4 * (blockIdx.x * 256 + threadIdx.x) + 3] = b.y; }
        bad++;
      }
      b = f2{b.y * 0.5f + 0.01f * u, b.x * -0.5f + 0.02f};                        // keep the values moving and bounded
      a = a * 0.25f;
    }
  }
  if (bad) atomicAdd(&bad_by_lane[lane], bad);
}

int main(int argc, char **argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20, iters = argc > 2 ? atoi(argv[2]) : 4000;
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  int *stop; CHECK(hipHostMalloc((void **)&stop, 64, hipHostMallocMapped));
  int *dstop; CHECK(hipHostGetDevicePointer((void **)&dstop, stop, 0));
  float *sink, *first; unsigned *bad;
  CHECK(hipMalloc(&sink, 256 * 256 * 4)); CHECK(hipMalloc(&bad, 64 * 4)); CHECK(hipMalloc(&first, 4 * 768 * 256 * 4));
  for (int form = 0; form < 2; form++)
    for (int with_nb = 0; with_nb < 2; with_nb++) {
      CHECK(hipMemset(bad, 0, 64 * 4)); CHECK(hipMemset(first, 0, 4 * 768 * 256 * 4)); CHECK(hipDeviceSynchronize());
      *stop = 0;
      if (with_nb) hipLaunchKernelGGL(neighbour, dim3(256), dim3(256), 0, s2, dstop, sink, 400000);
      for (int l = 0; l < launches; l++) {
        if (form == 0) hipLaunchKernelGGL(victim<0>, dim3(768), dim3(256), 0, s1, bad, first, iters, 0.999f, 0.5f);
        else hipLaunchKernelGGL(victim<1>, dim3(768), dim3(256), 0, s1, bad, first, iters, 0.999f, 0.5f);
        CHECK(hipStreamSynchronize(s1));
      }
      *stop = 1; CHECK(hipStreamSynchronize(s2));
      unsigned h[64]; CHECK(hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost));
      unsigned long q[4] = {0, 0, 0, 0};
      for (int i = 0; i < 64; i++) q[i >> 4] += h[i];
      printf("form %d (%s), %s: wrong results in lanes 0-15 %lu, 16-31 %lu, 32-47 %lu, 48-63 %lu  (%d launches x 3072 waves x %d sums)\n", form,
             form ? "swapped operand first, op_sel:[1,0]" : "op_sel:[0,1]", with_nb ? "MFMA neighbour on every SIMD" : "alone", q[0], q[1], q[2], q[3], launches, iters * 8);
      if (q[0] + q[1] + q[2] + q[3]) {
        static float hf[4 * 768 * 256]; CHECK(hipMemcpy(hf, first, sizeof hf, hipMemcpyDeviceToHost));
        for (int i = 0, shown = 0; i < 768 * 256 && shown < 4; i++)
          if (hf[4 * i] != 0.f || hf[4 * i + 1] != 0.f) { printf("   thread %d (lane %d): got %.9g want %.9g = a.lo %.9g + b.hi %.9g\n", i, i & 63, hf[4 * i], hf[4 * i + 1], hf[4 * i + 2], hf[4 * i + 3]); shown++; }
      }
    }
  return 0;
}
