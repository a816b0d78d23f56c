// The neighbour of the experiment: one small wave per SIMD that only issues matrix (or, kind 3, vector) instructions
// until the host sets *stop (tools/diag/coburst.hip, as loadable code object).  kinds: 0 v_mfma_f32_16x16x32_bf16,
// 1 v_mfma_i32_16x16x64_i8, 2 v_mfma_f32_16x16x4_f32, 3 v_fma_f32 only.
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int KIND>
__device__ __forceinline__ void burst(volatile int *stop, float *sink, int prio, int max_loops) {
  const int lane = threadIdx.x & 63;
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  if (prio == 3) __builtin_amdgcn_s_setprio(3);
  f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
  i4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
  bf8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  i4 ia = {lane, lane * 3, lane * 5, lane * 7}, ib = {lane * 11, lane * 13, lane * 17, lane * 19};
  float fa = 0.001f * lane, fb = 0.5f;
  for (int loop = 0; loop < max_loops; loop++) {
    for (int it = 0; it < 256; it++) {
      if constexpr (KIND == 0) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
      } else if constexpr (KIND == 1) {
        d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, d3, 0, 0, 0);
      } else if constexpr (KIND == 2) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c3, 0, 0, 0);
      } else {
        c0 = __builtin_elementwise_fma(c0, f4{fb, fb, fb, fb}, f4{fa, fa, fa, fa});
        c1 = __builtin_elementwise_fma(c1, f4{fb, fb, fb, fb}, f4{fa, fa, fa, fa});
      }
    }
    if (*stop) break;
  }
  const f4 c = c0 + c1 + c2 + c3;
  const i4 d = d0 + d1 + d2 + d3;
  sink[blockIdx.x * 256 + threadIdx.x] = c.x + c.y + c.z + c.w + (float)(d.x + d.y + d.z + d.w);
}
extern "C" __global__ __launch_bounds__(256) void burst0(volatile int *stop, float *sink, int prio, int max_loops) { burst<0>(stop, sink, prio, max_loops); }
extern "C" __global__ __launch_bounds__(256) void burst1(volatile int *stop, float *sink, int prio, int max_loops) { burst<1>(stop, sink, prio, max_loops); }
extern "C" __global__ __launch_bounds__(256) void burst2(volatile int *stop, float *sink, int prio, int max_loops) { burst<2>(stop, sink, prio, max_loops); }
extern "C" __global__ __launch_bounds__(256) void burst3(volatile int *stop, float *sink, int prio, int max_loops) { burst<3>(stop, sink, prio, max_loops); }
