// Which instruction of a wave goes wrong while ANOTHER wave of its SIMD issues dense v_mfma_f32_16x16x32_bf16?
// (tools/diag/coburst.py: the fused kernels' PCM changes under such a neighbour; v_mfma_f32_16x16x4_f32 and plain VALU
// neighbours change nothing.)  One kernel per instruction family ("chain"); every wave of the grid runs the same chain on
// the same lane-dependent data, so every wave's 64 results must equal the ones of a launch made without the neighbour.
//   hipcc --offload-arch=gfx950 -O3 -o valu_zoo valu_zoo.hip && ./valu_zoo [neighbour kind 0 bf16 | 1 i8 | 2 f32 | 3 valu] [reps] [iterations] [first chain] [s_setprio of the chains' waves]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) burst(volatile int *stop, float *sink, int max_loops) {
  const int lane = threadIdx.x & 63;
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

static const char *chain_name[] = {
  "v_fma_f32 (VGPR operands)", "v_fma_f32 (SGPR multiplier)", "v_pk_fma_f32 (VGPR operands)", "v_pk_fma_f32 (SGPR-pair multiplier)",
  "v_pk_mul_f32 + v_pk_add_f32", "v_add_f32 dpp wave_ror:1", "v_cvt_f32_i32 sdwa bytes", "v_rcp_f32", "ds_write_b128 / ds_read_b128",
  "global_load_dwordx4 of a pattern", "v_mad_u32_u24 / v_mul_hi_u32 / v_bfi", "ds_bpermute_b32", "v_readlane / v_readfirstlane sums", "v_mov_b32 dpp row_shr:1 + quad_perm",
  "v_add_f32 dpp row_bcast / wave_shr (reductions)", "v_min_f32 / v_cmp + v_cndmask / v_med3",
  "v_fma_f32 over 120 live registers, all operand distances", "v_pk_fma_f32 over 60 live register pairs", "v_pk_fma_f32, 60 pairs, SGPR-pair multipliers + v_pk_add",
  "v_pk_fma_f32 V2, V2, S2, V2 op_sel_hi:[1,0,1]", "v_pk_fma_f32 V2, V2, S2, V2 op_sel:[0,1,0]", "v_pk_fma_f32 ... op_sel_hi:[1,0,1] neg_lo/neg_hi:[0,1,0]",
  "v_fmamk_f32 (literal multiplier)", "v_pk_fma_f32 with taps just loaded by s_load_dwordx4",
  "8 global_load_dwordx4 in flight, each used right after its own s_waitcnt vmcnt(n)", "8 ds_read_b128 in flight, each used right after its own s_waitcnt lgkmcnt(n)",
  "8 buffer_load_dwordx4 in flight under VALU work, used after vmcnt(n)"};
constexpr int N_CHAINS = 27;

template <int CH>
__global__ void __launch_bounds__(256) zoo(unsigned *out, const uint4 *pattern, int iters, float sk0, float sk1, const f4 *ftab, int prio) {
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  if (prio == 2) __builtin_amdgcn_s_setprio(2);
  if (prio == 3) __builtin_amdgcn_s_setprio(3);
  __shared__ __attribute__((aligned(16))) float lds[4 * 64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float x = 0.25f + 0.001f * lane, y = -0.5f + 0.002f * lane, z = 0.125f;
  f2 p = {x, y}, q = {0.5f * y, 0.25f * x}, acc2 = {0.f, 0.f};
  float acc = 0.f;
  unsigned u = 0x9e3779b9u * (lane + 1), h = 0;
  for (int it = 0; it < iters; it++) {
    if constexpr (CH == 0) {
#pragma unroll
      for (int k = 0; k < 16; k++) { acc = __builtin_fmaf(acc, 0.999f, x); x = __builtin_fmaf(x, 0.5f, y); asm volatile("" : "+v"(acc), "+v"(x)); }
    } else if constexpr (CH == 1) {
#pragma unroll
      for (int k = 0; k < 16; k++) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "s"(sk0), "v"(x)); asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x) : "s"(sk1), "v"(y)); }
    } else if constexpr (CH == 2) {
#pragma unroll
      for (int k = 0; k < 16; k++) { acc2 = __builtin_elementwise_fma(acc2, f2{0.999f, 0.998f}, p); p = __builtin_elementwise_fma(p, f2{0.5f, 0.25f}, q); asm volatile("" : "+v"(acc2), "+v"(p)); }
    } else if constexpr (CH == 3) {
      const f2 sk = {sk0, sk1};
#pragma unroll
      for (int k = 0; k < 16; k++) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2) : "s"(sk), "v"(p)); asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(p) : "s"(sk), "v"(q)); }
    } else if constexpr (CH == 4) {
#pragma unroll
      for (int k = 0; k < 16; k++) {
        f2 t;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(p), "v"(q));
        asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(acc2) : "v"(t));
        p = p * f2{0.75f, 0.5f} + q;
        asm volatile("" : "+v"(p));
      }
    } else if constexpr (CH == 5) {
#pragma unroll
      for (int k = 0; k < 16; k++) { asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %0 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x)); x = x * 0.5f + y; asm volatile("" : "+v"(x)); }
    } else if constexpr (CH == 6) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float c0, c1, c2, c3; int w = (int)(u ^ 0x80808080u);
        asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(c0) : "v"(w));
        asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(c1) : "v"(w));
        asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(c2) : "v"(w));
        asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(c3) : "v"(w));
        acc = acc * 0.5f + ((c0 + c1) + (c2 + c3));
        u = u * 1664525u + 1013904223u;
      }
    } else if constexpr (CH == 7) {
#pragma unroll
      for (int k = 0; k < 8; k++) { float r; asm volatile("v_rcp_f32 %0, %1\n\ts_nop 1" : "=v"(r) : "v"(x)); acc = acc * 0.5f + r; x = x * 0.999f + 0.01f; asm volatile("" : "+v"(x)); }
    } else if constexpr (CH == 8) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        f4 v = {x, y, acc, z};
        *reinterpret_cast<f4 *>(&lds[(wave * 64 + lane) * 4]) = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const f4 r = *reinterpret_cast<const f4 *>(&lds[(wave * 64 + ((lane + 1) & 63)) * 4]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        acc = acc * 0.5f + (r.x + r.y) + (r.z + r.w); x = x * 0.5f + 0.25f;
      }
    } else if constexpr (CH == 9) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint4 w = pattern[((it * 4 + k) & 1023) * 64 + lane];
        h = h * 31u + (w.x ^ (w.y * 3u) ^ (w.z * 5u) ^ (w.w * 7u));
      }
    } else if constexpr (CH == 10) {
#pragma unroll
      for (int k = 0; k < 16; k++) { h = __umulhi(u, 0x51eb851fu) + (u & 0xffffffu) * (h & 0xfffu); u = (u & 0x00ff00ffu) | (h & ~0x00ff00ffu); u = u * 1664525u + 1013904223u; }
    } else if constexpr (CH == 11) {
#pragma unroll
      for (int k = 0; k < 4; k++) { const int r = __builtin_amdgcn_ds_bpermute(((lane + 1 + k) & 63) * 4, (int)u); h = h * 31u + (unsigned)r; u = u * 1664525u + 1013904223u; }
    } else if constexpr (CH == 12) {
#pragma unroll
      for (int k = 0; k < 4; k++) { const unsigned a = __builtin_amdgcn_readlane(u, 48 + k), b = __builtin_amdgcn_readlane(u, 63 - k), c = __builtin_amdgcn_readfirstlane(u); h = h * 31u + (a ^ (b * 3u) ^ (c * 5u)); u = u * 1664525u + 1013904223u + h; }
    } else if constexpr (CH == 13) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int a = __builtin_amdgcn_update_dpp(0, (int)u, 0x111, 0xf, 0xf, false);       /* row_shr:1 */
        const int b = __builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xf, 0xf, true);          /* quad_perm [1,0,3,2] */
        h = h * 31u + (unsigned)a + 3u * (unsigned)b; u = u * 1664525u + 1013904223u;
      }
    } else if constexpr (CH == 14) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float s = x;
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s));
        acc = acc * 0.5f + s; x = x * 0.75f + 0.1f;
      }
    } else if constexpr (CH == 19 || CH == 20 || CH == 21) {
      const f2 sk = {sk0, sk1};
#pragma unroll
      for (int k = 0; k < 16; k++) {
        if constexpr (CH == 19) { asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[1,0,1]" : "+v"(acc2) : "s"(sk), "v"(p)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p) : "s"(sk), "v"(q)); }
        if constexpr (CH == 20) { asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[0,1,0]" : "+v"(acc2) : "s"(sk), "v"(p)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0]" : "+v"(p) : "s"(sk), "v"(q)); }
        if constexpr (CH == 21) { asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "+v"(acc2) : "s"(sk), "v"(p)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p) : "s"(sk), "v"(q)); }
      }
    } else if constexpr (CH == 22) {
#pragma unroll
      for (int k = 0; k < 16; k++) { asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(acc) : "v"(x)); asm volatile("v_fmamk_f32 %0, %0, 0x3f000000, %1" : "+v"(x) : "v"(y)); }
    } else if constexpr (CH == 23) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int idx = __builtin_amdgcn_readfirstlane((it * 4 + k) & 1023);
        const f4 t = ftab[idx];                                    /* uniform address: s_load_dwordx4 */
        asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[1,0,1]" : "+v"(acc2) : "s"(f2{t.x, t.y}), "v"(p));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p) : "s"(f2{t.z, t.w}), "v"(q));
      }
    } else if constexpr (CH == 24 || CH == 26) {
      const uint4 *src = pattern + ((it * 8) & 1023) * 64 + lane;      /* rows of 64 words: word k of this iteration at + 64 k */
      uint4 w0, w1, w2, w3, w4, w5, w6, w7;
      asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:1024\n\tglobal_load_dwordx4 %2, %8, off offset:2048\n\t"
                   "global_load_dwordx4 %3, %8, off offset:3072\n\tglobal_load_dwordx4 %4, %9, off\n\tglobal_load_dwordx4 %5, %9, off offset:1024\n\t"
                   "global_load_dwordx4 %6, %9, off offset:2048\n\tglobal_load_dwordx4 %7, %9, off offset:3072"
                   : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5), "=&v"(w6), "=&v"(w7) : "v"(src), "v"(src + 256) : "memory");
      if constexpr (CH == 26) {                      /* arithmetic while they fly, like stages D / F under the prefetch */
#pragma unroll
        for (int k = 0; k < 32; k++) { acc2 = __builtin_elementwise_fma(acc2, f2{0.999f, 0.998f}, p); p = __builtin_elementwise_fma(p, f2{0.5f, 0.25f}, q); asm volatile("" : "+v"(acc2), "+v"(p)); }
      }
#define USE_WORD(N, W) asm volatile("s_waitcnt vmcnt(" #N ")\n\tv_alignbit_b32 %0, %0, %0, 27\n\tv_xor_b32 %0, %0, %1\n\tv_alignbit_b32 %0, %0, %0, 27\n\tv_xor_b32 %0, %0, %2\n\t" \
                                "v_alignbit_b32 %0, %0, %0, 27\n\tv_xor_b32 %0, %0, %3\n\tv_alignbit_b32 %0, %0, %0, 27\n\tv_xor_b32 %0, %0, %4" : "+v"(h) : "v"(W.x), "v"(W.y), "v"(W.z), "v"(W.w))
      USE_WORD(7, w0); USE_WORD(6, w1); USE_WORD(5, w2); USE_WORD(4, w3); USE_WORD(3, w4); USE_WORD(2, w5); USE_WORD(1, w6); USE_WORD(0, w7);
#undef USE_WORD
    } else if constexpr (CH == 25) {
      if (it == 0) {
        for (int i = lane; i < 256; i += 64) lds[wave * 256 + i] = (float)(i * 7);     /* (4 KB in all: 64 words of 16 bytes per wave) */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      const unsigned a = (unsigned)(size_t)(&lds[wave * 256]) + 16 * ((lane + it) & 7);
      uint4 w0, w1, w2, w3, w4, w5, w6, w7;
      asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:128\n\tds_read_b128 %2, %8 offset:256\n\tds_read_b128 %3, %8 offset:384\n\t"
                   "ds_read_b128 %4, %8 offset:512\n\tds_read_b128 %5, %8 offset:640\n\tds_read_b128 %6, %8 offset:768\n\tds_read_b128 %7, %8 offset:896"
                   : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5), "=&v"(w6), "=&v"(w7) : "v"(a) : "memory");
#define USE_WORD(N, W) asm volatile("s_waitcnt lgkmcnt(" #N ")\n\tv_alignbit_b32 %0, %0, %0, 27\n\tv_xor_b32 %0, %0, %1\n\tv_alignbit_b32 %0, %0, %0, 27\n\tv_xor_b32 %0, %0, %2\n\t" \
                                "v_alignbit_b32 %0, %0, %0, 27\n\tv_xor_b32 %0, %0, %3\n\tv_alignbit_b32 %0, %0, %0, 27\n\tv_xor_b32 %0, %0, %4" : "+v"(h) : "v"(W.x), "v"(W.y), "v"(W.z), "v"(W.w))
      USE_WORD(7, w0); USE_WORD(6, w1); USE_WORD(5, w2); USE_WORD(4, w3); USE_WORD(3, w4); USE_WORD(2, w5); USE_WORD(1, w6); USE_WORD(0, w7);
#undef USE_WORD
      h += (unsigned)it;
    } else if constexpr (CH >= 16) {
      /* handled below (own loop) */
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float m = __builtin_fminf(x, y), M = __builtin_amdgcn_fmed3f(x, y, z);
        acc = acc * 0.5f + (x > z ? m : M);
        x = x * 0.75f + 0.1f; y = y * -0.5f + 0.05f; z = z * 0.9f + 0.01f * acc;
        asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
      }
    }
  }
  if constexpr (CH == 16) {
    float r[120];
#pragma unroll
    for (int i = 0; i < 120; i++) r[i] = 0.01f * (float)(i + 1) + 0.001f * lane;
    for (int it = 0; it < iters / 8; it++) {
#pragma unroll
      for (int d = 1; d <= 7; d += 2) {
#pragma unroll
        for (int i = 0; i < 120; i++) r[i] = __builtin_fmaf(r[(i + d) % 120], 0.5f, r[(i + 3 * d + 1) % 120] * 0.25f + 0.001f);
#pragma unroll
        for (int i = 0; i < 120; i += 8) asm volatile("" : "+v"(r[i]), "+v"(r[i + 1]), "+v"(r[i + 2]), "+v"(r[i + 3]), "+v"(r[i + 4]), "+v"(r[i + 5]), "+v"(r[i + 6]), "+v"(r[i + 7]));
      }
    }
#pragma unroll
    for (int i = 0; i < 120; i++) h = h * 31u + __builtin_bit_cast(unsigned, r[i]);
  }
  if constexpr (CH == 17 || CH == 18) {
    f2 r[60];
    const f2 sk = {sk0, sk1};
#pragma unroll
    for (int i = 0; i < 60; i++) r[i] = f2{0.01f * (float)(i + 1) + 0.001f * lane, 0.02f * (float)(i + 1) - 0.001f * lane};
    for (int it = 0; it < iters / 8; it++) {
#pragma unroll
      for (int d = 1; d <= 7; d += 2) {
#pragma unroll
        for (int i = 0; i < 60; i++) {
          if constexpr (CH == 17) r[i] = __builtin_elementwise_fma(r[(i + d) % 60], f2{0.5f, 0.25f}, r[(i + 3 * d + 1) % 60] * f2{0.25f, 0.5f});
          else {
            f2 t = r[(i + d) % 60] + r[(i + 3 * d + 1) % 60];
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r[i]) : "s"(sk), "v"(t), "v"(r[(i + 2 * d) % 60]));
            r[i] *= f2{0.25f, 0.25f};
          }
        }
#pragma unroll
        for (int i = 0; i < 60; i += 4) asm volatile("" : "+v"(r[i]), "+v"(r[i + 1]), "+v"(r[i + 2]), "+v"(r[i + 3]));
      }
    }
#pragma unroll
    for (int i = 0; i < 60; i++) h = h * 31u + __builtin_bit_cast(unsigned, r[i].x) + 7u * __builtin_bit_cast(unsigned, r[i].y);
  }
  const unsigned r = __builtin_bit_cast(unsigned, acc) ^ __builtin_bit_cast(unsigned, acc2.x) ^ (__builtin_bit_cast(unsigned, acc2.y) * 3u) ^
                     (__builtin_bit_cast(unsigned, x) * 5u) ^ (__builtin_bit_cast(unsigned, p.x) * 7u) ^ (__builtin_bit_cast(unsigned, p.y) * 11u) ^ h ^ (u * 13u);
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}

static const f4 *g_ftab;
static int g_prio;
template <int CH>
static void launch_zoo(hipStream_t st, int grid, unsigned *out, const uint4 *pat, int iters) {
  hipLaunchKernelGGL(zoo<CH>, dim3(grid), dim3(256), 0, st, out, pat, iters, 0.999f, 0.5f, g_ftab, g_prio);
}
typedef void (*launch_fn)(hipStream_t, int, unsigned *, const uint4 *, int);
static launch_fn launchers[N_CHAINS] = {launch_zoo<0>, launch_zoo<1>, launch_zoo<2>, launch_zoo<3>, launch_zoo<4>, launch_zoo<5>, launch_zoo<6>, launch_zoo<7>,
                                        launch_zoo<8>, launch_zoo<9>, launch_zoo<10>, launch_zoo<11>, launch_zoo<12>, launch_zoo<13>, launch_zoo<14>, launch_zoo<15>, launch_zoo<16>, launch_zoo<17>, launch_zoo<18>, launch_zoo<19>, launch_zoo<20>, launch_zoo<21>, launch_zoo<22>, launch_zoo<23>, launch_zoo<24>, launch_zoo<25>, launch_zoo<26>};

int main(int argc, char **argv) {
  const int kind = argc > 1 ? atoi(argv[1]) : 0, reps = argc > 2 ? atoi(argv[2]) : 5;
  const int grid = 256 * 3, iters = argc > 3 ? atoi(argv[3]) : 4000;
  const int first_chain = argc > 4 ? atoi(argv[4]) : 0;
  g_prio = argc > 5 ? atoi(argv[5]) : 0;                 /* s_setprio of the zoo's waves (the neighbour stays at 0) */
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  int *stop; CHECK(hipHostMalloc((void **)&stop, 64, hipHostMallocMapped));
  int *dstop; CHECK(hipHostGetDevicePointer((void **)&dstop, stop, 0));
  float *sink; CHECK(hipMalloc(&sink, 256 * 256 * sizeof(float)));
  unsigned *out; CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
  std::vector<uint4> hp(1024 * 64 + 1024);
  for (size_t i = 0; i < hp.size(); i++) hp[i] = uint4{(unsigned)(i * 2654435761u), (unsigned)(i * 40503u + 7), (unsigned)(i ^ 0x5a5a5a5au), (unsigned)(i * i + 1)};
  uint4 *pat; CHECK(hipMalloc(&pat, hp.size() * sizeof(uint4)));
  CHECK(hipMemcpy(pat, hp.data(), hp.size() * sizeof(uint4), hipMemcpyHostToDevice));
  std::vector<f4> hf(1024);
  for (int i = 0; i < 1024; i++) hf[i] = f4{0.5f + 0.0001f * i, 0.25f, 0.125f + 0.0002f * i, 0.75f};
  f4 *ftab; CHECK(hipMalloc(&ftab, hf.size() * sizeof(f4)));
  CHECK(hipMemcpy(ftab, hf.data(), hf.size() * sizeof(f4), hipMemcpyHostToDevice));
  g_ftab = ftab;
  std::vector<unsigned> ref((size_t)grid * 256), got((size_t)grid * 256);
  printf("neighbour kind %d (0 bf16 16x16x32, 1 i8 16x16x64, 2 f32 16x16x4, 3 v_fma), %d waves per chain and launch, %d launches, s_setprio %d\n", kind, grid * 4, reps, g_prio);
  for (int ch = first_chain; ch < N_CHAINS; ch++) {
    launchers[ch](s1, grid, out, pat, iters);
    CHECK(hipStreamSynchronize(s1));
    CHECK(hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost));
    long self_bad = 0;                                  /* every wave must equal wave 0 already */
    for (size_t i = 0; i < ref.size(); i++) self_bad += ref[i] != ref[i & 63];
    long bad = 0, bad_waves = 0, by_row[4] = {0, 0, 0, 0}; int overlapped = 0;
    for (int r = 0; r < reps; r++) {
      *stop = 0;
      switch (kind) {
        case 0: hipLaunchKernelGGL(burst<0>, dim3(256), dim3(256), 0, s2, dstop, sink, 300000); break;
        case 1: hipLaunchKernelGGL(burst<1>, dim3(256), dim3(256), 0, s2, dstop, sink, 300000); break;
        case 2: hipLaunchKernelGGL(burst<2>, dim3(256), dim3(256), 0, s2, dstop, sink, 300000); break;
        default: hipLaunchKernelGGL(burst<3>, dim3(256), dim3(256), 0, s2, dstop, sink, 300000); break;
      }
      launchers[ch](s1, grid, out, pat, iters);
      CHECK(hipStreamSynchronize(s1));
      const bool neighbour_alive = hipStreamQuery(s2) == hipErrorNotReady;
      overlapped += neighbour_alive;
      *stop = 1;
      CHECK(hipStreamSynchronize(s2));
      CHECK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
      for (size_t w = 0; w < got.size() / 64; w++) {
        int wb = 0;
        for (int l = 0; l < 64; l++) if (got[w * 64 + l] != ref[l]) { wb++; by_row[l >> 4]++; }
        bad += wb; bad_waves += wb > 0;
      }
    }
    printf("chain %2d %-52s clean launch self-consistent: %s | with neighbour: %ld wrong lanes in %ld of %ld waves (lanes 0-15 %ld, 16-31 %ld, 32-47 %ld, 48-63 %ld); neighbour still running at the end of %d launches\n", ch,
           chain_name[ch], self_bad ? "NO" : "yes", bad, bad_waves, (long)reps * grid * 4, by_row[0], by_row[1], by_row[2], by_row[3], overlapped);
    fflush(stdout);
  }
  return 0;
}
