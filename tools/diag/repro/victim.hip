// Victim of the neighbour experiment (profiles/archive/r03m_*), cut out of the product: stage A of the
// vector-ALU fast family (decimate8_own of csrc/fmd_kernels.inc: the lane's own eight IQ words, SDWA byte conversions,
// v_pk_fma_f32 with scalar taps, the rot90 join as v_pk_add_f32 with operand modifiers, DPP hand-over of three outputs)
// and nothing else.  Every wave of the grid runs the same tiles of the same IQ bytes, so every wave's outputs must equal
// the ones of a launch made with no neighbour on the device.  Device code only; built into code objects by mkvariants.py
// (source-level switches -DV_* and assembly-level variants), loaded by host.cpp through the module API.
#include <hip/hip_runtime.h>
#include <type_traits>
typedef float f2 __attribute__((ext_vector_type(2)));
#ifndef V_WAVES
#define V_WAVES 3          /* launch bounds: waves per SIMD the kernel is register-budgeted for */
#endif

struct vparams { float g[16]; float c_i, c_q; };   /* taps / 128 and the constant term (fmdk_params.fbs, c_i, c_q) */

template <int B, int E, typename F> __device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
__device__ __forceinline__ int opaque_zero() { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }
// The join of the even- and odd-phase sums.  V_ROTFORM picks the instruction form under test (the results of forms other
// than 0 / 9 are not the decimator's - every run is only compared with a clean launch of the same code object):
//  0 shipped: v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]     1 the same without neg     2 neg_lo only (no half swap)
//  3 op_sel:[1,1] op_sel_hi:[0,0] (both operands swapped)     4 op_sel_hi:[1,0] only (low half of b to both lanes)     5 no modifier
//  6 v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_hi:[0,0,1] (stage B's dot / cross form)     7 v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]
//  8 op_sel:[1,0] (a swapped: a.hi to the low lane)     9 two VOP2: v_sub_f32 / v_add_f32     10 two VOP3 with neg: v_add_f32 a.x, -b.y / v_add_f32 a.y, b.x
//  11 v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,1] (b.hi to BOTH lanes)     12 v_pk_mov_b32-style swap first (two v_mov), then neg_lo only
#ifndef V_ROTFORM
#define V_ROTFORM 0
#endif
__device__ __forceinline__ f2 pk_add_rot90(f2 a, f2 b) {          /* a + (-b.y, b.x) */
#ifdef V_PLAINROT
  return f2{a.x - b.y, a.y + b.x};
#else
  f2 r;
#ifdef V_JOINPAD
  asm volatile("s_nop 0" : "+v"(a), "+v"(b));
#endif
  if constexpr (V_ROTFORM == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (V_ROTFORM == 1) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (V_ROTFORM == 2) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (V_ROTFORM == 3) asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (V_ROTFORM == 4) asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (V_ROTFORM == 5) asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (V_ROTFORM == 6) asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (V_ROTFORM == 7) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (V_ROTFORM == 8) asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (V_ROTFORM == 9) { asm("v_sub_f32 %0, %1, %2" : "=v"(r.x) : "v"(a.x), "v"(b.y)); asm("v_add_f32 %0, %1, %2" : "=v"(r.y) : "v"(a.y), "v"(b.x)); }
  else if constexpr (V_ROTFORM == 10) { asm("v_add_f32_e64 %0, %1, -%2" : "=v"(r.x) : "v"(a.x), "v"(b.y)); asm("v_add_f32_e64 %0, |%1|, %2" : "=v"(r.y) : "v"(a.y), "v"(b.x)); }
  else if constexpr (V_ROTFORM == 11) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (V_ROTFORM == 12) { f2 t; asm("v_mov_b32 %0, %1" : "=v"(t.x) : "v"(b.y)); asm("v_mov_b32 %0, %1" : "=v"(t.y) : "v"(b.x)); asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(t)); }
  /* 13 v_pk_mov_b32 op_sel:[1,0] (+ a plain add)   14 mirror of 0: the fresh operand second, op_sel:[1,0] op_sel_hi:[0,1]   15 v_pk_mul_f32 op_sel:[1,0]
   * with the fresh operand second (b.hi to both lanes)   16 scalar pair with op_sel:[0,1,0]   17 horizontal on the fresh operand   18 horizontal on the other */
  else if constexpr (V_ROTFORM == 13) { f2 t; asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(a), "v"(b)); asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(b)); }
  else if constexpr (V_ROTFORM == 14) asm("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (V_ROTFORM == 15) asm("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (V_ROTFORM == 16) { const f2 sk = {0.75f, 0.625f}; asm("v_pk_fma_f32 %0, %1, %3, %2 op_sel:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "s"(sk)); }
  else if constexpr (V_ROTFORM == 17) { f2 t; asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a)); asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(b)); }
  else { f2 t; asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(b)); asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(a)); }
  return r;
#endif
}
__device__ __forceinline__ float sbyte(uint32_t w, int i) { return (float)(int8_t)((w >> (8 * i)) & 0xffu); }
__device__ __forceinline__ constexpr float sgn_i(int p) { return (p == 0 || p == 3) ? 1.f : -1.f; }
__device__ __forceinline__ constexpr float sgn_q(int p) { return (p == 0 || p == 1) ? 1.f : -1.f; }

__device__ __forceinline__ f2 fma2(f2 a, float s, f2 c) {
#ifdef V_SCALARFMA
  return f2{__builtin_fmaf(a.x, s, c.x), __builtin_fmaf(a.y, s, c.y)};
#else
  return __builtin_elementwise_fma(a, f2{s, s}, c);
#endif
}

__device__ __forceinline__ void decimate8_own(const vparams &P, uint4 (&q)[8], f2 (&y2)[8], f2 (&carry)[3], int lane) {
  const int z = opaque_zero();
  float g[16];
#pragma unroll
  for (int k = 0; k < 16; k++) g[k] = P.g[k + z];
#ifdef V_VTAPS
#pragma unroll
  for (int k = 0; k < 16; k++) asm volatile("" : "+v"(g[k]));
#endif
  f2 pa[11], ma[11];
  const f2 c0 = {P.c_i, P.c_q};
  static_for<0, 8>([&](auto w_) {
    constexpr int wd = decltype(w_)::value;
    const uint4 x = q[wd];
    uint32_t kx = 0x80808080u;
    if constexpr (wd == 0) asm volatile("" : "+s"(kx));
    else asm volatile("" : "+v"(pa[wd]), "+v"(pa[wd + 1]), "+v"(pa[wd + 2]), "+v"(ma[wd]), "+v"(ma[wd + 1]), "+v"(ma[wd + 2]), "+s"(kx));
    const uint32_t d[4] = {x.x ^ kx, x.y ^ kx, x.z ^ kx, x.w ^ kx};
    static_for<0, 8>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
      const float ui = sbyte(d[t >> 1], 2 * (t & 1)), uq = sbyte(d[t >> 1], 2 * (t & 1) + 1);
      const f2 uiq = {ui, uq};
      static_for<0, 4>([&](auto o_) {
        constexpr int r = wd + decltype(o_)::value, j = 8 * (wd - r + 3) + t;
        constexpr int p = j & 3;
        const float gk = g[j < 16 ? j : 31 - j];
        constexpr bool first_word = wd == (r > 3 ? r - 3 : 0);
        if constexpr ((p & 1) == 0) {
          const float sg = sgn_i(p) * gk;
          if constexpr (first_word && t == 0) {
            if constexpr (r < 8) pa[r] = fma2(uiq, sg, c0);
            else pa[r] = uiq * f2{sg, sg};
          } else {
            pa[r] = fma2(uiq, sg, pa[r]);
          }
        } else {
          const float sg = sgn_q(p) * gk;
          if constexpr (first_word && t == 1) ma[r] = uiq * f2{sg, sg};
          else ma[r] = fma2(uiq, sg, ma[r]);
        }
      });
    });
  });
#pragma unroll
  for (int r = 0; r < 8; r++) y2[r] = pk_add_rot90(pa[r], ma[r]);
  float o[6], snd[6], a[6];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const f2 out = pk_add_rot90(pa[8 + k], ma[8 + k]);
    o[2 * k] = out.x; o[2 * k + 1] = out.y;
    snd[2 * k] = (lane == 63) ? carry[k].x : out.x;
    snd[2 * k + 1] = (lane == 63) ? carry[k].y : out.y;
    a[2 * k] = y2[k].x; a[2 * k + 1] = y2[k].y;
  }
#ifndef V_NODPP
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %6, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %7, %1 wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %2, %8, %2 wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %3, %9, %3 wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %4, %10, %4 wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %5, %11, %5 wave_ror:1 row_mask:0xf bank_mask:0xf"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5])
               : "v"(snd[0]), "v"(snd[1]), "v"(snd[2]), "v"(snd[3]), "v"(snd[4]), "v"(snd[5]));
#else
#pragma unroll
  for (int k = 0; k < 6; k++) a[k] += snd[k] * 0.f;     /* keep the values live, no cross-lane operation */
#endif
#pragma unroll
  for (int k = 0; k < 3; k++) {
    y2[k] = f2{a[2 * k], a[2 * k + 1]};
    carry[k] = f2{o[2 * k], o[2 * k + 1]};
  }
}

__device__ __forceinline__ void load_words(uint4 (&q)[8], __amdgpu_buffer_rsrc_t rsrc, int tile, int lane, int tiles) {
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
#ifdef V_NOLOAD
  /* no memory instruction at all: the words are a hash of (tile, lane, word) */
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t h = (uint32_t)(tile * 512 + lane * 8 + i) * 2654435761u;
    q[i] = make_uint4(h, h * 40503u + 7u, h ^ 0x5a5a5a5au, h * h + 1u);
  }
  (void)rsrc; (void)tiles;
#else
  const int byte0 = (tile % tiles) * 8192 + lane * 128;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u4v t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte0 + 16 * i, 0, 0);
    q[i] = make_uint4(t.x, t.y, t.z, t.w);
  }
#endif
}

// out: [wave of the grid][tile][512] float2, every wave the same tiles of the same bytes
extern "C" __global__ __launch_bounds__(256, V_WAVES) void victim(const vparams P, const uint8_t *__restrict__ iq, float2 *__restrict__ out,
                                                                  int tiles, int prio) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int unit = blockIdx.x * 4 + wave;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(iq), 0, (uint32_t)tiles * 8192u, 0x00020000);
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  if (prio == 2) __builtin_amdgcn_s_setprio(2);
  if (prio == 3) __builtin_amdgcn_s_setprio(3);
  f2 carry[3] = {};
  uint4 qn[8];
#ifndef V_LOADTOP
  load_words(qn, rsrc, 0, lane, tiles);
#endif
  float2 *o = out + (size_t)unit * tiles * 512;
  for (int g = 0; g < tiles; g++) {
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));
#ifdef V_PRIOFLIP
    __builtin_amdgcn_s_setprio(0);
#endif
    uint4 q[8];
#ifdef V_LOADTOP
    load_words(q, rsrc, g, lane_t, tiles);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#else
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = qn[i];
#endif
    f2 y2[8];
    decimate8_own(P, q, y2, carry, lane_t);
#ifdef V_PRIOFLIP
    __builtin_amdgcn_s_setprio(3);
#endif
#ifndef V_LOADTOP
    load_words(qn, rsrc, g + 1, lane_t, tiles);
#endif
    float2 *ot = o + (size_t)g * 512 + 8 * lane_t;
#pragma unroll
    for (int r = 0; r < 8; r++) ot[r] = make_float2(y2[r].x, y2[r].y);
  }
}

// counts[0..3] wrong values by lane quarter, counts[4..11] by output r, counts[12] total, counts[13] records written;
// rec: up to 256 x {unit, tile, lane, r, got.x, got.y, want.x, want.y}
extern "C" __global__ void compare(const float2 *__restrict__ out, const float2 *__restrict__ ref, unsigned long long n_per_unit,
                                   unsigned long long n_total, unsigned *counts, unsigned *rec) {
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n_total; i += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long j = i % n_per_unit;
    const float2 a = out[i], b = ref[j];
    if (__float_as_uint(a.x) != __float_as_uint(b.x) || __float_as_uint(a.y) != __float_as_uint(b.y)) {
      const int lane = (int)((j % 512) / 8), r = (int)(j % 8);
      atomicAdd(&counts[lane >> 4], 1u);
      atomicAdd(&counts[4 + r], 1u);
      atomicAdd(&counts[12], 1u);
      const unsigned k = atomicAdd(&counts[13], 1u);
      if (k < 256) {
        unsigned *p = rec + 8 * k;
        p[0] = (unsigned)(i / n_per_unit); p[1] = (unsigned)(j / 512); p[2] = lane; p[3] = r;
        p[4] = __float_as_uint(a.x); p[5] = __float_as_uint(a.y); p[6] = __float_as_uint(b.x); p[7] = __float_as_uint(b.y);
      }
    }
  }
}
