/*
 * fmd_kernels.hip - fused IQ -> PCM kernel for gfx950 (MI355X).
 *
 * Execution model: ONE WAVEFRONT IS ONE WORKER.  A worker owns one time chunk of
 * one stream and walks it tile by tile (FMDK_TILE rate_in samples = 8 x as many
 * IQ samples) through every stage of the reference's chain; the 64 lanes split
 * each tile (8 consecutive outputs per lane).  Workers never synchronise with
 * each other: there is no workgroup barrier in the tile loop, values cross lanes
 * through wave shuffles or through the worker's private slice of LDS (DS
 * operations of one wave execute in order).  A CU holds 8-12 such workers, so
 * while one waits on LDS / HBM another issues arithmetic.
 *
 * A stream's launch is cut into n_chunks chunks of whole tiles so that any
 * stream count fills the chip.  A chunk that does not start the launch replays
 * warm_tiles tiles before its first tile from zero state and discards their
 * output: all histories are finite (FIRs) or contract below fp32 resolution
 * (de-emphasis), so its first real sample sees the state a sequential run has.
 *
 * HBM traffic is the algorithmic minimum: u8 IQ is read once (each lane loads
 * the 176 bytes its 8 outputs need as 16-byte words, one tile ahead of use; the
 * 48-byte overlap between neighbouring lanes is served by L1), int16 PCM is
 * written once; decimated IQ stays in registers, discriminator / MPX filter
 * outputs / resampled frames live in the worker's LDS slice.
 *
 * Stages per tile (reference src/rtl_fm_player.c):
 *   A  u8 -> f32, j^n rotation, 32-tap /8 FIR        :195-239, :253-411
 *   B  polynomial-atan2 FM discriminator             :606-685
 *   Q  block-start overwrite quirk (stereo)          :534-598
 *   C  three 90-tap MPX FIRs + 38 kHz carrier        :533-568, :472-481
 *   D  rational resampler, second FIR at emit times  :570-598 (stereo), :500-532 (mono)
 *   F  de-emphasis, f32 -> s16, PCM store (when the frame buffer fills or a
 *      block ends)                                   :687-735
 *
 * Two arithmetic contracts (template parameter EX):
 *   exact: the reference's operation order with unfused multiply/add (this
 *          file is compiled with -ffp-contract=off) -> bit-identical PCM;
 *   fast:  same summation order with explicit fused multiply-adds and the
 *          u8 offset folded into the decimator taps -> PCM within +-1 LSB.
 * No MFMA: the path is int8/fp32 streaming work; the
 * bound that matters is fp32 VALU issue, so the hot loops keep the non-FMA
 * instruction count and the LDS traffic per FMA low (8 outputs per lane share
 * one pair-sum and one tap read; scalar taps for the decimator; interleaved
 * {L+R, L-R} history so the resampler reads 8-byte pairs).
 */
#include <hip/hip_runtime.h>

#include <cstddef>
#include <type_traits>

#include "fmd_internal.h"

namespace {

constexpr int TW = FMDK_TILE;          /* rate_in samples per tile (8 per lane) */
constexpr int WPB = FMDK_WAVES;        /* workers (waves) per workgroup */
constexpr int NT = 64 * WPB;
constexpr int CAPW = FMDK_FRAME_CAP;   /* pending resampler outputs per worker */
constexpr int DEEMPH_GROUP = 16;       /* frames per de-emphasis lane */
static_assert(TW == 512, "a tile is 64 lanes x 8 outputs");
static_assert(CAPW / DEEMPH_GROUP <= 64, "one flush must fit one wave");

constexpr float K_PI = 3.14159265f;    /* include/rtl_fm_player.h:40 */
constexpr float K_PI_2 = 1.5707963f;   /* :41 */
constexpr float K_PI_4 = 0.78539816f;  /* :42 */

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f8 __attribute__((ext_vector_type(8)));   /* 8 per-lane outputs: an SSA value, never an alloca */

/* history slots in front of each FIR tile: >= size - 1, and >= 92 for the
 * aligned window reads of the 90-tap stereo path */
template <int HALF> constexpr int hist_of() { return HALF == 45 ? 96 : (HALF == 64 ? 128 : 256); }

template <int HV, bool STEREO = true>
struct __attribute__((aligned(16))) WaveMem {
  float v[HV + TW];          /* discriminator output, HV history slots in front   */
  float2 ms[STEREO ? HV + TW : 2];   /* stereo: {L+R low-pass, (L-R band-pass) x carrier} */
  float fr[CAPW + CAPW / 32];/* resampler outputs waiting for the flush; one pad float per 32 (fidx):
                                flush lanes stride 16 frames x channels = 32 floats apart     */
  uint4 iq[8 * 64 + 4];      /* the tile's IQ, 16-byte word c = 8 * col + row stored at [64 * row + col]
                                (a lane's 11 window reads are then conflict-free), + 3 halo words */
  float de[4];               /* de-emphasis state: [0..1] current, [2..3] next    */
  float pp[4];               /* scratch for the generic (runtime-size) MPX path   */
  float head[8];             /* the launch's first three decimator outputs (I,Q)  */
  long long prof[12];        /* per-stage cycle sums (fmd_debug_taps.prof), [11] = last stamp */
};

template <int HV, bool STEREO>
struct __attribute__((aligned(16))) Smem {
  f4 tap_mpx[128];           /* {fm[k], fp[k], fs[k], 0}, zero beyond size/2 */
  WaveMem<HV, STEREO> w[WPB];
};

/* A zero the optimiser cannot see through: indexing the kernarg tap tables with
 * it keeps their scalar loads inside the stage that uses them (hoisted to the
 * top of the kernel they do not fit the SGPR file and spill to VGPR lanes). */
__device__ __forceinline__ int opaque_zero() {
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}
/* Compiler-only fence: nothing (loads, VALU) is scheduled across it, which bounds
 * how far the LDS reads of a loop run ahead of the arithmetic. */
__device__ __forceinline__ void sched_fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

/* ---- arithmetic helpers ------------------------------------------------ */

template <bool EX>
__device__ __forceinline__ float mac(float acc, float a, float b) {
  if constexpr (EX) {
    float p = a * b;       /* -ffp-contract=off keeps this unfused */
    return acc + p;
  } else {
    return __builtin_fmaf(a, b, acc);
  }
}

__device__ __forceinline__ float ubyte(uint32_t w, int i) {
  return (float)((w >> (8 * i)) & 0xffu);   /* v_cvt_f32_ubyteN */
}

/* (u - 127.5) / 128: exact in fp32, one fused op (reference table [0]) */
__device__ __forceinline__ float t0(float u) { return __builtin_fmaf(u, 0.0078125f, -0.99609375f); }

/* a / b: IEEE division in the exact kernels, v_rcp_f32 (1 ulp) and a multiply in the fast ones */
template <bool EX>
__device__ __forceinline__ float fdiv(float a, float b) {
  if constexpr (EX) return a / b;
  else return a * __builtin_amdgcn_rcpf(b);
}

/* src/rtl_fm_player.c:606-667 through the magnitude ratio (see oracle/fm_oracle.c) */
template <bool EX>
__device__ __forceinline__ float poly_atan2(float y, float x) {
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  const bool xmaj = ax >= ay;
  const float num = xmaj ? ay : ax, den = xmaj ? ax : ay;
  const float a = fdiv<EX>(num, den);
  float r0;
  if constexpr (EX) {
    r0 = a * (K_PI_4 - (a - 1.f) * (0.2447f + 0.0663f * a));
  } else {
    const float q = __builtin_fmaf(0.0663f, a, 0.2447f);
    r0 = a * __builtin_fmaf(-(a - 1.f), q, K_PI_4);
  }
  float r;
  if (x < 0.f) {
    if (y < 0.f) r = xmaj ? r0 - K_PI : -r0 - K_PI_2;
    else         r = xmaj ? -r0 + K_PI : K_PI_2 + r0;
  } else {
    if (y < 0.f) r = xmaj ? -r0 : r0 - K_PI_2;
    else         r = xmaj ? r0 : K_PI_2 - r0;
  }
  if (y == 0.f) r = (x < 0.f) ? K_PI : 0.f;
  if (x == 0.f) r = (y < 0.f) ? -K_PI_2 : ((y > 0.f) ? K_PI_2 : 0.f);
  return r;
}

/* src/rtl_fm_player.c:472-481 */
template <bool EX>
__device__ __forceinline__ float carrier38(float x, float y) {
  const float z = fdiv<EX>(y, x);
  float den;
  if constexpr (EX) den = 1.f + (z * z);
  else den = __builtin_fmaf(z, z, 1.f);
  const float c = fdiv<EX>(z + z, den);
  return (x == 0.f) ? 0.f : c;
}

template <bool EX>
__device__ __forceinline__ float carrier_of(float vp, float vq, float swf, float cwf) {
  const float x = vp * swf;
  float y;
  if constexpr (EX) y = vp * cwf - vq;
  else y = __builtin_fmaf(vp, cwf, -vq);
  return carrier38<EX>(x, y);
}

/* src/rtl_fm_player.c:711-735 */
__device__ __forceinline__ int16_t to_s16(float x, float coef) {
  const float t = x * coef;
  int r = (int)__builtin_rintf(t);
  if (t > 32767.0f) r = 32767;
  if (t < -32768.0f) r = -32768;
  return (int16_t)r;
}

/* compile-time loop: f(integral_constant<int, I>) for I in [B, E) -- indices stay
 * constant expressions, so register "arrays" never become private memory */
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
template <int I>
__device__ __forceinline__ float elem(const f4 &a, const f4 &b, const f4 &c, const f4 &d) {
  if constexpr (I < 4) return a[I & 3];
  else if constexpr (I < 8) return b[I & 3];
  else if constexpr (I < 12) return c[I & 3];
  else return d[I & 3];
}

/* element (idx & 7) of an 8-entry register array, idx uniform or per lane */
__device__ __forceinline__ float pick8(const f8 &a, int idx) {
  float r = a[0];
#pragma unroll
  for (int i = 1; i < 8; i++) r = ((idx & 7) == i) ? a[i] : r;
  return r;
}

/* index into the padded pending-frame buffer */
__device__ __forceinline__ int fidx(int i) { return i + (i >> 5); }

/* ---- discriminator history layout ---------------------------------------------
 * In the 90-tap stereo kernel the lanes read the v[] window as aligned 16-byte
 * words at a 32-byte lane stride, a 2-way bank conflict for ds_read_b128 (its
 * 16-lane groups see only the even 16-byte slots).  Flipping the lowest word-index
 * bit with bit 4 (word w lives at w ^ ((w >> 4) & 1)) makes every such read
 * conflict free at no cost in space.  SWZ = false keeps the layout linear. */
template <bool SWZ> __device__ __forceinline__ int vword(int w4) { return SWZ ? (w4 ^ ((w4 >> 4) & 1)) : w4; }
template <bool SWZ> __device__ __forceinline__ int vidx(int m) { return SWZ ? (m ^ ((m >> 4) & 4)) : m; }

/* ---- tile load: global -> LDS, asynchronous ------------------------------- */

/* rate_in sample n covers IQ bytes [16n, 16n+16) of the stream (one 16-byte
 * word); output m of the /8 FIR needs words m-3 .. m.  The tile's 512 words are
 * copied with global_load_lds (no registers, completes in the background): wave
 * instruction `row` moves word 8 * lane + row of the tile to LDS slot
 * 64 * row + lane; a ninth, 3-lane instruction fetches the three words in front
 * of the tile.  Word indices are clamped into the stream (clamped words only
 * feed outputs that are masked or patched). */
template <int HV, typename WM>
__device__ __forceinline__ void load_tile_async(WM &w, const uint4 *iq16, int n_tile, int n_total,
                                                int lane) {
#pragma unroll
  for (int row = 0; row < 8; row++) {
    int c = n_tile + 8 * lane + row;
    c = c >= n_total ? n_total - 1 : c;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(iq16 + c),
                                     (__attribute__((address_space(3))) void *)(&w.iq[64 * row]), 16, 0, 0);
  }
  if (lane < 3) {
    int c = n_tile - 3 + lane;
    c = c < 0 ? 0 : c;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(iq16 + c),
                                     (__attribute__((address_space(3))) void *)(&w.iq[512]), 16, 0, 0);
  }
}

/* The lane's 11 words (outputs 8 * lane .. 8 * lane + 7) from the LDS image. */
template <int HV, typename WM>
__device__ __forceinline__ void read_window(const WM &w, int lane, uint32_t (&d)[44]) {
#pragma unroll
  for (int i = 0; i < 11; i++) {
    /* word 8 * lane - 3 + i: rows 5..7 of the previous column, then rows 0..7 of column `lane` */
    const int slot = i < 3 ? (lane ? 64 * (5 + i) + lane - 1 : 512 + i) : 64 * (i - 3) + lane;
    const uint4 q = w.iq[slot];
    d[4 * i] = q.x; d[4 * i + 1] = q.y; d[4 * i + 2] = q.z; d[4 * i + 3] = q.w;
  }
}

/* ---- stage A: decimating IQ low-pass ----------------------------------- */

/* Rotation by j^p of window sample with phase p (src/rtl_fm_player.c:206-226):
 * which byte feeds the I / Q sum and with which sign. */
template <bool ROT> __device__ __forceinline__ constexpr int sel_i(int p) { return ROT ? (p & 1) : 0; }
template <bool ROT> __device__ __forceinline__ constexpr int sel_q(int p) { return ROT ? ((p & 1) ^ 1) : 1; }
template <bool ROT> __device__ __forceinline__ constexpr float sgn_i(int p) {
  return !ROT ? 1.f : ((p == 0 || p == 3) ? 1.f : -1.f);
}
template <bool ROT> __device__ __forceinline__ constexpr float sgn_q(int p) {
  return !ROT ? 1.f : ((p == 0 || p == 1) ? 1.f : -1.f);
}

/* Eight consecutive outputs from the lane's 88 IQ samples (sample j = bytes
 * 2j, 2j+1 of d[]); output r uses samples 8r .. 8r+31, phase = index mod 4. */
template <bool EX, bool ROT>
__device__ __forceinline__ void decimate8(const fmdk_params &P, const uint32_t (&d)[44], f8 &yi, f8 &yq) {
  if constexpr (EX) {
    const int z = opaque_zero();
#pragma unroll
    for (int r = 0; r < 8; r++) {
      /* sum_k (c[k] + c[31-k]) * fb[k], left to right (src/rtl_fm_player.c:371-403) */
      float ai = 0.f, aq = 0.f;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int ja = 8 * r + k, jb = 8 * r + 31 - k;
        const int pa = k & 3, pb = (31 - k) & 3;
        const float ia = sgn_i<ROT>(pa) * t0(ubyte(d[ja >> 1], 2 * (ja & 1) + sel_i<ROT>(pa)));
        const float ib = sgn_i<ROT>(pb) * t0(ubyte(d[jb >> 1], 2 * (jb & 1) + sel_i<ROT>(pb)));
        const float qa = sgn_q<ROT>(pa) * t0(ubyte(d[ja >> 1], 2 * (ja & 1) + sel_q<ROT>(pa)));
        const float qb = sgn_q<ROT>(pb) * t0(ubyte(d[jb >> 1], 2 * (jb & 1) + sel_q<ROT>(pb)));
        const float fbk = P.fb[k + z];
        const float pi = (ia + ib) * fbk;
        const float pq = (qa + qb) * fbk;
        ai = (k == 0) ? pi : ai + pi;
        aq = (k == 0) ? pq : aq + pq;
      }
      yi[r] = ai;
      yq[r] = aq;
    }
  } else {
    /* offset and 1/128 folded into the taps: y = c + sum_j s[j] g[min(j,31-j)] u[j],
     * j ascending, g = fb / 128 and s the rotation sign (a free operand modifier).
     * Sample-outer order: each byte is converted once and feeds the (up to four)
     * outputs whose window holds it; the 16 distinct taps are scalar operands. */
    const int z = opaque_zero();
    float g[16];
#pragma unroll
    for (int k = 0; k < 16; k++) g[k] = P.fbs[k + z];
#pragma unroll
    for (int r = 0; r < 8; r++) { yi[r] = P.c_i; yq[r] = P.c_q; }
#pragma unroll
    for (int sx = 0; sx < 88; sx++) {
      if (sx % 8 == 0) sched_fence();           /* keep conversions next to their FMAs (register pressure) */
      const float ub[2] = {ubyte(d[sx >> 1], 2 * (sx & 1)), ubyte(d[sx >> 1], 2 * (sx & 1) + 1)};   /* I, Q */
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int j = sx - 8 * r;
        if (j >= 0 && j < 32) {
          const int p = j & 3;
          const float gk = g[j < 16 ? j : 31 - j];
          yi[r] = __builtin_fmaf(sgn_i<ROT>(p) * gk, ub[sel_i<ROT>(p)], yi[r]);
          yq[r] = __builtin_fmaf(sgn_q<ROT>(p) * gk, ub[sel_q<ROT>(p)], yq[r]);
        }
      }
    }
  }
}

/* First three outputs of the launch: their window reaches into the carried
 * float history lowpass_tb (src/rtl_fm_player.c:261-363).  Computed once, before
 * the tile loop, by lanes 0..5 (output m = lane / 2, component = lane & 1) into
 * head[6]; the first tile's lane 0 then takes them instead of its own. */
template <bool ROT>
__device__ __forceinline__ void decimate_head(const fmdk_params &P, const uint8_t *raw, const float *tb,
                                              int lane, float *head) {
  if (lane < 6) {
    const int m = lane >> 1, comp = lane & 1;
    float acc = 0.f;
    for (int k = 0; k < 16; k++) {
      float pr[2];
      for (int e = 0; e < 2; e++) {
        const int j = e ? 31 - k : k;
        const int g = 8 * m - 24 + j;          /* IQ sample index from the start of the stream */
        float c;
        if (g < 0) {
          c = tb[2 * (24 + g) + comp];
        } else {
          const int p = j & 3;
          const int sel = comp ? sel_q<ROT>(p) : sel_i<ROT>(p);
          const float sg = comp ? sgn_q<ROT>(p) : sgn_i<ROT>(p);
          c = sg * t0((float)raw[2 * g + sel]);
        }
        pr[e] = c;
      }
      const float prod = (pr[0] + pr[1]) * P.fb[k];
      acc = (k == 0) ? prod : acc + prod;
    }
    head[lane] = acc;
  }
}

/* ---- stage B: discriminator --------------------------------------------- */

template <bool EX>
__device__ __forceinline__ float discriminate(float pr, float pj, float re, float im) {
  float cr, dt;
  if constexpr (EX) {
    cr = pr * im - pj * re;          /* pre_r * Q - pre_j * I */
    dt = re * pr + im * pj;          /* I * pre_r + Q * pre_j */
  } else {
    cr = __builtin_fmaf(pr, im, -(pj * re));
    dt = __builtin_fmaf(re, pr, im * pj);
  }
  return poly_atan2<EX>(cr, dt);
}

/* ---- stage C: MPX filters at rate_in (stereo) --------------------------- */

/* ms[m] = { sum fm[k] p[m,k],  (sum fs[k] p[m,k]) * carrier(vp[m], vp[m-1]) },
 * vp[m] = sum fp[k] p[m,k],  p[m,k] = v[m-89+k] + v[m-k]   (:538-566).
 * HALF == 45: every lane owns 8 consecutive outputs and walks the 45 taps in
 * 12 chunks of 4 (taps 45..47 are zero); a chunk needs 7 aligned 16-byte window
 * reads and 4 tap reads for 8 x 4 x (1 add + 3 FMA).  pp is the pilot output of
 * the sample before the tile (in), of the tile's last sample (out). */
template <bool EX, int HALF, int HV, typename WM>
__device__ __forceinline__ void mpx_tile(const fmdk_params &P, const f4 *tap_mpx, WM &w, int lane,
                                         int tm, float &pp) {
  if constexpr (HALF == 45) {
    constexpr int R = 8;
    const int m0 = R * lane;
    f8 am = 0.f, ap = 0.f, as = 0.f;
    if (m0 < tm) {
      const f4 *v4s = reinterpret_cast<const f4 *>(w.v);   /* x[i] = v[m0 - 92 + i], word-swizzled */
      /* chunk c (8 taps 8c .. 8c+7; 45..47 are zero): window words x[8c .. 8c+19]
       * and x[84-8c .. 99-8c]: nine 16-byte window reads + eight tap reads for
       * 8 outputs x 8 taps x (1 add + 3 FMA) */
      const int wbase = (HV + m0 - 92) >> 2;
#pragma unroll 1
      for (int c = 0; c < 6; c++) {           /* rolled: keeps the register footprint at one chunk */
        f4 lo[5], hi[4];
#pragma unroll
        for (int i = 0; i < 5; i++) lo[i] = v4s[vword<true>(wbase + 2 * c + i)];
#pragma unroll
        for (int i = 0; i < 4; i++) hi[i] = v4s[vword<true>(wbase + 21 - 2 * c + i)];
        static_for<0, 8>([&](auto kk_) {
          constexpr int kk = decltype(kk_)::value;
          const f4 t = tap_mpx[8 * c + kk];
          static_for<0, R>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            constexpr int il = r + kk + 3, ih = r + 8 - kk;    /* x[r+k+3] + x[r+92-k] */
            const float p = lo[il >> 2][il & 3] + hi[ih >> 2][ih & 3];
            am[r] = mac<EX>(am[r], p, t.x);
            ap[r] = mac<EX>(ap[r], p, t.y);
            as[r] = mac<EX>(as[r], p, t.z);
          });
        });
      }
    }
    const float up = __shfl_up(ap[R - 1], 1);
    const float pp_new = __shfl(pick8(ap, tm - 1), (tm - 1) >> 3);
    if (m0 < tm) {
      float prev = lane ? up : pp;
      const float swf = P.swf, cwf = P.cwf;
      f8 bs;
#pragma unroll
      for (int r = 0; r < R; r++) {
        bs[r] = as[r] * carrier_of<EX>(ap[r], prev, swf, cwf);
        prev = ap[r];
      }
      f4 *dst = reinterpret_cast<f4 *>(&w.ms[HV + m0]);
#pragma unroll
      for (int r = 0; r < R; r += 2) dst[r >> 1] = f4{am[r], bs[r], am[r + 1], bs[r + 1]};
    }
    pp = pp_new;
  } else {
    const int half = P.half, size = P.size;
    const float swf = P.swf, cwf = P.cwf;
    for (int m = lane; m < tm; m += 64) {
      const float *x = &w.v[HV + m - (size - 1)];
      float am = 0.f, ap = 0.f, as = 0.f, aq = 0.f;   /* aq: pilot output of sample m-1, same order */
      for (int k = 0; k < half; k++) {
        const float p = x[k] + x[size - 1 - k];
        const float pq = x[k - 1] + x[size - 2 - k];
        const f4 t = tap_mpx[k];
        am = mac<EX>(am, p, t.x);
        ap = mac<EX>(ap, p, t.y);
        as = mac<EX>(as, p, t.z);
        aq = mac<EX>(aq, pq, t.y);
      }
      if (m == 0) aq = pp;
      if (m == tm - 1) w.pp[0] = ap;
      w.ms[HV + m] = make_float2(am, as * carrier_of<EX>(ap, aq, swf, cwf));
    }
    pp = w.pp[0];
  }
}

/* Symmetric fm FIR over the `2*HALF` floats ending at newest (mono, :511-529). */
template <bool EX, int HALF>
__device__ __forceinline__ float fir_mono(const fmdk_params &P, const f4 *tap_mpx, const f4 (&tp)[16],
                                          const float *newest) {
  float acc = 0.f;
  if constexpr (HALF > 0) {
    /* taps come in registers (tp: fm[0..63]); the pair reads run a group of 16 ahead */
    constexpr int S = 2 * HALF, G = 16, NG = HALF / G;
    static_assert(HALF % G == 0, "tap groups");
    const float *x0 = newest - (S - 1);
    float xa[2 * G], xb[2 * G];
    auto load_group = [&](float (&x)[2 * G], int g) {
#pragma unroll
      for (int i = 0; i < G; i++) {
        x[2 * i] = x0[g * G + i];
        x[2 * i + 1] = x0[S - 1 - (g * G + i)];
      }
    };
    load_group(xa, 0);
    static_for<0, NG>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      float(&cur)[2 * G] = (g & 1) ? xb : xa;
      float(&nxt)[2 * G] = (g & 1) ? xa : xb;
      if constexpr (g + 1 < NG) load_group(nxt, g + 1);
      static_for<0, G>([&](auto i_) {
        constexpr int i = decltype(i_)::value, k = g * G + i;
        acc = mac<EX>(acc, cur[2 * i] + cur[2 * i + 1], tp[k >> 2][k & 3]);
      });
      sched_fence();
    });
  } else {
    const int size = P.size, half = P.half;
    const float *x = newest - (size - 1);
    for (int k = 0; k < half; k++) acc = mac<EX>(acc, x[k] + x[size - 1 - k], tap_mpx[k].x);
  }
  return acc;
}

/* The two stage-2 FIRs of the stereo path at one instant (:574-591). */
template <bool EX, int HALF>
__device__ __forceinline__ void fir_stereo(const fmdk_params &P, const f4 *tap_mpx, const f4 (&tp)[16],
                                           const float2 *newest, float &om, float &os) {
  om = 0.f; os = 0.f;
  if constexpr (HALF > 0) {
    /* taps come in registers (tp: fm[0..47], loaded once per tile); the 45 pair
     * reads are issued a group of 9 ahead of the arithmetic that consumes them */
    constexpr int S = 2 * HALF, G = 9, NG = HALF / G;
    static_assert(HALF % G == 0, "tap groups");
    const float2 *x0 = newest - (S - 1);
    float2 xa[2 * G], xb[2 * G];
    auto load_group = [&](float2 (&x)[2 * G], int g) {
#pragma unroll
      for (int i = 0; i < G; i++) {
        x[2 * i] = x0[g * G + i];
        x[2 * i + 1] = x0[S - 1 - (g * G + i)];
      }
    };
    load_group(xa, 0);
    static_for<0, NG>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      float2(&cur)[2 * G] = (g & 1) ? xb : xa;
      float2(&nxt)[2 * G] = (g & 1) ? xa : xb;
      if constexpr (g + 1 < NG) load_group(nxt, g + 1);
      static_for<0, G>([&](auto i_) {
        constexpr int i = decltype(i_)::value, k = g * G + i;
        const float t = tp[k >> 2][k & 3];
        om = mac<EX>(om, cur[2 * i].x + cur[2 * i + 1].x, t);
        os = mac<EX>(os, cur[2 * i].y + cur[2 * i + 1].y, t);
      });
      sched_fence();
    });
  } else {
    const int size = P.size, half = P.half;
    const float2 *x = newest - (size - 1);
    for (int k = 0; k < half; k++) {
      const float2 a = x[k], b = x[size - 1 - k];
      const float t = tap_mpx[k].x;
      om = mac<EX>(om, a.x + b.x, t);
      os = mac<EX>(os, a.y + b.y, t);
    }
  }
}

/* Block-start quirk (src/rtl_fm_player.c:534-598):
 * when the resampler emits on sample 0 of a block, the right-channel output is
 * stored over discriminator sample 1 before that sample is read. */
template <bool EX, int HV, bool SWZ, typename WM>
__device__ __forceinline__ void q1_patch(const fmdk_params &P, const f4 *tap_mpx, WM &w, int lane,
                                         float pp) {
  float f = 0.f;
  if (lane < 3) {
    const int size = P.size, half = P.half;
    const int x0 = HV - (size - 1);
    const float *tap = reinterpret_cast<const float *>(tap_mpx) + lane;
    for (int k = 0; k < half; k++)
      f = mac<EX>(f, w.v[vidx<SWZ>(x0 + k)] + w.v[vidx<SWZ>(x0 + size - 1 - k)], tap[4 * k]);
  }
  const float vp = __shfl(f, 1), vs = __shfl(f, 2);
  if (lane == 0) {
    w.ms[HV] = make_float2(f, vs * carrier_of<EX>(vp, pp, P.swf, P.cwf));
    float om, os;
    const f4 no_tp[16] = {};
    fir_stereo<EX, 0>(P, tap_mpx, no_tp, &w.ms[HV], om, os);
    w.v[vidx<SWZ>(HV + 1)] = om - os;
  }
}

/* ---- stage D: resampler outputs ------------------------------------------ */

/* local index of the q-th emit of this tile; acc_t = accumulator at tile start.
 * i = ceil(((q+1) * fast - acc_t) / slow) - 1, by a float estimate corrected
 * with exact integer checks. */
__device__ __forceinline__ int emit_index(uint32_t acc_t, int q, uint32_t slow, uint32_t fast,
                                          float inv_slow) {
  const uint32_t need = (uint32_t)(q + 1) * fast - acc_t;   /* > 0 */
  uint32_t e = (uint32_t)((float)need * inv_slow);
  if (e * slow < need) e++;
  else if (e > 0 && (e - 1) * slow >= need) e--;
  if (e * slow < need) e++;
  return (int)e - 1;
}

template <bool EX, int MODE, int HALF, int HV, typename WM>
__device__ __forceinline__ void resample_tile(const fmdk_params &P, const f4 *tap_mpx, WM &w,
                                              int lane, uint32_t acc_t, int nq, int pend) {
  const uint32_t slow = (uint32_t)P.slow, fast = (uint32_t)P.fast;
  const float inv_slow = 1.0f / (float)P.slow;
  const bool rs = P.resample != 0;
  f4 tp[16];                                  /* fm[0..63] for the unrolled FIRs */
  if constexpr (MODE != 0 && HALF > 0) {
    const float *tf = reinterpret_cast<const float *>(tap_mpx);
#pragma unroll
    for (int i = 0; i < 16; i++) tp[i] = f4{tf[16 * i], tf[16 * i + 4], tf[16 * i + 8], tf[16 * i + 12]};
  }
  for (int q = lane; q < nq; q += 64) {
    const int i = rs ? emit_index(acc_t, q, slow, fast, inv_slow) : q;
    if constexpr (MODE == 2) {
      float om, os;
      fir_stereo<EX, HALF>(P, tap_mpx, tp, &w.ms[HV + i], om, os);
      w.fr[fidx(pend + 2 * q)] = om + os;          /* :595 */
      w.fr[fidx(pend + 2 * q + 1)] = om - os;      /* :596 */
    } else if constexpr (MODE == 1) {
      w.fr[fidx(pend + q)] = fir_mono<EX, HALF>(P, tap_mpx, tp, &w.v[HV + i]);
    } else {
      w.fr[fidx(pend + q)] = w.v[HV + i];
    }
  }
}

/* ---- stage F: de-emphasis + s16 + store ----------------------------------- */

/* De-emphasis is a first-order recurrence (:687-709).  Each lane produces
 * DEEMPH_GROUP consecutive frames of one channel; a lane whose segment does not
 * start at the first pending frame restarts the recurrence P.warm frames early
 * from zero (lambda^warm < 1e-12, below fp32 resolution), the others continue
 * from the carried state, so the result equals the sequential evaluation. */
template <bool EX, int CH, int HV, typename WM>
__device__ __forceinline__ void flush_frames(const fmdk_params &P, WM &w, int lane, int pend,
                                             int16_t *pcm_out, float *mpx_dbg, bool store) {
  const int frames = pend / CH;
  const float coef = P.coef;
  if (mpx_dbg) {
    for (int i = lane; i < pend; i += 64) mpx_dbg[i] = w.fr[fidx(i)];
  }
  if (P.deemph) {
    const int groups = (frames + DEEMPH_GROUP - 1) / DEEMPH_GROUP;
    const float lam = P.lambda;
    const int warm = P.warm;
    float ylast = 0.f;
    bool have_last = false;
    if (lane < groups * CH) {
      const int g = lane / CH, c = lane % CH;
      const int f_out = g * DEEMPH_GROUP;
      const int f_end = min(f_out + DEEMPH_GROUP, frames);
      int fb = f_out - warm;                  /* warm is a multiple of 16 (host), so is f_out */
      float y = 0.f;
      if (fb <= 0) { fb = 0; y = w.de[c]; }
      /* blocks of 16 steps: the 16 inputs do not depend on the recurrence, so they
       * are read together (one LDS latency per block instead of one per step) */
      for (; fb < f_out; fb += 16) {          /* warm-up blocks: whole, nothing stored */
        f8 xa, xb;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          xa[j] = w.fr[fidx((fb + j) * CH + c)];
          xb[j] = w.fr[fidx((fb + 8 + j) * CH + c)];
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const float x = j < 8 ? xa[j & 7] : xb[j & 7];
          const float t = y - x;
          if constexpr (EX) y = x + lam * t;
          else y = __builtin_fmaf(lam, t, x);
        }
      }
      {                                        /* the lane's own 16 frames (the last group may be short) */
        f8 xa, xb;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          xa[j] = w.fr[fidx(min(f_out + j, frames - 1) * CH + c)];
          xb[j] = w.fr[fidx(min(f_out + 8 + j, frames - 1) * CH + c)];
        }
        int16_t *o = pcm_out + f_out * CH + c;
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const float x = j < 8 ? xa[j & 7] : xb[j & 7];
          const float t = y - x;
          float yn;
          if constexpr (EX) yn = x + lam * t;
          else yn = __builtin_fmaf(lam, t, x);
          const bool live = f_out + j < f_end;
          y = live ? yn : y;
          if (store && live) o[j * CH] = to_s16(yn, coef);
        }
      }
      ylast = y;
      have_last = (f_end == frames);
    }
    if (have_last) w.de[lane % CH] = ylast;      /* all lanes have read de[] above (same wave) */
  } else if (store) {
    for (int i = lane; i < pend; i += 64) pcm_out[i] = to_s16(w.fr[fidx(i)], coef);
  }
}

/* ---- carried state in HBM ------------------------------------------------- */

struct DevState {   /* == fmd_stream_state */
  float tb[48];
  float pre_r, pre_j, pp, de_l, de_r;
  int32_t acc;
  int32_t reserved[2];
  float br[256], bm[256], bs[256];
};
static_assert(sizeof(DevState) == sizeof(fmd_stream_state), "state layout");

/* ---- the fused kernel ----------------------------------------------------- */

template <bool EX, int MODE, int HALF>
__global__ __launch_bounds__(NT, 2) void fmd_fused_kernel(const fmdk_params P, const uint8_t *__restrict__ iq_all,
                                                      int16_t *__restrict__ pcm_all,
                                                      int32_t *__restrict__ lens_all,
                                                      const DevState *__restrict__ state_in_all,
                                                      DevState *__restrict__ state_out_all, float *dbg_y,
                                                      float *dbg_v, float *dbg_mpx, long long *dbg_prof) {
  constexpr int HV = hist_of<HALF>();
  constexpr int CH = (MODE == 2) ? 2 : 1;
  constexpr bool SWZ = (MODE == 2 && HALF == 45);   /* word-swizzled discriminator history */
  __shared__ Smem<HV, MODE == 2> sm;
  for (int i = threadIdx.x; i < 128; i += NT) sm.tap_mpx[i] = f4{P.fm[i], P.fp[i], P.fs[i], 0.f};
  __syncthreads();                                /* the only workgroup barrier */

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = P.n_chunks;
  const int unit = blockIdx.x * WPB + wave;
  if (unit >= P.n_streams * K) return;
  const int stream = unit / K, chunk = unit - stream * K;
  WaveMem<HV, MODE == 2> &w = sm.w[wave];
  const f4 *tap_mpx = sm.tap_mpx;

  /* optional per-stage cycle accounting (fmd_debug_taps.prof): lane 0 keeps the
   * sums in the worker's LDS slice (registers would be live across every stage) */
  if (dbg_prof && lane == 0) {
    for (int i = 0; i < 11; i++) w.prof[i] = 0;
    w.prof[11] = w.prof[10] = clock64();
  }
#define FMD_STAMP(i)                                   \
  if (dbg_prof && lane == 0) {                         \
    const long long now_ = clock64();                  \
    w.prof[i] += now_ - w.prof[11];                    \
    w.prof[11] = now_;                                 \
  }

  const int M = P.block_len >> 4;                 /* rate_in samples per block */
  const int tpb = (M + TW - 1) / TW;              /* tiles per block */
  const int T = tpb * P.n_blocks;                 /* tiles per stream */
  const int n_total = M * P.n_blocks;             /* rate_in samples (= 16-byte IQ words) per stream */
  const uint32_t slow = (uint32_t)P.slow, fast = (uint32_t)P.fast;
  const uint8_t *iq_stream = iq_all + (size_t)stream * P.n_blocks * P.block_len;
  const uint4 *iq16 = reinterpret_cast<const uint4 *>(iq_stream);
  const DevState *st_in = state_in_all + stream;
  const int size = P.size;

  /* this worker's tiles; a chunk > 0 starts warm_tiles tiles early and discards
   * what those produce */
  const int t_lo = (int)((long long)chunk * T / K), t_hi = (int)((long long)(chunk + 1) * T / K);
  const int g_first = chunk > 0 ? t_lo - P.warm_tiles : 0;

  /* carried state (chunk 0) or zero state (replaying chunks) */
  const bool carried = (chunk == 0);
  for (int i = lane; i < size; i += 64) {
    w.v[vidx<SWZ>(HV - size + i)] = carried ? st_in->br[i] : 0.f;
    if constexpr (MODE == 2)
      w.ms[HV - size + i] = carried ? make_float2(st_in->bm[i], st_in->bs[i]) : make_float2(0.f, 0.f);
  }
  if (lane == 0) {
    w.de[0] = carried ? st_in->de_l : 0.f;
    w.de[1] = carried ? st_in->de_r : 0.f;
  }
  float ycr = carried ? st_in->pre_r : 0.f, ycj = carried ? st_in->pre_j : 0.f;   /* last decimated sample */
  float pp = carried ? st_in->pp : 0.f;                                          /* last pilot output */
  uint32_t acc = (uint32_t)st_in->acc;
  {
    const int b0 = g_first / tpb;
    const long long n0 = (long long)b0 * M + (long long)(g_first - b0 * tpb) * TW;
    if (P.resample) acc = (uint32_t)(((unsigned long long)acc + (unsigned long long)n0 * slow) % fast);
  }

  if (g_first < t_hi) {
    const int b0 = g_first / tpb;
    load_tile_async<HV>(w, iq16, b0 * M + (g_first - b0 * tpb) * TW, n_total, lane);
  }
  if (chunk == 0) {
    if (P.offset_tuning) decimate_head<false>(P, iq_stream, st_in->tb, lane, w.head);
    else decimate_head<true>(P, iq_stream, st_in->tb, lane, w.head);
  }
  FMD_STAMP(9)

  int pend = 0, pcm_off = 0, tm_last = 0, n_last = 0;
  bool q1 = false;
  for (int g = g_first; g < t_hi; g++) {
    const int b = g / tpb, off = (g - b * tpb) * TW;
    const int tm = min(TW, M - off);
    const int n_tile = b * M + off;               /* stream sample index of the tile's first sample */
    const bool discard = g < t_lo;
    const size_t slot = (size_t)stream * P.n_blocks + b;
    int16_t *pcm_blk = pcm_all + slot * P.pcm_stride;
    float *mpx_blk = (dbg_mpx && !discard) ? dbg_mpx + slot * M : nullptr;
    const int m0 = 8 * lane;

    if (off == 0) {
      pend = 0; pcm_off = 0;
      q1 = (MODE == 2) && P.resample && (acc + slow >= fast);
    } else if (g == t_lo && chunk > 0) {
      /* first real tile in the middle of a block: settle the de-emphasis state on
       * what the replay produced, then continue the block's PCM where a
       * sequential run would be: CH x (emits before `off` in this block) */
      flush_frames<EX, CH, HV>(P, w, lane, pend, pcm_blk, nullptr, false);
      pend = 0;
      unsigned long long a0 = (unsigned long long)st_in->acc;
      if (P.resample) {
        a0 = (a0 + (unsigned long long)b * M * slow) % fast;
        pcm_off = CH * (int)((a0 + (unsigned long long)off * slow) / fast);
      } else {
        pcm_off = CH * off;
      }
    }

    /* ---- A: /8 low-pass, 8 outputs per lane, straight from registers ---- */
    f8 yi, yq;
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* the tile's LDS image has landed */
      uint32_t d[44];
      read_window<HV>(w, lane, d);
      if (P.offset_tuning) {
        /* make the words opaque on this side: otherwise the byte conversions, common
         * to both sides, are hoisted above the branch and all 176 results are live at once */
#pragma unroll
        for (int i = 0; i < 44; i++) asm volatile("" : "+v"(d[i]));
        decimate8<EX, false>(P, d, yi, yq);
      } else {
        decimate8<EX, true>(P, d, yi, yq);
      }
    }
    if (n_tile == 0 && chunk == 0 && lane == 0) {
#pragma unroll
      for (int m = 0; m < 3; m++) { yi[m] = w.head[2 * m]; yq[m] = w.head[2 * m + 1]; }
    }
    /* next tile's IQ: in flight while the rest of this tile is computed */
    if (g + 1 < t_hi) {
      const int b2 = (g + 1) / tpb;
      load_tile_async<HV>(w, iq16, b2 * M + ((g + 1) - b2 * tpb) * TW, n_total, lane);
    }
    FMD_STAMP(1)
    if (dbg_y && !discard) {
      float2 *o = reinterpret_cast<float2 *>(dbg_y) + slot * M + off + m0;
#pragma unroll
      for (int r = 0; r < 8; r++)
        if (m0 + r < tm) o[r] = make_float2(yi[r], yq[r]);
    }

    /* ---- B: discriminator; the sample before the lane's first comes by shuffle ---- */
    {
      float pr = __shfl_up(yi[7], 1), pj = __shfl_up(yq[7], 1);
      if (lane == 0) { pr = ycr; pj = ycj; }
      f8 v;
#pragma unroll
      for (int r = 0; r < 8; r++) {
        v[r] = discriminate<EX>(pr, pj, yi[r], yq[r]);
        pr = yi[r]; pj = yq[r];
      }
      ycr = __shfl(pick8(yi, tm - 1), (tm - 1) >> 3);
      ycj = __shfl(pick8(yq, tm - 1), (tm - 1) >> 3);
      f4 *dst = reinterpret_cast<f4 *>(w.v);
      dst[vword<SWZ>((HV + m0) >> 2)] = f4{v[0], v[1], v[2], v[3]};
      dst[vword<SWZ>(((HV + m0) >> 2) + 1)] = f4{v[4], v[5], v[6], v[7]};
      if (dbg_v && !discard) {
        float *o = dbg_v + slot * M + off + m0;
#pragma unroll
        for (int r = 0; r < 8; r++)
          if (m0 + r < tm) o[r] = v[r];
      }
    }
    FMD_STAMP(2)

    /* ---- Q + C: stereo MPX filters ---- */
    if constexpr (MODE == 2) {
      if (q1 && off == 0 && tm > 1) q1_patch<EX, HV, SWZ>(P, tap_mpx, w, lane, pp);
      FMD_STAMP(3)
      mpx_tile<EX, HALF, HV>(P, tap_mpx, w, lane, tm, pp);
      FMD_STAMP(4)
    }

    /* ---- D: resampler outputs of this tile ---- */
    int nq;
    if (P.resample) nq = (int)(((unsigned long long)acc + (unsigned long long)tm * slow) / fast);
    else nq = tm;
    if (pend + nq * CH > CAPW) {
      flush_frames<EX, CH, HV>(P, w, lane, pend, pcm_blk + pcm_off, mpx_blk ? mpx_blk + pcm_off : nullptr,
                               !discard);
      pcm_off += pend;
      pend = 0;
      FMD_STAMP(8)
    }
    resample_tile<EX, MODE, HALF, HV>(P, tap_mpx, w, lane, acc, nq, pend);
    pend += nq * CH;
    if (P.resample) acc = (uint32_t)(((unsigned long long)acc + (unsigned long long)tm * slow) % fast);
    FMD_STAMP(6)

    /* ---- roll the FIR histories to the front of their buffers ---- */
    {
      constexpr int NR = (HV + 63) / 64;
      float rv[NR];
      float2 rm[NR];
#pragma unroll
      for (int i = 0; i < NR; i++) {             /* all reads first, then all writes (one wave: in order) */
        const int idx = lane + 64 * i;
        rv[i] = idx < HV ? w.v[vidx<SWZ>(tm + idx)] : 0.f;
        if constexpr (MODE == 2) rm[i] = idx < HV ? w.ms[tm + idx] : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < NR; i++) {
        const int idx = lane + 64 * i;
        if (idx < HV) {
          w.v[vidx<SWZ>(idx)] = rv[i];
          if constexpr (MODE == 2) w.ms[idx] = rm[i];
        }
      }
    }
    FMD_STAMP(7)

    /* ---- F: end of block -> PCM ---- */
    if (off + tm == M) {
      flush_frames<EX, CH, HV>(P, w, lane, pend, pcm_blk + pcm_off, mpx_blk ? mpx_blk + pcm_off : nullptr,
                               !discard);
      if (lane == 0 && !discard) lens_all[slot] = pcm_off + pend;
      pend = 0;
      FMD_STAMP(8)
    }
    tm_last = tm;
    n_last = n_tile;
    /* a chunk that ends in the middle of a block hands its pending frames over now */
    if (g + 1 == t_hi && pend > 0) {
      flush_frames<EX, CH, HV>(P, w, lane, pend, pcm_blk + pcm_off, mpx_blk ? mpx_blk + pcm_off : nullptr,
                               !discard);
      pend = 0;
    }
  }

  /* carried state -> HBM (the chunk that ends the launch) */
  if (chunk == K - 1 && g_first < t_hi) {
    DevState *st = state_out_all + stream;
    /* a launch may end in the middle of nothing: blocks are whole, so pend == 0 here */
    /* lowpass_tb: the last 24 complex samples, rotated, as floats (:366) */
    if (lane < 48) {
      const uint8_t *raw = iq_stream + ((size_t)(n_last + tm_last) * 16 - 48);
      const int j = lane >> 1, comp = lane & 1, p = j & 3;   /* 24 samples: phase = j mod 4 */
      int sel; float sg;
      if (P.offset_tuning) { sel = comp; sg = 1.f; }
      else {
        sel = comp ? sel_q<true>(p) : sel_i<true>(p);
        sg = comp ? sgn_q<true>(p) : sgn_i<true>(p);
      }
      st->tb[lane] = sg * t0((float)raw[2 * j + sel]);
    }
    for (int i = lane; i < size; i += 64) {
      st->br[i] = w.v[vidx<SWZ>(HV - size + i)];
      if constexpr (MODE == 2) {
        const float2 m = w.ms[HV - size + i];
        st->bm[i] = m.x;
        st->bs[i] = m.y;
      }
    }
    if (lane == 0) {
      st->pre_r = ycr;
      st->pre_j = ycj;
      st->pp = (MODE == 2) ? pp : st_in->pp;
      st->de_l = w.de[0];
      st->de_r = w.de[1];
      st->acc = (int32_t)acc;
    }
  }
  FMD_STAMP(9)
  if (dbg_prof && lane == 0) {
    long long *o = dbg_prof + 16 * (size_t)unit;
    for (int i = 0; i < 10; i++) o[i] = w.prof[i];
    o[15] = w.prof[11] - w.prof[10];
  }
#undef FMD_STAMP
}

template <bool EX, int MODE, int HALF>
int launch_one(const fmdk_params *p, int n_streams, const void *iq, void *pcm, void *lens,
               const void *state_in, void *state_out, const fmd_debug_taps *dbg, hipStream_t stream) {
  const int units = n_streams * p->n_chunks;
  hipLaunchKernelGGL((fmd_fused_kernel<EX, MODE, HALF>), dim3((units + WPB - 1) / WPB), dim3(NT), 0, stream,
                     *p, static_cast<const uint8_t *>(iq), static_cast<int16_t *>(pcm),
                     static_cast<int32_t *>(lens), static_cast<const DevState *>(state_in),
                     static_cast<DevState *>(state_out),
                     dbg ? static_cast<float *>(dbg->y) : nullptr,
                     dbg ? static_cast<float *>(dbg->v) : nullptr,
                     dbg ? static_cast<float *>(dbg->mpx) : nullptr,
                     dbg ? static_cast<long long *>(dbg->prof) : nullptr);
  return (int)hipGetLastError();
}

template <bool EX>
int launch_math(const fmdk_params *p, int n_streams, const void *iq, void *pcm, void *lens,
                const void *state_in, void *state_out, const fmd_debug_taps *dbg, hipStream_t stream) {
  /* rate_out2 <= 0: full_demod skips lp_real_f32 altogether (src/rtl_fm_player.c:781) */
  if (!p->resample) return launch_one<EX, 0, 0>(p, n_streams, iq, pcm, lens, state_in, state_out, dbg, stream);
  if (p->mode == 2) {
    if (p->half == 45) return launch_one<EX, 2, 45>(p, n_streams, iq, pcm, lens, state_in, state_out, dbg, stream);
    return launch_one<EX, 2, 0>(p, n_streams, iq, pcm, lens, state_in, state_out, dbg, stream);
  }
  if (p->mode == 1) {
    if (p->half == 64) return launch_one<EX, 1, 64>(p, n_streams, iq, pcm, lens, state_in, state_out, dbg, stream);
    return launch_one<EX, 1, 0>(p, n_streams, iq, pcm, lens, state_in, state_out, dbg, stream);
  }
  return launch_one<EX, 0, 0>(p, n_streams, iq, pcm, lens, state_in, state_out, dbg, stream);
}

}  // namespace

extern "C" int fmdk_launch(const fmdk_params *p, int math, int n_streams, const void *d_iq, void *d_pcm,
                           void *d_lens, const void *d_state_in, void *d_state_out,
                           const fmd_debug_taps *dbg, void *hip_stream) {
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (math == FMD_MATH_EXACT)
    return launch_math<true>(p, n_streams, d_iq, d_pcm, d_lens, d_state_in, d_state_out, dbg, st);
  return launch_math<false>(p, n_streams, d_iq, d_pcm, d_lens, d_state_in, d_state_out, dbg, st);
}

/* Tiles a replaying chunk must walk before its first real tile.  FIR memories:
 * 24 IQ + 1 (discriminator) + 2 x (size - 1) rate_in samples; the de-emphasis
 * restart needs (warm + group) frames = that many x fast / slow rate_in samples.
 * The last tile of a block may be short, so count tiles against the worst case. */
extern "C" int fmdk_warm_tiles(const fmdk_params *p) {
  long long need = 8 + 2LL * p->size;
  if (p->deemph) {
    if (p->warm >= (1 << 20)) return 0;          /* non-contracting recurrence: never split */
    need += ((long long)p->warm + DEEMPH_GROUP) * (p->resample ? (p->fast + p->slow - 1) / p->slow : 1);
  }
  const long long M = p->block_len >> 4, tpb = (M + TW - 1) / TW;
  const long long full = (need + TW - 1) / TW;
  if (M % TW == 0) return (int)full;
  if (full + 1 <= tpb) return (int)(full + 1);     /* at most one short tile inside the replay */
  return (int)(((need + M - 1) / M + 1) * tpb);    /* whole blocks */
}

extern "C" int fmdk_tile(void) { return TW; }

extern "C" const char *fmdk_kernel_name(const fmdk_params *p, int math) {
  (void)p;
  (void)math;
  return "fmd_fused_kernel";
}

extern "C" int fmdk_lds_bytes(void) { return (int)sizeof(Smem<96, true>); }
