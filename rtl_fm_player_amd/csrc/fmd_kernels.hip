/*
 * fmd_kernels.hip - fused IQ -> PCM kernel for gfx950 (MI355X).
 *
 * A workgroup owns one time chunk of one stream and walks it in time order, so
 * every piece of carried state (FIR histories, pilot sample, resampler
 * accumulator, de-emphasis) is handed from tile to tile through LDS exactly as
 * the reference hands it from call to call.  A stream's launch is cut into
 * n_chunks such chunks (whole blocks) so that two workgroups fit per CU for any
 * stream count; a chunk that does not start the launch replays warm_tiles tiles
 * before its first block from zero state and discards their output: all
 * histories are finite (FIRs) or contract below fp32 resolution (de-emphasis),
 * so its first real sample sees the same state as a sequential run.  HBM traffic is the algorithmic
 * minimum: the u8 IQ is read once (16 B per lane, straight into LDS with
 * global_load_lds, one sub-tile ahead of the arithmetic), the int16 PCM is
 * written once; every intermediate (decimated IQ, discriminator output, the
 * MPX filter outputs, resampled frames) lives in LDS.
 *
 * Stages (reference src/rtl_fm_player.c):
 *   per sub-tile of FMDK_SUB rate_in samples (8 x as many IQ samples):
 *     A  u8 -> f32, j^n rotation, 32-tap /8 FIR        :195-239, :253-411
 *     B  polynomial-atan2 FM discriminator             :606-685
 *   per tile of FMDK_TILE rate_in samples:
 *     Q  block-start overwrite quirk (stereo)          :534-598
 *     C  three 90-tap MPX FIRs + 38 kHz carrier        :533-568, :472-481
 *     D  rational resampler, second FIR at emit times  :570-598 (stereo), :500-532 (mono)
 *   per block:
 *     F  de-emphasis, f32 -> s16, PCM store            :687-735
 *
 * Two arithmetic contracts (template parameter EX):
 *   exact: the reference's operation order with unfused multiply/add (this
 *          file is compiled with -ffp-contract=off) -> bit-identical PCM;
 *   fast:  same summation order with explicit fused multiply-adds and the
 *          u8 offset folded into the decimator taps -> PCM within +-1 LSB.
 * No MFMA: the path is int8/fp32 streaming work; the
 * bound that matters is fp32 VALU issue, so the hot loops are written to keep
 * the non-FMA instruction count and the LDS traffic per FMA low:
 *   - stage C gives each lane 8 consecutive outputs (register blocking): the
 *     three filters share one pair-sum, one tap read serves 8 outputs;
 *   - decimator / resampler taps are scalar (kernarg) operands, MPX taps are
 *     wave-uniform LDS reads;
 *   - {L+R, L-R} histories are interleaved so the resampler reads 8-byte pairs.
 */
#include <hip/hip_runtime.h>

#include "fmd_internal.h"

namespace {

constexpr int TM = FMDK_TILE;
constexpr int SUB = FMDK_SUB;
constexpr int HV = FMDK_HIST;
constexpr int CAPF = FMDK_FRAME_CAP;
constexpr int NT = FMDK_THREADS;
constexpr int NW = NT / 64;
constexpr int DEEMPH_GROUP = 16;   /* frames per de-emphasis lane */
static_assert(TM % SUB == 0 && TM == 8 * NT && SUB == 2 * NT, "tiling assumptions of stages A and C");

constexpr float K_PI = 3.14159265f;    /* include/rtl_fm_player.h:40 */
constexpr float K_PI_2 = 1.5707963f;   /* :41 */
constexpr float K_PI_4 = 0.78539816f;  /* :42 */

typedef float f4 __attribute__((ext_vector_type(4)));

struct __attribute__((aligned(16))) Smem {
  uint4 iq[2][SUB + 4];      /* double buffer: 48 halo bytes, then 16 bytes per rate_in sample */
  float2 y[SUB + 2];         /* y[0] = last y of the previous sub-tile, y[1+m]  */
  float v[HV + TM];          /* discriminator, HV history slots in front        */
  float2 ms[HV + TM];        /* stereo: {L+R low-pass, (L-R band-pass) x carrier} */
  float fr[CAPF];            /* resampler outputs waiting for the flush         */
  f4 tap_mpx[128];           /* {fm[k], fp[k], fs[k], 0}, zero beyond size/2    */
  float edge[NW + 8];        /* pilot output of each wave's last lane           */
  float de[4];               /* de-emphasis state: [0..1] current, [2..3] next  */
  float pp;                  /* pilot band-pass output of the previous sample   */
  float pp_next;
};

__shared__ Smem g_s;

/* LDS-only workgroup barrier: does not drain outstanding global_load_lds /
 * global stores (a __syncthreads() would add s_waitcnt vmcnt(0)). */
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
/* Barrier that also waits for this wave's global_load_lds writes to land. */
__device__ __forceinline__ void full_barrier() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
/* A zero the optimiser cannot see through: indexing the kernarg tap tables with
 * it keeps their scalar loads inside the stage that uses them (hoisted out of
 * the tile loop they no longer fit the SGPR file and spill to VGPR lanes). */
__device__ __forceinline__ int opaque_zero() {
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}
/* Compiler-only fence: nothing (loads, VALU) is scheduled across it, which bounds
 * how far the LDS reads of an unrolled loop run ahead of the arithmetic. */
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

/* src/rtl_fm_player.c:606-667 through the magnitude ratio (see oracle/fm_oracle.c) */
template <bool EX>
__device__ __forceinline__ float poly_atan2(float y, float x) {
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  const bool xmaj = ax >= ay;
  const float num = xmaj ? ay : ax, den = xmaj ? ax : ay;
  const float a = num / den;                                /* IEEE divide */
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
__device__ __forceinline__ float carrier38(float x, float y) {
  const float z = y / x;
  const float c = (z + z) / (1.f + (z * z));
  return (x == 0.f) ? 0.f : c;
}

template <bool EX>
__device__ __forceinline__ float carrier_of(float vp, float vq, float swf, float cwf) {
  const float x = vp * swf;
  float y;
  if constexpr (EX) y = vp * cwf - vq;
  else y = __builtin_fmaf(vp, cwf, -vq);
  return carrier38(x, y);
}

/* src/rtl_fm_player.c:711-735 */
__device__ __forceinline__ int16_t to_s16(float x, float coef) {
  const float t = x * coef;
  int r = (int)__builtin_rintf(t);
  if (t > 32767.0f) r = 32767;
  if (t < -32768.0f) r = -32768;
  return (int16_t)r;
}

/* ---- sub-tile load: global -> LDS, asynchronous -------------------------- */

/* Copies 16-byte chunks [first, n16) of src into g_s.iq[buf].  Each wave
 * instruction moves 64 lanes x 16 B to a contiguous 1 KiB of LDS
 * (global_load_lds: wave-uniform LDS base + lane * 16). */
__device__ __forceinline__ void load_sub_async(const uint4 *src, int buf, int first, int n16) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int base = first + 64 * wave; base < n16; base += 64 * NW) {
    const int i = base + lane;
    if (i < n16) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i),
                                       (__attribute__((address_space(3))) void *)(&g_s.iq[buf][base]), 16, 0,
                                       0);
    }
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

/* Two consecutive outputs per lane: 80 raw bytes (5 x 16 B) from the LDS sub-tile. */
template <bool EX, bool ROT>
__device__ __forceinline__ void decimate_sub(const fmdk_params &P, int buf, int sm) {
  Smem &s = g_s;
  const int z = opaque_zero();
  for (int item = threadIdx.x; 2 * item < sm; item += NT) {
    uint32_t d[20];
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const uint4 q = s.iq[buf][2 * item + i];
      d[4 * i] = q.x; d[4 * i + 1] = q.y; d[4 * i + 2] = q.z; d[4 * i + 3] = q.w;
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      float ai, aq;
      if constexpr (EX) {
        /* sum_k (c[k] + c[31-k]) * fb[k], left to right (src/rtl_fm_player.c:371-403) */
        ai = 0.f; aq = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const int ja = 8 * r + k, jb = 8 * r + 31 - k;     /* sample index in the 40-sample span */
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
      } else {
        /* offset and 1/128 folded into signed taps: sum_j ts[j] * u[j] + c */
        ai = P.c_i; aq = P.c_q;
#pragma unroll
        for (int j = 0; j < 32; j++) {
          const int js = 8 * r + j, p = j & 3;
          ai = __builtin_fmaf(P.ts_i[j + z], ubyte(d[js >> 1], 2 * (js & 1) + sel_i<ROT>(p)), ai);
          aq = __builtin_fmaf(P.ts_q[j + z], ubyte(d[js >> 1], 2 * (js & 1) + sel_q<ROT>(p)), aq);
        }
      }
      const int m = 2 * item + r;
      if (m < sm) s.y[1 + m] = make_float2(ai, aq);
    }
  }
}

/* First three outputs of the first block of a launch: their window reaches
 * into the carried float history lowpass_tb (src/rtl_fm_player.c:261-363). */
template <bool ROT>
__device__ __forceinline__ void decimate_head(const fmdk_params &P, int buf, const float *tb, int sm) {
  Smem &s = g_s;
  const int lane = threadIdx.x;
  if (lane < 6 && (lane >> 1) < sm) {
    const int m = lane >> 1, comp = lane & 1;
    const uint8_t *raw = reinterpret_cast<const uint8_t *>(s.iq[buf]) + 48;
    float acc = 0.f;
    for (int k = 0; k < 16; k++) {
      float pr[2];
      for (int e = 0; e < 2; e++) {
        const int j = e ? 31 - k : k;
        const int g = 8 * m - 24 + j;          /* sample index within the block */
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
    float2 *yy = &s.y[1 + m];
    if (comp) yy->y = acc; else yy->x = acc;
  }
}

/* ---- stage B: discriminator --------------------------------------------- */

template <bool EX>
__device__ __forceinline__ void discriminate_sub(int v_off, int sm) {
  Smem &s = g_s;
  for (int m = threadIdx.x; m < sm; m += NT) {
    const float2 p = s.y[m], c = s.y[m + 1];
    float cr, dt;
    if constexpr (EX) {
      cr = p.x * c.y - p.y * c.x;          /* pre_r * Q - pre_j * I */
      dt = c.x * p.x + c.y * p.y;          /* I * pre_r + Q * pre_j */
    } else {
      cr = __builtin_fmaf(p.x, c.y, -(p.y * c.x));
      dt = __builtin_fmaf(c.x, p.x, c.y * p.y);
    }
    s.v[HV + v_off + m] = poly_atan2<EX>(cr, dt);
  }
}

/* ---- stage C: MPX filters at rate_in (stereo) --------------------------- */

/* ms[m] = { sum fm[k] p[m,k],  (sum fs[k] p[m,k]) * carrier(vp[m], vp[m-1]) },
 * vp[m] = sum fp[k] p[m,k],  p[m,k] = v[m-89+k] + v[m-k]   (:538-566).
 * HALF == 45: every lane owns 8 consecutive outputs and walks the 45 taps in
 * 12 chunks of 4 (taps 45..47 are zero); a chunk needs 7 aligned 16-byte window
 * reads and 4 tap reads for 8 x 4 x (1 add + 3 FMA).  Contains two workgroup
 * barriers (the previous lane's last pilot output comes through a shuffle, the
 * previous wave's through LDS). */
template <bool EX, int HALF>
__device__ __forceinline__ void mpx_tile(const fmdk_params &P, int tm) {
  Smem &s = g_s;
  if constexpr (HALF == 45) {
    constexpr int R = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = R * threadIdx.x;
    float am[R], ap[R], as[R];
#pragma unroll
    for (int r = 0; r < R; r++) { am[r] = 0.f; ap[r] = 0.f; as[r] = 0.f; }
    if (m0 < tm) {
      const f4 *v4 = reinterpret_cast<const f4 *>(s.v) + ((HV + m0 - 92) >> 2);   /* w[i] = v[m0 - 92 + i] */
      /* chunk c: window w[4c .. 4c+15] and w[88-4c .. 99-4c], taps 4c .. 4c+3.
       * Two register sets: chunk c+1 is read while chunk c is consumed. */
      f4 wa[11], wb[11];
      auto load_chunk = [&](f4 (&w)[11], int c) {
        w[0] = v4[c]; w[1] = v4[c + 1]; w[2] = v4[c + 2]; w[3] = v4[c + 3];
        w[4] = v4[22 - c]; w[5] = v4[23 - c]; w[6] = v4[24 - c];
        w[7] = s.tap_mpx[4 * c]; w[8] = s.tap_mpx[4 * c + 1];
        w[9] = s.tap_mpx[4 * c + 2]; w[10] = s.tap_mpx[4 * c + 3];
      };
      auto use_chunk = [&](const f4 (&w)[11]) {
        const float lo[16] = {w[0].x, w[0].y, w[0].z, w[0].w, w[1].x, w[1].y, w[1].z, w[1].w,
                              w[2].x, w[2].y, w[2].z, w[2].w, w[3].x, w[3].y, w[3].z, w[3].w};
        const float hi[12] = {w[4].x, w[4].y, w[4].z, w[4].w, w[5].x, w[5].y, w[5].z, w[5].w,
                              w[6].x, w[6].y, w[6].z, w[6].w};
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const f4 t = w[7 + kk];
#pragma unroll
          for (int r = 0; r < R; r++) {
            const float p = lo[r + kk + 3] + hi[r + 4 - kk];   /* w[r+k+3] + w[r+92-k] */
            am[r] = mac<EX>(am[r], p, t.x);
            ap[r] = mac<EX>(ap[r], p, t.y);
            as[r] = mac<EX>(as[r], p, t.z);
          }
        }
      };
      load_chunk(wa, 0);
#pragma unroll 1
      for (int c = 0; c < 12; c += 2) {       /* rolled: the schedule below is the schedule */
        load_chunk(wb, c + 1);
        use_chunk(wa);
        sched_fence();
        load_chunk(wa, c + 2 < 12 ? c + 2 : 0);   /* the last refill is a harmless re-read of chunk 0 */
        use_chunk(wb);
        sched_fence();
      }
#pragma unroll
      for (int r = 0; r < R; r++)
        if (m0 + r == tm - 1) s.pp_next = ap[r];
    }
    if (lane == 63) s.edge[wave + 1] = ap[R - 1];
    if (threadIdx.x == 0) s.edge[0] = s.pp;
    const float up = __shfl_up(ap[R - 1], 1);
    lds_barrier();
    if (m0 < tm) {
      float prev = lane ? up : s.edge[wave];
      const float swf = P.swf, cwf = P.cwf;
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (m0 + r < tm) s.ms[HV + m0 + r] = make_float2(am[r], as[r] * carrier_of<EX>(ap[r], prev, swf, cwf));
        prev = ap[r];
      }
    }
    lds_barrier();
    if (threadIdx.x == 0) s.pp = s.pp_next;
  } else {
    const int half = P.half, size = P.size;
    const float swf = P.swf, cwf = P.cwf;
    for (int m = threadIdx.x; m < tm; m += NT) {
      const float *w = &s.v[HV + m - (size - 1)];
      float am = 0.f, ap = 0.f, as = 0.f, aq = 0.f;   /* aq: pilot output of sample m-1, same order */
      for (int k = 0; k < half; k++) {
        const float p = w[k] + w[size - 1 - k];
        const float pq = w[k - 1] + w[size - 2 - k];
        const f4 t = s.tap_mpx[k];
        am = mac<EX>(am, p, t.x);
        ap = mac<EX>(ap, p, t.y);
        as = mac<EX>(as, p, t.z);
        aq = mac<EX>(aq, pq, t.y);
      }
      if (m == 0) aq = s.pp;
      if (m == tm - 1) s.pp_next = ap;
      s.ms[HV + m] = make_float2(am, as * carrier_of<EX>(ap, aq, swf, cwf));
    }
    lds_barrier();
    if (threadIdx.x == 0) s.pp = s.pp_next;
    lds_barrier();
  }
}

/* Symmetric fm FIR over the `2*HALF` floats ending at newest (mono, :511-529). */
template <bool EX, int HALF>
__device__ __forceinline__ float fir_mono(const fmdk_params &P, const float *newest, int z) {
  const Smem &s = g_s;
  float acc = 0.f;
  if constexpr (HALF > 0) {
    constexpr int S = 2 * HALF, G = 16, NG = (HALF + G - 1) / G;
    const float *w = newest - (S - 1);
    const f4 *tp = s.tap_mpx;
    (void)z;
    float xa[2 * G], xb[2 * G], ta[G], tb[G];
    auto load_group = [&](float (&x)[2 * G], float (&t)[G], int g) {
#pragma unroll
      for (int i = 0; i < G; i++) {
        const int k = g * G + i, kc = k < HALF ? k : HALF - 1;
        x[2 * i] = w[kc];
        x[2 * i + 1] = w[S - 1 - kc];
        t[i] = tp[k].x;
      }
    };
    auto use_group = [&](const float (&x)[2 * G], const float (&t)[G]) {
#pragma unroll
      for (int i = 0; i < G; i++) acc = mac<EX>(acc, x[2 * i] + x[2 * i + 1], t[i]);
    };
    static_assert(NG % 2 == 0, "group pairs");
    load_group(xa, ta, 0);
#pragma unroll 1
    for (int g = 0; g < NG; g += 2) {
      load_group(xb, tb, g + 1);
      use_group(xa, ta);
      sched_fence();
      load_group(xa, ta, g + 2 < NG ? g + 2 : 0);
      use_group(xb, tb);
      sched_fence();
    }
  } else {
    const int size = P.size, half = P.half;
    const float *w = newest - (size - 1);
    for (int k = 0; k < half; k++) acc = mac<EX>(acc, w[k] + w[size - 1 - k], s.tap_mpx[k].x);
  }
  return acc;
}

/* The two stage-2 FIRs of the stereo path at one instant (:574-591). */
template <bool EX, int HALF>
__device__ __forceinline__ void fir_stereo(const fmdk_params &P, const float2 *newest, int z, float &om,
                                           float &os) {
  const Smem &s = g_s;
  om = 0.f; os = 0.f;
  if constexpr (HALF > 0) {
    constexpr int S = 2 * HALF, G = 8, NG = (HALF + G - 1) / G;     /* groups of 8 taps, last one partial */
    const float2 *w = newest - (S - 1);
    const f4 *tp = s.tap_mpx;
    (void)z;
    float2 xa[2 * G], xb[2 * G];
    float ta[G], tb[G];
    /* taps beyond HALF are zero in tap_mpx; window indices are clamped into the window */
    auto load_group = [&](float2 (&x)[2 * G], float (&t)[G], int g) {
#pragma unroll
      for (int i = 0; i < G; i++) {
        const int k = g * G + i, kc = k < HALF ? k : HALF - 1;
        x[2 * i] = w[kc];
        x[2 * i + 1] = w[S - 1 - kc];
        t[i] = tp[k].x;
      }
    };
    auto use_group = [&](const float2 (&x)[2 * G], const float (&t)[G]) {
#pragma unroll
      for (int i = 0; i < G; i++) {
        om = mac<EX>(om, x[2 * i].x + x[2 * i + 1].x, t[i]);
        os = mac<EX>(os, x[2 * i].y + x[2 * i + 1].y, t[i]);
      }
    };
    static_assert(NG % 2 == 0, "group pairs");
    load_group(xa, ta, 0);
#pragma unroll 1
    for (int g = 0; g < NG; g += 2) {
      load_group(xb, tb, g + 1);
      use_group(xa, ta);
      sched_fence();
      load_group(xa, ta, g + 2 < NG ? g + 2 : 0);
      use_group(xb, tb);
      sched_fence();
    }
  } else {
    const int size = P.size, half = P.half;
    const float2 *w = newest - (size - 1);
    for (int k = 0; k < half; k++) {
      const float2 a = w[k], b = w[size - 1 - k];
      const float t = s.tap_mpx[k].x;
      om = mac<EX>(om, a.x + b.x, t);
      os = mac<EX>(os, a.y + b.y, t);
    }
  }
}

/* Block-start quirk (src/rtl_fm_player.c:534-598):
 * when the resampler emits on sample 0 of a block, the right-channel output is
 * stored over discriminator sample 1 before that sample is read. */
template <bool EX, int HALF>
__device__ __forceinline__ void q1_patch(const fmdk_params &P) {
  Smem &s = g_s;
  const int lane = threadIdx.x;
  float f = 0.f;
  if (lane < 3) {
    const int size = P.size, half = P.half;
    const float *w = &s.v[HV - (size - 1)];
    const float *tap = reinterpret_cast<const float *>(s.tap_mpx) + lane;
    for (int k = 0; k < half; k++) f = mac<EX>(f, w[k] + w[size - 1 - k], tap[4 * k]);
  }
  const float vp = __shfl(f, 1), vs = __shfl(f, 2);
  if (lane == 0) s.ms[HV] = make_float2(f, vs * carrier_of<EX>(vp, s.pp, P.swf, P.cwf));
  lds_barrier();
  if (lane == 0) {
    float om, os;
    fir_stereo<EX, 0>(P, &s.ms[HV], 0, om, os);
    s.v[HV + 1] = om - os;
  }
  lds_barrier();
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

template <bool EX, int MODE, int HALF>
__device__ __forceinline__ void resample_tile(const fmdk_params &P, uint32_t acc_t, int nq, int pend) {
  Smem &s = g_s;
  const uint32_t slow = (uint32_t)P.slow, fast = (uint32_t)P.fast;
  const float inv_slow = 1.0f / (float)P.slow;
  const bool rs = P.resample != 0;
  const int z = opaque_zero();
  for (int q = threadIdx.x; q < nq; q += NT) {
    const int i = rs ? emit_index(acc_t, q, slow, fast, inv_slow) : q;
    if constexpr (MODE == 2) {
      float om, os;
      fir_stereo<EX, HALF>(P, &s.ms[HV + i], z, om, os);
      *reinterpret_cast<float2 *>(&s.fr[pend + 2 * q]) = make_float2(om + os, om - os);   /* :595-596 */
    } else if constexpr (MODE == 1) {
      s.fr[pend + q] = fir_mono<EX, HALF>(P, &s.v[HV + i], z);
    } else {
      s.fr[pend + q] = s.v[HV + i];
    }
  }
}

/* ---- stage F: de-emphasis + s16 + store ----------------------------------- */

/* De-emphasis is a first-order recurrence (:687-709).  Each lane produces
 * DEEMPH_GROUP consecutive frames of one channel; a lane whose segment does not
 * start at the first pending frame restarts the recurrence P.warm frames early
 * from zero (lambda^warm < 1e-12, below fp32 resolution), the others continue
 * from the carried state, so the result equals the sequential evaluation. */
template <bool EX, int CH>
__device__ __forceinline__ void flush_frames(const fmdk_params &P, int pend, int16_t *pcm_out,
                                             float *mpx_dbg, bool store) {
  Smem &s = g_s;
  const int frames = pend / CH;
  const float coef = P.coef;
  if (mpx_dbg) {
    for (int i = threadIdx.x; i < pend; i += NT) mpx_dbg[i] = s.fr[i];
  }
  if (P.deemph) {
    const int groups = (frames + DEEMPH_GROUP - 1) / DEEMPH_GROUP;
    const float lam = P.lambda;
    const int warm = P.warm;
    for (int task = threadIdx.x; task < groups * CH; task += NT) {
      const int g = task / CH, c = task % CH;
      const int f_out = g * DEEMPH_GROUP;
      int f = f_out - warm;
      float y = 0.f;
      if (f <= 0) { f = 0; y = s.de[c]; }
      const int f_end = min(f_out + DEEMPH_GROUP, frames);
#pragma unroll 4
      for (; f < f_out; f++) {             /* warm-up, nothing stored */
        const float x = s.fr[f * CH + c];
        const float t = y - x;
        if constexpr (EX) y = x + lam * t;
        else y = __builtin_fmaf(lam, t, x);
      }
      for (; f < f_end; f++) {
        const float x = s.fr[f * CH + c];
        const float t = y - x;
        if constexpr (EX) y = x + lam * t;
        else y = __builtin_fmaf(lam, t, x);
        if (store) pcm_out[f * CH + c] = to_s16(y, coef);
      }
      if (f_end == frames) s.de[2 + c] = y;
    }
    lds_barrier();
    if (threadIdx.x < CH && frames > 0) s.de[threadIdx.x] = s.de[2 + threadIdx.x];
  } else {
    if (store)
      for (int i = threadIdx.x; i < pend; i += NT) pcm_out[i] = to_s16(s.fr[i], coef);
  }
  lds_barrier();
}

/* ---- history roll ---------------------------------------------------------- */

template <int MODE>
__device__ __forceinline__ void roll_history(int tm) {
  Smem &s = g_s;
  const int tid = threadIdx.x;
  float hv = 0.f;
  float2 hm = make_float2(0.f, 0.f);
  if (tid < HV) {
    hv = s.v[tm + tid];
    if constexpr (MODE == 2) hm = s.ms[tm + tid];
  }
  lds_barrier();
  if (tid < HV) {
    s.v[tid] = hv;
    if constexpr (MODE == 2) s.ms[tid] = hm;
  }
  /* the next barrier orders these writes before any read */
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

template <int MODE>
__device__ __forceinline__ void state_in(const fmdk_params &P, const DevState *st, bool carried) {
  Smem &s = g_s;
  const int tid = threadIdx.x, size = P.size;
  for (int i = tid; i < 128; i += NT) s.tap_mpx[i] = f4{P.fm[i], P.fp[i], P.fs[i], 0.f};
  for (int i = tid; i < size; i += NT) {
    s.v[HV - size + i] = carried ? st->br[i] : 0.f;
    if constexpr (MODE == 2)
      s.ms[HV - size + i] = carried ? make_float2(st->bm[i], st->bs[i]) : make_float2(0.f, 0.f);
  }
  if (tid == 0) {
    s.y[0] = carried ? make_float2(st->pre_r, st->pre_j) : make_float2(0.f, 0.f);
    s.pp = carried ? st->pp : 0.f;
    s.de[0] = carried ? st->de_l : 0.f;
    s.de[1] = carried ? st->de_r : 0.f;
  }
}

template <int MODE>
__device__ __forceinline__ void state_out(const fmdk_params &P, DevState *st, int last_buf, int sm_last,
                                          uint32_t acc) {
  Smem &s = g_s;
  const int tid = threadIdx.x, size = P.size;
  /* lowpass_tb: the last 24 complex samples, rotated, as floats (:366) */
  if (tid < 48) {
    const uint8_t *raw = reinterpret_cast<const uint8_t *>(s.iq[last_buf]) + 16 * sm_last;   /* 48 bytes */
    const int j = tid >> 1, comp = tid & 1, p = j & 3;   /* 24 samples: phase = j mod 4 */
    int sel; float sg;
    if (P.offset_tuning) { sel = comp; sg = 1.f; }
    else {
      sel = comp ? sel_q<true>(p) : sel_i<true>(p);
      sg = comp ? sgn_q<true>(p) : sgn_i<true>(p);
    }
    st->tb[tid] = sg * t0((float)raw[2 * j + sel]);
  }
  for (int i = tid; i < size; i += NT) {
    st->br[i] = s.v[HV - size + i];
    if constexpr (MODE == 2) {
      const float2 m = s.ms[HV - size + i];
      st->bm[i] = m.x;
      st->bs[i] = m.y;
    }
  }
  if (tid == 0) {
    st->pre_r = s.y[0].x;
    st->pre_j = s.y[0].y;
    if constexpr (MODE == 2) st->pp = s.pp;
    st->de_l = s.de[0];
    st->de_r = s.de[1];
    st->acc = (int32_t)acc;
  }
}

/* ---- the fused kernel ----------------------------------------------------- */

template <bool EX, int MODE, int HALF>
__global__ __launch_bounds__(NT, 2) void fmd_fused_kernel(const fmdk_params P, const uint8_t *__restrict__ iq_all,
                                                         int16_t *__restrict__ pcm_all,
                                                         int32_t *__restrict__ lens_all,
                                                         const DevState *__restrict__ state_in_all,
                                                         DevState *__restrict__ state_out_all, float *dbg_y,
                                                         float *dbg_v, float *dbg_mpx, long long *dbg_prof) {
  Smem &s = g_s;
  /* optional per-stage cycle accounting (fmd_debug_taps.prof) */
  long long pf[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long pf_last = 0, pf_start = 0;
  if (dbg_prof) pf_start = pf_last = clock64();
#define FMD_STAMP(i)                                   \
  if (dbg_prof) {                                      \
    const long long now_ = clock64();                  \
    pf[i] += now_ - pf_last;                           \
    pf_last = now_;                                    \
  }
  constexpr int CH = (MODE == 2) ? 2 : 1;
  const int tid = threadIdx.x;
  const int K = P.n_chunks;
  const int stream = blockIdx.x / K, chunk = blockIdx.x - stream * K;
  const int M = P.block_len >> 4;                 /* rate_in samples per block */
  const uint32_t slow = (uint32_t)P.slow, fast = (uint32_t)P.fast;
  const uint8_t *iq_stream = iq_all + (size_t)stream * P.n_blocks * P.block_len;
  const DevState *st_in = state_in_all + stream;

  /* this workgroup's blocks, and the rate_in sample range it walks (a chunk > 0
   * starts warm_tiles tiles early and discards what those produce) */
  const int b_lo = (int)((long long)chunk * P.n_blocks / K);
  const int b_hi = (int)((long long)(chunk + 1) * P.n_blocks / K);
  const long long n_real = (long long)b_lo * M;
  const long long n_lo = n_real - (chunk > 0 ? (long long)P.warm_tiles * TM : 0);
  const long long n_hi = (long long)b_hi * M;

  state_in<MODE>(P, st_in, chunk == 0);
  uint32_t acc = (uint32_t)st_in->acc;            /* uniform */
  if (chunk > 0 && P.resample)
    acc = (uint32_t)(((unsigned long long)acc + (unsigned long long)n_lo * slow) % fast);

  /* first sub-tile; at the start of the launch there are no 48 halo bytes in
   * front of it: chunk 0 takes them from the float history (decimate_head), a
   * replaying chunk starts from zero state anyway */
  if (n_lo < n_hi) {
    const int off = (int)(n_lo % M);
    const bool no_halo = (n_lo == 0);
    const uint4 *src = reinterpret_cast<const uint4 *>(iq_stream + n_lo * 16) - 3;
    load_sub_async(src, 0, no_halo ? 3 : 0, min(SUB, M - off) + 3);
    if (no_halo && tid < 3) s.iq[0][tid] = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
  }
  FMD_STAMP(9)

  int buf = 0, sm_last = 0;
  int pend = 0, pcm_off = 0;
  bool q1 = false, head = (chunk == 0);
  for (long long pos = n_lo; pos < n_hi;) {       /* one iteration per tile */
    const int b = (int)(pos / M), off = (int)(pos - (long long)b * M);
    const int tm = min(TM, M - off);
    const bool discard = pos < n_real;
    const size_t slot = (size_t)stream * P.n_blocks + b;
    int16_t *pcm_blk = pcm_all + slot * P.pcm_stride;
    float *mpx_blk = (dbg_mpx && !discard) ? dbg_mpx + slot * M : nullptr;
    if (off == 0 || pos == n_lo) {
      pend = 0; pcm_off = 0;
      q1 = (MODE == 2) && off == 0 && P.resample && (acc + slow >= fast);
    }

    /* ---- A + B on the tile's sub-tiles ---- */
    for (int u0 = 0; u0 < tm; u0 += SUB) {
      const int sm = min(SUB, tm - u0);
      /* this sub-tile's IQ has landed; start fetching the next one */
      full_barrier();
      const long long nxt = pos + u0 + sm;
      if (nxt < n_hi) {
        const int off2 = (int)(nxt % M);
        load_sub_async(reinterpret_cast<const uint4 *>(iq_stream + nxt * 16) - 3, buf ^ 1, 0,
                       min(SUB, M - off2) + 3);
      }
      FMD_STAMP(0)

      if (P.offset_tuning) decimate_sub<EX, false>(P, buf, sm);
      else decimate_sub<EX, true>(P, buf, sm);
      if (head) {
        lds_barrier();
        if (P.offset_tuning) decimate_head<false>(P, buf, st_in->tb, sm);
        else decimate_head<true>(P, buf, st_in->tb, sm);
        head = false;
      }
      lds_barrier();
      FMD_STAMP(1)
      if (dbg_y && !discard) {
        float2 *o = reinterpret_cast<float2 *>(dbg_y) + slot * M + off + u0;
        for (int m = tid; m < sm; m += NT) o[m] = s.y[1 + m];
      }

      discriminate_sub<EX>(u0, sm);
      const float2 ylast = s.y[sm];
      lds_barrier();
      if (tid == 0) s.y[0] = ylast;
      FMD_STAMP(2)
      if (dbg_v && !discard) {
        float *o = dbg_v + slot * M + off + u0;
        for (int m = tid; m < sm; m += NT) o[m] = s.v[HV + u0 + m];
      }
      sm_last = sm;
      buf ^= 1;
    }

    /* ---- Q + C: stereo MPX filters ---- */
    if constexpr (MODE == 2) {
      if (q1 && off == 0 && tm > 1) q1_patch<EX, HALF>(P);
      FMD_STAMP(3)
      mpx_tile<EX, HALF>(P, tm);
      FMD_STAMP(4)
    }

    /* ---- D: resampler outputs of this tile ---- */
    int nq;
    if (P.resample) nq = (int)(((unsigned long long)acc + (unsigned long long)tm * slow) / fast);
    else nq = tm;
    if (pend + nq * CH > CAPF) {
      flush_frames<EX, CH>(P, pend, pcm_blk + pcm_off, mpx_blk ? mpx_blk + pcm_off : nullptr, !discard);
      pcm_off += pend;
      pend = 0;
      FMD_STAMP(8)
    }
    resample_tile<EX, MODE, HALF>(P, acc, nq, pend);
    pend += nq * CH;
    if (P.resample) acc = (uint32_t)(((unsigned long long)acc + (unsigned long long)tm * slow) % fast);
    lds_barrier();
    FMD_STAMP(6)

    /* ---- roll the FIR histories to the front of their buffers ---- */
    roll_history<MODE>(tm);
    FMD_STAMP(7)

    /* ---- F: end of block -> PCM ---- */
    if (off + tm == M) {
      lds_barrier();
      flush_frames<EX, CH>(P, pend, pcm_blk + pcm_off, mpx_blk ? mpx_blk + pcm_off : nullptr, !discard);
      if (tid == 0 && !discard) lens_all[slot] = pcm_off + pend;
      FMD_STAMP(8)
    }
    pos += tm;
  }

  /* carried state -> HBM (the chunk that ends the launch) */
  lds_barrier();
  if (chunk == K - 1 && n_lo < n_hi) state_out<MODE>(P, state_out_all + stream, buf ^ 1, sm_last, acc);
  FMD_STAMP(9)
  if (dbg_prof && tid == 0) {
    long long *o = dbg_prof + 16 * (size_t)blockIdx.x;
    for (int i = 0; i < 10; i++) o[i] = pf[i];
    o[15] = pf_last - pf_start;
  }
#undef FMD_STAMP
}

template <bool EX, int MODE, int HALF>
int launch_one(const fmdk_params *p, int n_streams, const void *iq, void *pcm, void *lens,
               const void *state_in, void *state_out, const fmd_debug_taps *dbg, hipStream_t stream) {
  hipLaunchKernelGGL((fmd_fused_kernel<EX, MODE, HALF>), dim3(n_streams * p->n_chunks), dim3(NT), 0, stream,
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

/* FIR memories: 24 IQ + 1 (discriminator) + 2 x (size - 1) rate_in samples; the
 * de-emphasis restart needs warm frames = warm * fast / slow rate_in samples. */
extern "C" int fmdk_warm_tiles(const fmdk_params *p) {
  long long need = 8 + 2LL * p->size;
  if (p->deemph) {
    if (p->warm >= (1 << 20)) return 0;
    need += ((long long)p->warm + DEEMPH_GROUP) * (p->resample ? (p->fast + p->slow - 1) / p->slow : 1);
  }
  const long long tiles = (need + TM - 1) / TM;
  if (tiles * TM > (p->block_len >> 4)) return 0;      /* replay must stay inside the previous block */
  return (int)tiles;
}

extern "C" const char *fmdk_kernel_name(const fmdk_params *p, int math) {
  (void)p;
  (void)math;
  return "fmd_fused_kernel";
}

extern "C" int fmdk_lds_bytes(void) { return (int)sizeof(Smem); }
