/*
 * fmd_host.c - C host layer of libfmdemod_mi355x.so.
 *
 * Plain C above the HIP runtime's C API: configuration and filter design,
 * device buffers and carried state, the batch API, the reference-shaped
 * entry points (same names / struct layout as rtl_fm_player.c) and the
 * rtlsdr_read_async-compatible ingest ring.  All arithmetic of the hot path
 * runs in the kernels of fmd_kernels.inc (built as fmd_kernels_{exact,fast,mfma}.hip); nothing here computes a sample.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "fmd_internal.h"

/* ---- layout of the reference structs ---- */
_Static_assert(offsetof(struct demod_state, buf) == 16, "demod_state.buf");
_Static_assert(offsetof(struct demod_state, buf_len) == 262160, "demod_state.buf_len");
_Static_assert(offsetof(struct demod_state, lowpassed) == 262164, "demod_state.lowpassed");
_Static_assert(offsetof(struct demod_state, lp_len) == 1310740, "demod_state.lp_len");
_Static_assert(offsetof(struct demod_state, lowpass_tb) == 1310744, "demod_state.lowpass_tb");
_Static_assert(offsetof(struct demod_state, result) == 1311176, "demod_state.result");
_Static_assert(offsetof(struct demod_state, result_len) == 1835464, "demod_state.result_len");
_Static_assert(offsetof(struct demod_state, rate_in) == 1835508, "demod_state.rate_in");
_Static_assert(offsetof(struct demod_state, rate_out) == 1835512, "demod_state.rate_out");
_Static_assert(offsetof(struct demod_state, rate_out2) == 1835516, "demod_state.rate_out2");
_Static_assert(offsetof(struct demod_state, pre_r_f32) == 1835536, "demod_state.pre_r_f32");
_Static_assert(offsetof(struct demod_state, deemph) == 1835592, "demod_state.deemph");
_Static_assert(offsetof(struct demod_state, deemph_l_f32) == 1835612, "demod_state.deemph_l_f32");
_Static_assert(offsetof(struct demod_state, deemph_lambda) == 1835620, "demod_state.deemph_lambda");
_Static_assert(offsetof(struct demod_state, volume) == 1835624, "demod_state.volume");
_Static_assert(offsetof(struct demod_state, prev_lpr_index) == 1835632, "demod_state.prev_lpr_index");
_Static_assert(offsetof(struct demod_state, lpr) == 1835640, "demod_state.lpr");
_Static_assert(offsetof(struct demod_state, rw) == 1835720, "demod_state.rw");
_Static_assert(offsetof(struct demod_state, ready) == 1835776, "demod_state.ready");
_Static_assert(offsetof(struct demod_state, ready_m) == 1835824, "demod_state.ready_m");
_Static_assert(offsetof(struct demod_state, output_target) == 1835864, "demod_state.output_target");
_Static_assert(sizeof(struct demod_state) == 1835872, "sizeof(struct demod_state)");
_Static_assert(sizeof(struct lp_real) == 80, "sizeof(struct lp_real)");
_Static_assert(offsetof(struct lp_real, swf) == 48, "lp_real.swf");
_Static_assert(offsetof(struct lp_real, pos) == 60, "lp_real.pos");
_Static_assert(offsetof(struct lp_real, mode) == 72, "lp_real.mode");

#define FMD_PI 3.14159265f   /* PI_F  include/rtl_fm_player.h:40 */
#define FMD_2PI 6.28318531f  /* PI2_F include/rtl_fm_player.h:39 */

static __thread char g_err[256];

static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

const char *fmd_last_error(void) { return g_err; }

#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(FMD_E_HIP, "%s failed: %s (%d)", #expr, hipGetErrorString(e_), (int)e_); \
  } while (0)

int fmd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

/* Environment knobs that change the arithmetic thresholds or the kernels' shape exist only in tuning builds
 * (make EXTRA_CFLAGS=-DFMD_TUNING; there FMD_MFMA also pins the family behind FMD_MATH_FAST): the shipped library reads ONE
 * variable, FMD_MATH_FAST, and only for the reference-shaped surface whose signatures have no room for the choice.  The batch
 * API's kernel family is fmd_config.math and nothing else. */
#ifndef FMD_CARRIER_L2_DEFAULT
#define FMD_CARRIER_L2_DEFAULT 0.5f     /* measured: 0.125 fails 4 of the noise / hand-over tests, 0.25 and up none (r04x); 1: always the full redo */
#endif
static const char *tuning_env(const char *name) {
#ifdef FMD_TUNING
  return getenv(name);
#else
  (void)name;
  return NULL;
#endif
}

/* ---- filter design: init_lp_f32 / init_lp_real_f32 restated ------------- */

float fmd_deemph_lambda(int output_rate, double tau) {
  return (float)exp(-1.0 / ((double)output_rate * tau));   /* src/rtl_fm_player.c:1577 */
}

static void design_fb(float *fb) {   /* src/rtl_fm_player.c:241-251 */
  for (int i = 0; i < 16; i++) {
    float j = (float)i - 15.5f;
    fb[i] = (sinf(0.125f * FMD_PI * j) / (FMD_PI * j)) * (0.54f - 0.46f * cosf(FMD_PI * (float)i / 15.5f));
  }
}

static void design_mpx(int size, int rate_in, float *fm, float *fp, float *fs, float *swf, float *cwf) {
  /* src/rtl_fm_player.c:420-452 */
  const float rate = (float)rate_in;
  const float wf = FMD_2PI * 19000.0f / rate;
  *swf = sinf(wf);
  *cwf = cosf(wf);
  const float fmh = 16000.0f / rate, fpl = 18000.0f / rate, fph = 20000.0f / rate;
  const float fsl = 21000.0f / rate, fsh = 55000.0f / rate;
  for (int i = 0; i < (size >> 1); i++) {
    const float fi = (float)i - (float)(size - 1) / 2.0f;
    const float fh = 0.54f - 0.46f * cosf(FMD_2PI * (float)i / (float)(size - 1));
    float fv;
    fv = (fi == 0) ? 2.0f * fmh : sinf(FMD_2PI * fmh * fi) / (FMD_PI * fi);
    fm[i] = fv * fh;
    fv = (fi == 0) ? 2.0f * (fph - fpl) : (sinf(FMD_2PI * fph * fi) - sinf(FMD_2PI * fpl * fi)) / (FMD_PI * fi);
    fp[i] = fv * fh;
    fv = (fi == 0) ? 2.0f * (fsh - fsl) : (sinf(FMD_2PI * fsh * fi) - sinf(FMD_2PI * fsl * fi)) / (FMD_PI * fi);
    fs[i] = fv * fh;
  }
}

static int check_config(const fmd_config *c) {
  if (!c) return fail(FMD_E_ARG, "config is NULL");
  if (c->rate_in <= 0) return fail(FMD_E_ARG, "rate_in must be positive");
  if (c->mode < 0 || c->mode > 2) return fail(FMD_E_ARG, "lpr.mode must be 0, 1 or 2");
  if (c->size < 2 || c->size > 256 || (c->size & 1)) return fail(FMD_E_ARG, "lpr.size must be even, 2..256");
  if (c->block_len < 64 || (c->block_len & 15)) return fail(FMD_E_ARG, "block_len must be a multiple of 16, >= 64");
  if (c->math < FMD_MATH_EXACT || c->math > FMD_MATH_FAST_MFMA_F)
    return fail(FMD_E_ARG, "math must be FMD_MATH_EXACT, _FAST, _FAST_VALU, _FAST_MFMA or _FAST_MFMA_F (4 - 6, the retired family names, mean _FAST)");
  /* the +-1 LSB kernels evaluate the de-emphasis blockwise with powers of lambda (scan weights, restarts from zero):
   * a contraction is assumed.  lambda outside (0, 1) - never produced by fmd_deemph_lambda - belongs to the exact kernels */
  if (c->math != FMD_MATH_EXACT && c->deemph && !(c->deemph_lambda > 0.f && c->deemph_lambda < 1.f))
    return fail(FMD_E_UNSUPPORTED, "the fast kernels need 0 < deemph_lambda < 1 (got %g): use FMD_MATH_EXACT", (double)c->deemph_lambda);
  if (c->rate_out2 > 0) {
    if (c->rate_out <= 0 || c->rate_out > 2000000) return fail(FMD_E_UNSUPPORTED, "rate_out must be 1..2000000");
    if (c->rate_out2 > c->rate_out)
      return fail(FMD_E_UNSUPPORTED, "rate_out2 > rate_out overflows the reference's accumulator");
    /* stereo writes two outputs per emit over its own input; beyond 1/3 the
     * in-place overwrite reaches more than the block's second sample */
    if (c->mode == 2 && 3LL * c->rate_out2 > c->rate_out)
      return fail(FMD_E_UNSUPPORTED, "stereo needs rate_out2 <= rate_out / 3");
  } else if (c->mode == 2) {
    return fail(FMD_E_UNSUPPORTED, "stereo without the resampler is not supported");
  }
  return FMD_OK;
}

int fmd_design_taps(const fmd_config *cfg, fmd_taps *out) {
  if (!out) return fail(FMD_E_ARG, "taps is NULL");
  int rc = check_config(cfg);
  if (rc) return rc;
  memset(out, 0, sizeof(*out));
  design_fb(out->fb);
  design_mpx(cfg->size, cfg->rate_in, out->fm, out->fp, out->fs, &out->swf, &out->cwf);
  return FMD_OK;
}

/* ---- batch object --------------------------------------------------------- */

struct fmd_ingest;

struct fmd_batch {
  fmd_config cfg;
  fmd_taps taps;
  fmdk_params kp;
  int n_streams;
  int device;
  int pcm_stride;
  hipStream_t stream;
  hipEvent_t ev0, ev1;
  int no_timing;               /* fmd_batch_set_timing(b, 0): no event pair around the kernel */
  int timed;
  void *d_state[2];            /* fmd_stream_state[n_streams], ping-pong: a multi-chunk launch
                                  reads one and writes the other                        */
  int cur;                     /* index of the buffer holding the current state          */
  hipStream_t last_stream;     /* stream of the most recent launch (b->stream or the caller's)      */
  int launched;                /* a launch has been queued on last_stream                            */
  hipEvent_t ev_order;         /* orders the state ping-pong when consecutive launches change stream */
  int n_cus;
  int time_split;              /* fmd_batch_set_time_split: 0 default, > 0 workers per CU to cut for, < 0 never split */
  /* staging for the host-buffer path, grown on demand */
  void *d_iq, *d_pcm, *d_lens;
  void *d_dec_tables;          /* FMD_MATH_FAST_MFMA_F: the phase tables of the decimating second stage (build_dec_tables), NULL otherwise */
  size_t cap_blocks;
  /* ingest */
  struct fmd_ingest **ingest;  /* [n_streams], NULL when unbound */
  /* fmd_batch_pump_begin/_end: two jobs in flight, each with its own pinned and device buffers */
  struct pump_slot {
    int16_t *h_pcm; int32_t *h_lens;                   /* pinned */
    void *d_iq, *d_pcm, *d_lens;
    size_t cap_blocks;
    int n_blocks;                                      /* > 0: job in flight */
    int ring_held;                                     /* its bytes are still held in the rings (H2D source) */
    int failed;                                        /* hipError_t of a D2H enqueue that failed after the kernel was launched */
    hipEvent_t h2d_done, done;
  } pump[2];
  int pump_head, pump_tail;    /* next slot to begin / oldest slot not yet ended */
  hipStream_t copy_stream;     /* H2D of job k+1 runs beside the kernel of job k */
};

static int max_result_len(const fmd_config *c) {
  const long m = c->block_len / 16;
  long n;
  if (c->rate_out2 > 0) n = (m * (long)c->rate_out2) / c->rate_out + 1;
  else n = m;
  if (c->mode == 2) n *= 2;
  return (int)n;
}

/* Stage A on the matrix pipe (FMD_MATH_FAST_MFMA): the A operand of v_mfma_i32_16x16x64_i8.
 * Output m of the /8 low-pass (src/rtl_fm_player.c:253-411, rotation :206-226 folded in) is
 *   y_c[m] = sum_{j<32} sgn_c(j) fb[min(j, 31-j)] x[8m - 24 + j][sel_c(j)],   x = (u - 127.5) / 128,
 * a dot product of the 64 window bytes with a vector that has 32 non-zero entries.  With E = sgn round(fb 2^26)
 * (|E| < 2^23, three balanced int8 limbs) and s = u - 128 the sum  S = sum E s  is EXACT integer arithmetic and
 *   y = 2^-33 (S + sum E / 2) = 2^-17 S0 + 2^-25 S1 + 2^-33 S2 + bias.
 * Tap quantisation moves y by at most 32 x 2^-27 |x| <= 2.4e-7 (rms 2.4e-8): the size of the fp32 rounding of the
 * reference's own sum, inside the +-1 LSB contract like the fused sums of FMD_MATH_FAST_VALU.
 * Entry [limb][comp][d] holds the 16 bytes (8 samples x {I, Q}) of taps 8d .. 8d+7. */
static int build_a_tab(const fmd_taps *t, int offset_tuning, fmdk_params *k) {   /* -1: a tap does not fit three limbs */
  long long sum[2] = {0, 0};
  int8_t *tab = (int8_t *)k->a_tab;
  memset(k->a_tab, 0, sizeof(k->a_tab));
  for (int j = 0; j < 32; j++) {
    const double tap = (double)t->fb[j < 16 ? j : 31 - j];
    const long long T = llround(tap * 67108864.0);        /* 2^26 */
    const int p = j & 3, d = j >> 3, jj = j & 7;
    for (int comp = 0; comp < 2; comp++) {
      int sel = comp, sg = 1;
      if (!offset_tuning) {                                /* j^p: I = (+I, -Q, -I, +Q), Q = (+Q, +I, -Q, -I) */
        sel = comp ? ((p & 1) ^ 1) : (p & 1);
        sg = comp ? ((p == 0 || p == 1) ? 1 : -1) : ((p == 0 || p == 3) ? 1 : -1);
      }
      long long E = sg * T;
      sum[comp] += E;
      int limb[3];
      for (int i = 2; i >= 0; i--) {                       /* balanced digits, least significant first */
        long long r = ((E % 256) + 256) % 256;
        if (r >= 128) r -= 256;
        limb[i] = (int)r;
        E = (E - r) / 256;
      }
      if (E != 0) return -1;                               /* |tap| >= 0.1245: beyond 2^23 / 2^26 (the reference's largest is 0.1239) */
      for (int l = 0; l < 3; l++) tab[(((l * 2 + comp) * 4 + d) * 16) + 2 * jj + sel] = (int8_t)limb[l];
    }
  }
  k->a_bias_i = (float)ldexp((double)sum[0], -34);
  k->a_bias_q = (float)ldexp((double)sum[1], -34);
  return 0;
}

/* Matrix-pipe form of stage C (fmd_kernels.inc, mpx_tile_i8): per filter the largest qf that keeps round(h 2^qf) inside three balanced
 * int8 limbs (-8 421 504 .. 8 355 711), and the scale that puts the integer sums together.  -1: not the 90-tap stereo kind, or a
 * filter is all zero. */
static int build_ci_scales(const fmd_taps *t, int size, fmdk_params *k) {
  if (size != 90) return -1;
  const float *taps[3] = {t->fm, t->fp, t->fs};
  for (int f = 0; f < 3; f++) {
    double mx = 0.0;
    for (int u = 0; u < 45; u++) mx = fmax(mx, fabs((double)taps[f][u]));
    if (!(mx > 0.0) || !isfinite(mx)) return -1;
    int qf = 40;
    while (qf > 0 && llround(mx * ldexp(1.0, qf)) > 8355711LL) qf--;
    if (qf < 8) return -1;                                  /* taps of magnitude 2^15: not a filter this form was made for */
    /* The kernel reads its int32 limb-pair sums as floats (accumulators started at the bits of 1.5 x 2^23, mpx_tile_i8): every weight
     * class must stay inside +-2^22 for ANY samples (limbs within +-128).  Classes by tap limb: class 0 = T0, class 1 = T0 + T1,
     * class 2 = T0 + T1 + T2, class 3 = T1 + T2 (the sample limb is what is left of the class index). */
    double sum_abs[3] = {0.0, 0.0, 0.0};
    for (int u = 0; u < 90; u++) {
      const long long E = llround((double)taps[f][u < 45 ? u : 89 - u] * ldexp(1.0, qf));
      const unsigned q = ((unsigned)(int)E + 0x808080u) ^ 0x808080u;
      for (int l = 0; l < 3; l++) sum_abs[l] += fabs((double)(signed char)(q >> (8 * (2 - l))));
    }
    if (128.0 * (sum_abs[0] + sum_abs[1] + sum_abs[2]) >= 4194304.0 - 65536.0) return -1;
    k->ci_qf[f] = qf;
    k->ci_scale[f] = (float)ldexp(1.0, 32 - 20 - qf);
    k->ci_scale_q[f] = (float)ldexp(1.0, 32 - qf);
  }
  return 0;
}

/* (stage_d_on_matrix_pipe, below: the second stage on the matrix pipe needs what stage C needs - build_ci_scales - and: at most eight groups of sixteen
 * frames per tile for stereo (rate_out >= 4 rate_out2; mono: sixteen, rate_out >= 2 rate_out2), both magic-number index forms, the error estimate under its
 * limit, and (L-R) x carrier inside the limbs' range |x| < 8 for any discriminator output - |v| <= pi, and what quirk Q1 can put in place of a sample:
 * |om - os| <= 2 pi sum|fm| sum|f| - true of any filter of the reference's design, checked for a caller's.) */
/* The 128-tap mono path's one filter in the same fixed-point form (resample_mono_dec): T = round(fm 2^qf) in three balanced int8 limbs,
 * every weight class of the limb-pair sums inside +-2^22 for any samples. */
static int build_ci_scales_mono(const fmd_taps *t, int size, fmdk_params *k) {
  if (size != 128) return -1;
  double mx = 0.0, sum_abs = 0.0, sa = 0.0;
  for (int u = 0; u < 64; u++) { mx = fmax(mx, fabs((double)t->fm[u])); sa += 2.0 * fabs((double)t->fm[u]); }
  if (!(mx > 0.0) || !isfinite(mx) || !(3.1415927 * sa < 7.9)) return -1;      /* (and the samples inside the limbs' range: |v| <= pi) */
  int qf = 40;
  while (qf > 0 && llround(mx * ldexp(1.0, qf)) > 8355711LL) qf--;
  if (qf < 8) return -1;
  for (int u = 0; u < 128; u++) {
    const long long E = llround((double)t->fm[u < 64 ? u : 127 - u] * ldexp(1.0, qf));
    const unsigned q = ((unsigned)(int)E + 0x808080u) ^ 0x808080u;
    for (int l = 0; l < 3; l++) sum_abs += fabs((double)(signed char)(q >> (8 * (2 - l))));
  }
  if (128.0 * sum_abs >= 4194304.0 - 65536.0) return -1;
  k->ci_qf[0] = qf;
  k->ci_scale[0] = (float)ldexp(1.0, 32 - 20 - qf);
  k->ci_scale_q[0] = (float)ldexp(1.0, 32 - qf);
  return 0;
}

/* What a second-stage filter's fixed-point form (taps T = round(h 2^qf) and samples q = round(x 2^20) in three balanced int8 limbs each, six of the nine limb
 * pairs kept) adds to a PCM value, in LSB: the filter's output IS the PCM value before de-emphasis and scaling, so an error e in it is e x coef LSB
 * (coef = volume x 32768).  Three terms, each as an rms ESTIMATE and as a worst-case BOUND from the filter's own taps and limbs:
 *   the limb pairs left out (tap limb + sample limb >= 3): S3 = sum_k (t1 s2 + t2 s1) at weight c0 2^-24 and S4 = sum_k t2 s2 at c0 2^-32, c0 = 2^(12 - qf).
 *     rms: 2 n products of two limbs of rms 74 each; bound: |sample limb| <= 128, so |S3| <= 128 sum_k (|t1| + |t2|), |S4| <= 128 sum_k |t2|;
 *   the samples' rounding to 2^-20: rms 2^-21 / sqrt 3 per sample through the filter (x sqrt(sum h^2)); bound 2^-21 sum |h|;
 *   the taps' rounding to 2^-qf: rms 2^-(qf+1) / sqrt 3 per tap, n taps, samples of rms ~1.8 at most (a discriminator output uniform in +-pi);
 *     bound n 2^-(qf+1) pi (the L-R channel's samples are (L-R band) x carrier: the same range).
 * 300 k stereo / mono: rms 0.004 at volume 0.4, 0.08 - 0.09 at volume 8 (max |difference| 1 LSB measured); 25 k narrow FM (largest tap 0.58: qf 23, c0
 * eight times the 300 k filters'): 0.035 at volume 0.4, 0.09 at 1, 0.26 at 3 (still 1 LSB at most in 262 144 values) and 0.70 at volume 8, where 3 LSB
 * were measured (profiles/archive/r5q_low_amp_volume_scan_before.txt).  The GATE is the rms estimate <= FMD_STAGE_D_MAX_LSB (ten standard deviations below
 * one step: a statistical guarantee); the bound is reported (fmd_config_error_estimate) and is below half a step for the reference's
 * default configurations. */
typedef struct { double rms, worst_samples, worst_taps, worst_dropped; int qf, n; } stage_error;
static stage_error fixed_point_error(const double *h, int n, int qf, double coef) {
  stage_error e = {0.0, 0.0, 0.0, 0.0, qf, n};
  double sh2 = 0.0, sabs = 0.0, a1 = 0.0, a2 = 0.0;
  for (int u = 0; u < n; u++) {
    sh2 += h[u] * h[u];
    sabs += fabs(h[u]);
    const long long E = llround(h[u] * ldexp(1.0, qf));
    const unsigned q = ((unsigned)(int)E + 0x808080u) ^ 0x808080u;
    a1 += fabs((double)(signed char)(q >> 8));
    a2 += fabs((double)(signed char)q);
  }
  const double c0 = ldexp(1.0, 12 - qf), ac = fabs(coef);
  const double dropped = c0 * ldexp(1.0, -24) * sqrt(2.0 * n) * 74.0 * 74.0;
  const double samples = ldexp(1.0, -21) / sqrt(3.0) * sqrt(sh2);
  const double taps = sqrt((double)n) * ldexp(1.0, -(qf + 1)) / sqrt(3.0) * 1.8;
  e.rms = ac * sqrt(dropped * dropped + samples * samples + taps * taps);
  e.worst_dropped = ac * c0 * (ldexp(1.0, -24) * 128.0 * (a1 + a2) + ldexp(1.0, -32) * 128.0 * a2);
  e.worst_samples = ac * ldexp(1.0, -21) * sabs;
  e.worst_taps = ac * (double)n * ldexp(1.0, -(qf + 1)) * 3.14159265358979;
  return e;
}
static int taps_qf(const double *h, int n) {
  double mx = 0.0;
  for (int u = 0; u < n; u++) mx = fmax(mx, fabs(h[u]));
  if (!(mx > 0.0) || !isfinite(mx)) return -1;
  int qf = 40;
  while (qf > 0 && llround(mx * ldexp(1.0, qf)) > 8355711LL) qf--;
  return qf;
}
static void fm_full(const float *fm, int n, double *h) { for (int u = 0; u < n; u++) h[u] = (double)fm[u < n / 2 ? u : n - 1 - u]; }
static void composite_taps(const float *fm, double *g /* [179] */) {
  for (int u = 0; u < 179; u++) {
    double a = 0.0;
    for (int i = 0; i < 90; i++) {
      const int j = u - i;
      if (j < 0 || j >= 90) continue;
      a += (double)fm[i < 45 ? i : 89 - i] * (double)fm[j < 45 ? j : 89 - j];
    }
    g[u] = a;
  }
}
static double stage_d_error_lsb(const float *fm, int n, int qf, float coef) {
  double h[256];
  fm_full(fm, n, h);
  return fixed_point_error(h, n, qf, (double)coef).rms;
}
#define FMD_STAGE_D_MAX_LSB 0.10

static int stage_d_on_matrix_pipe(const fmd_taps *t, const fmdk_params *k) {
  if (k->resample && k->mode == 1 && k->size == 128)       /* mono: rate_out >= 2 rate_out2 (at most sixteen groups of sixteen frames per tile) */
    return (long long)k->fast >= 2LL * k->slow && k->emit_magic && k->tf_magic &&
           stage_d_error_lsb(t->fm, 128, k->ci_qf[0], k->coef) <= FMD_STAGE_D_MAX_LSB;
  if (!(k->resample && k->mode == 2 && k->size == 90)) return 0;
  if ((long long)k->fast < 4LL * k->slow || !k->emit_magic || !k->tf_magic) return 0;
  if (stage_d_error_lsb(t->fm, 90, k->ci_qf[0], k->coef) > FMD_STAGE_D_MAX_LSB) return 0;
  double sm = 0.0, ss = 0.0;
  for (int u = 0; u < 45; u++) { sm += 2.0 * fabs((double)t->fm[u]); ss += 2.0 * fabs((double)t->fs[u]); }
  return 3.1415927 * sm < 7.9 && 3.1415927 * ss < 7.9;
}

/* The L+R chain of the stereo path as ONE filter (FMD_MATH_FAST_MFMA_F).  The reference low-passes the discriminator output with fm into the
 * bm ring at every sample (src/rtl_fm_player.c:545, :560) and low-passes that ring with fm again at the emit instants (:588): with no
 * non-linear step between them the two are the 179-tap filter g = fm * fm over the discriminator output.  g in double from the float taps,
 * T_g = round(g 2^qf) in three balanced int8 limbs like every filter of the matrix-pipe stages; the same bound on the weight classes
 * (accumulators read as floats) and the same error estimate as stage D's (one quantisation of the samples instead of two). */
static int build_lr_composite(const fmd_taps *t, fmdk_params *k) {
  if (k->size != 90) return -1;
  double g[179];
  composite_taps(t->fm, g);
  const int qf = taps_qf(g, 179);
  if (qf < 8) return -1;
  double sum_abs = 0.0;
  for (int u = 0; u < 179; u++) {
    const long long E = llround(g[u] * ldexp(1.0, qf));
    const unsigned q = ((unsigned)(int)E + 0x808080u) ^ 0x808080u;
    for (int l = 0; l < 3; l++) sum_abs += fabs((double)(signed char)(q >> (8 * (2 - l))));
    if (u < 90) k->gq[u] = (int32_t)E;
  }
  if (128.0 * sum_abs >= 4194304.0 - 65536.0) return -1;
  /* (the decimating form's window holds every tap of every row; round 5's full-rate form lacked the last two in two of sixteen rows and carried a term for them) */
  if (fixed_point_error(g, 179, qf, (double)k->coef).rms > FMD_STAGE_D_MAX_LSB) return -1;
  k->g_qf = qf;
  k->g_scale = (float)ldexp(1.0, 12 - qf);
  k->g_unit = (float)ldexp(1.0, -qf);
  return 0;
}

static void fill_params(fmd_batch *b) {
  fmdk_params *k = &b->kp;
  const fmd_config *c = &b->cfg;
  memset(k, 0, sizeof(*k));
  memcpy(k->fb, b->taps.fb, sizeof(k->fb));
  memcpy(k->fm, b->taps.fm, sizeof(k->fm));
  memcpy(k->fp, b->taps.fp, sizeof(k->fp));
  memcpy(k->fs, b->taps.fs, sizeof(k->fs));
  for (int j = 0; j < 127; j++) k->fm_sh[j] = b->taps.fm[j + 1];
  k->mono_2to1 = c->math != FMD_MATH_EXACT && c->mode == 1 && c->size == 128 && c->rate_out2 > 0 &&
                 c->rate_out == 2 * c->rate_out2;
  /* fast path of the /8 low-pass: y = c + sum_j s[j] (fb[min(j,31-j)] / 128) u[j]
   * with the (u - 127.5)/128 conversion folded in; s = j^n rotation signs */
  double ci = 0, cq = 0;
  for (int j = 0; j < 32; j++) {
    const float tap = b->taps.fb[j < 16 ? j : 31 - j];
    const int p = j & 3;
    float si = 1.f, sq = 1.f;
    if (!c->offset_tuning) {
      si = (p == 0 || p == 3) ? 1.f : -1.f;
      sq = (p == 0 || p == 1) ? 1.f : -1.f;
    }
    ci += (double)(si * tap);
    cq += (double)(sq * tap);
  }
  for (int j = 0; j < 16; j++) k->fbs[j] = b->taps.fb[j] / 128.0f;
  /* the kernel converts the bytes as u - 128 (small signed integers: the partial sums then stay
   * at signal level instead of carrying the 127.5 offset): (u - 127.5)/128 = (u - 128)/128 + 0.5/128 */
  k->c_i = (float)((0.5 / 128.0) * ci);
  k->c_q = (float)((0.5 / 128.0) * cq);
  k->swf = b->taps.swf;
  k->cwf = b->taps.cwf;
  k->lambda = c->deemph_lambda;
  {
    float lp = c->deemph_lambda;
    for (int j = 0; j < 16; j++) { k->lam_pow[j] = lp; lp *= c->deemph_lambda; }
  }
  if (c->math != FMD_MATH_EXACT) {
    /* per-tile flush of the fast kernels: group size and the scan's powers; with de-emphasis off
     * every power is zero and the flush passes its input through */
    const long long tile = fmdk_tile();
    /* most frames a tile can hold: floor((acc + tile slow) / fast) with acc <= fast - 1 */
    const long long fmax = c->rate_out2 > 0 ? (tile * c->rate_out2 + c->rate_out - 1) / c->rate_out : tile;
    const int ch = c->mode == 2 ? 2 : 1;
    k->flush_g = (fmax + 3) / 4 <= 64 / ch ? 4 : 8;     /* lanes: 32 groups per channel (stereo), 64 (mono) */
    if (ch == 1 && (fmax + 1) / 2 <= 64 && !tuning_env("FMD_NO_FLUSH2"))   /* (tuning builds: keep groups of four) */
      k->flush_g = 2;                                     /* mono with few frames per tile: shorter groups, fewer instructions */
    if (ch == 2 && (fmax + 2) / 3 <= 32 && !tuning_env("FMD_NO_FLUSH3"))
      k->flush_g = 3;                                     /* stereo likewise: groups of three fit its 32 lanes per channel up to 96 frames */
    const int on = c->deemph != 0;
    k->lam_eff = on ? c->deemph_lambda : 0.f;
    if (!on) memset(k->lam_pow, 0, sizeof(k->lam_pow));
    double a = on ? pow((double)c->deemph_lambda, (double)k->flush_g) : 0.0;
    k->log2_a = (on && c->deemph_lambda > 0.f) ? (float)((double)k->flush_g * log2((double)c->deemph_lambda)) : -1e30f;
    for (int j = 0; j < 8; j++) { k->lam_scan[j] = (float)a; a *= a; }
  }
  k->coef = c->volume * 32768.0f;               /* src/rtl_fm_player.c:717 */
  {
    /* origin threshold of the fast discriminator (fmdk_params.org_thr): an isolated phase error e / rho of a sample of magnitude rho reaches
     * the PCM as coef x (one tap of the filter behind the discriminator) x e / rho.  1e-3 was validated on narrow FM at volume 0.4
     * (coef x largest tap = 13 107 x 0.58 = 7 600: tests/test_gpu_parity.py::test_fast_math_nfm_noise_next_to_the_origin); a larger
     * product moves the threshold out in proportion, so that the PCM-level error at the threshold stays what it was there. */
    float hmax = 0.f;
    const float *first = (c->rate_out2 > 0 && c->mode != 0) ? b->taps.fm : NULL;      /* mode 0 / no resampler: the discriminator output goes out as it is */
    if (first) { for (int i = 0; i < (c->size >> 1); i++) hmax = fmaxf(hmax, fabsf(first[i])); if (c->mode == 2) for (int i = 0; i < (c->size >> 1); i++) hmax = fmaxf(hmax, fabsf(b->taps.fs[i])); }
    else hmax = 1.f;
    const float scale = fabsf(k->coef) * hmax / 7600.0f;
    k->org_thr = 1e-3f * (scale > 1.f ? (scale < 200.f ? scale : 200.f) : 1.f);
    k->org_thr15 = 1.5f * k->org_thr;
    k->pilot_pairs8 = fabsf(c->volume) >= 1.0f;      /* (mpx_tile_i8: eight limb pairs for the pilot filter instead of six) */
  }
  {
    /* carrier_fast: an error e in (x, y) moves sin 2 atan2 by 2 |e| / r; times |vs|, one tap of the
     * second-stage low-pass (largest |fm|) and the PCM scale it must stay below a quarter LSB.
     * |e| ~ 1.5 eps with eps = 1e-7 the rounding difference between the fast and the reference
     * pilot-filter sums  =>  r < K |vs| is redone exactly, K = 12 eps coef max|fm|.
     * Measured (tools/fuzz_parity.py 400 {1,2,3,4}, noise input): with K scaled by 0.2 and below the
     * 1 600 cases still hold 2-6 differences of 2-3 LSB, from 0.6 up none; this K is 3x that bound.
     * Noise input pays for it (every ~3rd tile holds such a sample at 300 kHz: 0.64 -> 0.83 ms per launch
     * of 256 x 16 blocks); an FM signal with a pilot never comes near (r ~ 0.06 against K |vs| ~ 0.002). */
    float gmax = 0.f;
    for (int i = 0; i < (c->size >> 1); i++) gmax = fmaxf(gmax, fabsf(b->taps.fm[i]));
    /* eps follows the rounding noise of the pilot-filter sum, ~ sqrt(sum fp^2): 1e-7 is the 300 kHz / 90-tap
     * figure (sum over the 90 taps of fp^2 = 0.0031); filters at lower rates are wider (48 kHz: 0.04-0.07), and
     * there the fuzz soak (tools/fuzz_parity.py 400 5..24) found two 2-LSB cases that needed 2-4 x this K.
     * K grows with the square of the noise ratio (capped at 25): default-rate streams keep the K above. */
    double sfp2 = 0.0;
    for (int i = 0; i < (c->size >> 1); i++) sfp2 += 2.0 * (double)b->taps.fp[i] * (double)b->taps.fp[i];
    double widen = sfp2 / 0.0031;
    if (widen < 1.0) widen = 1.0;
    if (widen > 25.0) widen = 25.0;
    float K = 12.0f * 1e-7f * (float)widen * fabsf(k->coef) * gmax;
    const char *ek = tuning_env("FMD_CARRIER_K");       /* tuning builds: override K (0 = never redo) */
    if (ek) K = (float)atof(ek);
    const char *es = tuning_env("FMD_CARRIER_SCALE");   /* ... or scale the derived K */
    if (es) K *= (float)atof(es);
    k->car_inv_k2 = K > 0.f ? 1.0f / (K * K) : 3.0e38f;
    k->car_inv_k2_q = k->car_inv_k2 < 3.0e38f * 0x1p-40f ? k->car_inv_k2 * 0x1p40f : 3.0e38f;
    /* Two levels (round 4).  A flagged sample first gets its pilot / L-R sums again from the worker's own window, in the
     * reference's ORDER of operations: that removes the order-of-summation part of the difference to the reference (what is left:
     * the window's samples are a few ulps off each, and roundings that fall differently because of it).  Only samples within
     * L K of the origin after that are recomputed from the IQ words.  L was measured like K (tools/fuzz_parity.py and the noise /
     * hand-over tests with FMD_CARRIER_L2 in a tuning build, profiles/archive/r04w_carrier_l2.txt). */
    float L2 = FMD_CARRIER_L2_DEFAULT;
    const char *el = tuning_env("FMD_CARRIER_L2");
    if (el) L2 = (float)atof(el);
    k->car_inv_k2_l2 = (K > 0.f && L2 > 0.f) ? 1.0f / (K * L2 * K * L2) : (L2 > 0.f ? 3.0e38f : 0.f);
  }
  k->size = c->size;
  k->half = c->size >> 1;
  k->mode = c->mode;
  k->slow = c->rate_out2 > 0 ? c->rate_out2 : 1;
  k->fast = c->rate_out2 > 0 ? c->rate_out : 1;
  k->inv_slow = 1.0f / (float)k->slow;      /* the estimates of the kernels' generic (no magic number) index forms */
  k->inv_fast = 1.0f / (float)k->fast;
  k->resample = c->rate_out2 > 0;
  /* floor(n / slow) for n < 2^29 as mulhi(n, m) >> sh: with l = ceil(log2 slow), p = 29 + l and
   * m = ceil(2^p / slow) the error term m slow - 2^p is below slow, so n (m slow - 2^p) < 2^p for every
   * n < 2^29 and the quotient is exact; m < 2^30 + 1 fits 32 bits.  Needs p >= 32, i.e. slow >= 5. */
  k->emit_magic = 0;
  k->emit_shift = 0;
  if (k->resample && k->slow >= 5 && (long long)k->fast * 600 < (1LL << 29)) {   /* numerators: < (frames per tile + 1) fast */
    int l = 0;
    while ((1LL << l) < k->slow) l++;
    const int p = 29 + l;
    const unsigned long long m = (((unsigned long long)1 << p) + (unsigned long long)k->slow - 1) / (unsigned long long)k->slow;
    if (p >= 32 && m <= 0xffffffffULL) { k->emit_magic = (uint32_t)m; k->emit_shift = (uint32_t)(p - 32); }
  }
  /* the same for the frames of a tile, floor((acc + tile slow) / fast): numerators below 2^30, p = 30 + l */
  k->tf_magic = 0;
  k->tf_shift = 0;
  if (k->resample && k->fast >= 5 && (long long)k->fast + (long long)fmdk_tile() * k->slow < (1LL << 30)) {
    int l = 0;
    while ((1LL << l) < k->fast) l++;
    const int p = 30 + l;
    const unsigned long long m = (((unsigned long long)1 << p) + (unsigned long long)k->fast - 1) / (unsigned long long)k->fast;
    if (p >= 32 && m <= 0xffffffffULL) { k->tf_magic = (uint32_t)m; k->tf_shift = (uint32_t)(p - 32); }
  }
  k->perm4 = k->resample && (4ll * k->fast) % k->slow == 0 && (((4ll * k->fast) / k->slow) & 1);
  k->deemph = c->deemph != 0;
  k->offset_tuning = c->offset_tuning != 0;
  /* Restart distance for the de-emphasis recurrence of the exact kernels.  A restarted trajectory is
   * within one fp32 ulp of the true one once lambda^n < 1e-7; from there a surviving 1-ulp difference
   * rounds away with probability ~ (1 - lambda) per step, i.e. survives k more steps with lambda^k.
   * lambda^warm < 1e-25 leaves 1e-18 per restart: with the 1.3 million restarts of a 256 x 16 block
   * launch, 1e-12 per launch that a carried state is one ulp off (round 1 used 1e-12: 6e-8 per restart). */
  int warm = 0;
  if (k->deemph) {
    const double lam = fabs((double)c->deemph_lambda);
    if (lam <= 0.0) warm = 1;
    else if (lam >= 1.0) warm = 1 << 30;   /* not contracting: never restart */
    else warm = (int)ceil(log(1e-25) / log(lam));
    if (warm < 16) warm = 16;
    if (warm < (1 << 29)) warm = (warm + 15) & ~15;   /* the kernel restarts in whole 16-frame blocks */
  }
  k->warm = warm;
  k->warm_fast = 0;
  if (k->deemph) {
    const double lam = fabs((double)c->deemph_lambda);
    k->warm_fast = (lam > 0.0 && lam < 1.0) ? (int)ceil(log(1e-9) / log(lam)) : warm;
    if (k->warm_fast < 1) k->warm_fast = 1;
  }
  k->block_len = c->block_len;
  k->pcm_stride = b->pcm_stride;
}

/* Configuration -> kernel family and launch parameters (b->cfg.math, b->taps, b->kp): everything fmd_batch_create decides before it touches
 * the device.  FMD_MATH_FAST and the named +-1 LSB families resolve downwards to what the configuration can run. */
static int resolve_family(fmd_batch *b, const fmd_config *cfg, const fmd_taps *taps) {
  int rc = 0;
  b->cfg = *cfg;
  /* FMD_MATH_FAST = the fastest +-1 LSB kernel family for the configuration: FMD_MATH_FAST_MFMA_F where it applies, else FMD_MATH_FAST_MFMA, else (a
   * caller's decimator taps beyond the 26-bit form) FMD_MATH_FAST_VALU.  The names of the families round 6 retired (_MFMA_C / _D / _E: include/fmdemod_mi355x.h)
   * are accepted and mean FMD_MATH_FAST.  FMD_MFMA is read by tuning builds only. */
  if (b->cfg.math == FMD_MATH_FAST || b->cfg.math == FMD_MATH_FAST_MFMA_C || b->cfg.math == FMD_MATH_FAST_MFMA_D || b->cfg.math == FMD_MATH_FAST_MFMA_E) {
    const char *e_m = tuning_env("FMD_MFMA");
    const int sel = e_m ? atoi(e_m) : 2;          /* 0: vector ALU only, 1: stage A on the matrix pipe, 2 (default): every stage that has a matrix form */
    b->cfg.math = sel == 0 ? FMD_MATH_FAST_VALU : sel == 1 ? FMD_MATH_FAST_MFMA : FMD_MATH_FAST_MFMA_F;
  }
  if (taps) b->taps = *taps;
  else if ((rc = fmd_design_taps(cfg, &b->taps))) return rc;
  b->pcm_stride = (max_result_len(cfg) + 7) & ~7;
  fill_params(b);
  if (b->cfg.math == FMD_MATH_FAST_MFMA_F) {
    /* What _MFMA_F needs; a configuration that lacks any of it runs the stage-A family, also when the caller named this one (it is a speed choice inside
     * one +-1 LSB contract).  Whole tiles (block_len a multiple of 8192 bytes) and the resampler on; the fixed-point forms of the filters fit their
     * accumulators (build_ci_scales*); rate_out >= 4 rate_out2 (stereo) / 2 rate_out2 (mono), the kernels' magic numbers exist and the second stage's
     * error estimate stays below FMD_STAGE_D_MAX_LSB (stage_d_on_matrix_pipe; stereo: the composite L+R filter's likewise, build_lr_composite); and
     * sixteen frames are a whole number P of samples, P a multiple of four (the groups' sample windows start P c - K0 bytes into the limb arrays: dword
     * reads) with the window K0 + P inside the K slices the kernels run: P <= 100 for stereo (five slices for the composite filter, three for fm),
     * 32 .. 128 for mono (four; below 64 a tile holds more than eight groups of sixteen frames: a column per group, fmdk_params.dec_wide). */
    const int whole = b->cfg.rate_out2 > 0 && (b->cfg.block_len % (16 * FMDK_TILE)) == 0;
    const long long p16 = 16LL * b->kp.fast, P = (b->kp.slow > 0 && p16 % b->kp.slow == 0) ? p16 / b->kp.slow : 0;
    int ok = 0;
    if (b->cfg.mode == 1)
      ok = whole && build_ci_scales_mono(&b->taps, b->cfg.size, &b->kp) == 0 && stage_d_on_matrix_pipe(&b->taps, &b->kp) && P % 4 == 0 && P >= 32 && P <= 128;
    else if (b->cfg.mode == 2)
      ok = whole && build_ci_scales(&b->taps, b->cfg.size, &b->kp) == 0 && stage_d_on_matrix_pipe(&b->taps, &b->kp) && P % 4 == 0 && P >= 64 && P <= 100 &&
           build_lr_composite(&b->taps, &b->kp) == 0;
    if (ok) {
      b->kp.dec_p = (int32_t)P;
      b->kp.dec_wide = b->cfg.mode == 1 && P < 64;
    } else {
      b->cfg.math = FMD_MATH_FAST_MFMA;
    }
  }
  if (b->cfg.math == FMD_MATH_FAST_MFMA || b->cfg.math == FMD_MATH_FAST_MFMA_F) {
    /* caller-supplied decimator taps too large for the 26-bit fixed-point form: the vector-ALU kernels take any taps */
    if (build_a_tab(&b->taps, b->cfg.offset_tuning != 0, &b->kp) != 0) {
      if (cfg->math == FMD_MATH_FAST_MFMA || cfg->math == FMD_MATH_FAST_MFMA_F) { return fail(FMD_E_UNSUPPORTED, "decimator taps beyond +-0.1245: FMD_MATH_FAST_MFMA needs |fb| < 2^-3.005"); }
      b->cfg.math = FMD_MATH_FAST_VALU;
    }
  }

  return FMD_OK;
}

/* The tap tables of the decimating second stage (csrc/stage_d.inc), as the kernel reads them: for byte phase r = 0 .. 15 and limb l = 0 .. 2, byte y of the
 * table = limb l of T[Y0 - r - y] (zero outside the filter), T = round(h 2^qf) in three balanced int8 limbs, Y0 = P - 1 + K0.  Stereo: the composite filter
 * (179 taps, gq, K0 180, 416 bytes per table) and behind it fm (90 taps, K0 92, 288 bytes); 128-tap mono: fm (K0 128, 368 bytes).  Returns malloc'd bytes. */
static uint8_t *build_dec_tables(const fmd_batch *b, size_t *bytes) {
  const fmdk_params *k = &b->kp;
  const int stereo = b->cfg.mode == 2, P = k->dec_p;
  const int fn = stereo ? FMDK_DF_N : FMDK_DM_N, ntaps = stereo ? 90 : 128, k0f = stereo ? FMDK_DEC_K0F : FMDK_DEC_K0M;
  const size_t gbytes = stereo ? (size_t)16 * 3 * FMDK_DG_N : 0, fbytes = (size_t)16 * 3 * (size_t)fn;
  uint8_t *t = (uint8_t *)calloc(1, gbytes + fbytes);
  if (!t) return NULL;
  for (int r = 0; r < 16; r++)
    for (int l = 0; l < 3; l++) {
      if (stereo)
        for (int y = 0; y < FMDK_DG_N; y++) {
          const int u = P - 1 + FMDK_DEC_K0G - r - y;
          if (u >= 0 && u < 179) t[((size_t)r * 3 + l) * FMDK_DG_N + y] = (uint8_t)((((uint32_t)k->gq[u < 90 ? u : 178 - u] + 0x808080u) ^ 0x808080u) >> (8 * (2 - l)));
        }
      for (int y = 0; y < fn; y++) {
        const int u = P - 1 + k0f - r - y;
        if (u >= 0 && u < ntaps) {
          const int E = (int)rintf(ldexpf(b->taps.fm[u < ntaps / 2 ? u : ntaps - 1 - u], k->ci_qf[0]));     /* (exact: |E| < 2^23) */
          t[gbytes + ((size_t)r * 3 + l) * (size_t)fn + y] = (uint8_t)((((uint32_t)E + 0x808080u) ^ 0x808080u) >> (8 * (2 - l)));
        }
      }
    }
  *bytes = gbytes + fbytes;
  return t;
}

int fmd_config_family(const fmd_config *cfg, const fmd_taps *taps) {
  int rc = check_config(cfg);
  if (rc) return rc;
  fmd_batch *b = (fmd_batch *)calloc(1, sizeof(*b));
  if (!b) return fail(FMD_E_NOMEM, "out of host memory");
  b->n_streams = 1;
  rc = resolve_family(b, cfg, taps);
  const int fam = b->cfg.math;
  free(b);
  return rc ? rc : fam;
}

int fmd_config_error_estimate(const fmd_config *cfg, const fmd_taps *taps, fmd_error_estimate *out) {
  if (!out) return fail(FMD_E_ARG, "out is NULL");
  memset(out, 0, sizeof(*out));
  int rc = check_config(cfg);
  if (rc) return rc;
  fmd_batch *b = (fmd_batch *)calloc(1, sizeof(*b));
  if (!b) return fail(FMD_E_NOMEM, "out of host memory");
  b->n_streams = 1;
  rc = resolve_family(b, cfg, taps);
  if (!rc) {
    out->family = b->cfg.math;
    out->limit_rms_lsb = (float)FMD_STAGE_D_MAX_LSB;
    double h[256], g[179];
    const double coef = (double)b->kp.coef;
    stage_error e[2];
    int n = 0;
    if (cfg->rate_out2 > 0 && cfg->mode == 2 && cfg->size == 90) {
      composite_taps(b->taps.fm, g);
      fm_full(b->taps.fm, 90, h);
      const int qg = taps_qf(g, 179), qh = taps_qf(h, 90);
      if (qg >= 8 && qh >= 8) { e[0] = fixed_point_error(g, 179, qg, coef); e[1] = fixed_point_error(h, 90, qh, coef); n = 2; }
    } else if (cfg->rate_out2 > 0 && cfg->mode == 1 && cfg->size == 128) {
      fm_full(b->taps.fm, 128, h);
      const int qh = taps_qf(h, 128);
      if (qh >= 8) { e[0] = fixed_point_error(h, 128, qh, coef); n = 1; }
    }
    out->filters = n;
    for (int i = 0; i < n; i++) {
      out->f[i].taps = e[i].n;
      out->f[i].qf = e[i].qf;
      out->f[i].rms_lsb = (float)e[i].rms;
      out->f[i].worst_samples_lsb = (float)e[i].worst_samples;
      out->f[i].worst_taps_lsb = (float)e[i].worst_taps;
      out->f[i].worst_dropped_lsb = (float)e[i].worst_dropped;
      out->f[i].worst_lsb = (float)(e[i].worst_samples + e[i].worst_taps + e[i].worst_dropped);
    }
  }
  free(b);
  return rc;
}

int fmd_batch_create(fmd_batch **out, const fmd_config *cfg, const fmd_taps *taps, int n_streams,
                     int device) {
  if (!out) return fail(FMD_E_ARG, "out is NULL");
  *out = NULL;
  int rc = check_config(cfg);
  if (rc) return rc;
  if (n_streams <= 0) return fail(FMD_E_ARG, "n_streams must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(FMD_E_NODEVICE, "no HIP device: the MI355X path has no CPU fallback");
  if (device < 0) HIP_TRY(hipGetDevice(&device));
  if (device >= ndev) return fail(FMD_E_ARG, "device %d out of range (%d devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));

  fmd_batch *b = (fmd_batch *)calloc(1, sizeof(*b));
  if (!b) return fail(FMD_E_NOMEM, "out of host memory");
  b->n_streams = n_streams;
  b->device = device;
  if ((rc = resolve_family(b, cfg, taps))) { free(b); return rc; }

  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&b->ev0)) != hipSuccess || (e = hipEventCreate(&b->ev1)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&b->ev_order, hipEventDisableTiming)) != hipSuccess ||
      (e = hipMalloc(&b->d_state[0], sizeof(fmd_stream_state) * (size_t)n_streams)) != hipSuccess ||
      (e = hipMalloc(&b->d_state[1], sizeof(fmd_stream_state) * (size_t)n_streams)) != hipSuccess ||
      (e = hipMemsetAsync(b->d_state[0], 0, sizeof(fmd_stream_state) * (size_t)n_streams, b->stream)) != hipSuccess ||
      (e = hipMemsetAsync(b->d_state[1], 0, sizeof(fmd_stream_state) * (size_t)n_streams, b->stream)) != hipSuccess ||
      (e = hipStreamSynchronize(b->stream)) != hipSuccess) {
    rc = fail(FMD_E_HIP, "device setup failed: %s", hipGetErrorString(e));
    fmd_batch_destroy(b);
    return rc;
  }
  if (b->cfg.math == FMD_MATH_FAST_MFMA_F && b->kp.dec_p > 0) {
    size_t nb = 0;
    uint8_t *t = build_dec_tables(b, &nb);
    if (!t) { fmd_batch_destroy(b); return fail(FMD_E_NOMEM, "out of host memory"); }
    e = hipMalloc(&b->d_dec_tables, nb);
    if (e == hipSuccess) e = hipMemcpy(b->d_dec_tables, t, nb, hipMemcpyHostToDevice);
    free(t);
    if (e != hipSuccess) { rc = fail(FMD_E_HIP, "device setup failed: %s", hipGetErrorString(e)); fmd_batch_destroy(b); return rc; }
    b->kp.dec_tables = b->d_dec_tables;
  }
  {
    hipDeviceProp_t prop;
    b->n_cus = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
  }
  b->ingest = (struct fmd_ingest **)calloc((size_t)n_streams, sizeof(*b->ingest));
  if (!b->ingest) { fmd_batch_destroy(b); return fail(FMD_E_NOMEM, "out of host memory"); }
  *out = b;
  return FMD_OK;
}

/* Wait for everything this batch has queued: its own stream, the copy stream of the pump and
 * the stream of the most recent launch when the caller supplied one. */
static hipError_t batch_quiesce(fmd_batch *b) {
  hipError_t e = hipSuccess, t;
  if (b->launched && b->last_stream && b->last_stream != b->stream &&
      (t = hipStreamSynchronize(b->last_stream)) != hipSuccess) e = t;
  if (b->copy_stream && (t = hipStreamSynchronize(b->copy_stream)) != hipSuccess) e = t;
  if (b->stream && (t = hipStreamSynchronize(b->stream)) != hipSuccess) e = t;
  if (e == hipSuccess) b->launched = 0;        /* nothing of this batch is in flight: the next launch needs no hand-over, whatever stream it is on */
  return e;
}

static void ingest_detach(struct fmd_ingest *g);

void fmd_batch_destroy(fmd_batch *b) {
  if (!b) return;
  hipSetDevice(b->device);
  batch_quiesce(b);
  /* rings outlive the batch (their owner destroys them with fmd_ingest_destroy, before or
   * after this call): detach them so that neither side touches freed memory */
  if (b->ingest)
    for (int i = 0; i < b->n_streams; i++)
      if (b->ingest[i]) ingest_detach(b->ingest[i]);
  if (b->d_dec_tables) hipFree(b->d_dec_tables);
  if (b->d_state[0]) hipFree(b->d_state[0]);
  if (b->d_state[1]) hipFree(b->d_state[1]);
  if (b->d_iq) hipFree(b->d_iq);
  if (b->d_pcm) hipFree(b->d_pcm);
  if (b->d_lens) hipFree(b->d_lens);
  for (int i = 0; i < 2; i++) {
    struct pump_slot *p = &b->pump[i];
    if (p->h_pcm) hipHostFree(p->h_pcm);
    if (p->h_lens) hipHostFree(p->h_lens);
    if (p->d_iq) hipFree(p->d_iq);
    if (p->d_pcm) hipFree(p->d_pcm);
    if (p->d_lens) hipFree(p->d_lens);
    if (p->h2d_done) hipEventDestroy(p->h2d_done);
    if (p->done) hipEventDestroy(p->done);
  }
  if (b->copy_stream) hipStreamDestroy(b->copy_stream);
  if (b->ev0) hipEventDestroy(b->ev0);
  if (b->ev1) hipEventDestroy(b->ev1);
  if (b->ev_order) hipEventDestroy(b->ev_order);
  if (b->stream) hipStreamDestroy(b->stream);
  free(b->ingest);
  free(b);
}

int fmd_batch_pcm_stride(const fmd_batch *b) { return b ? b->pcm_stride : FMD_E_ARG; }
int fmd_batch_n_streams(const fmd_batch *b) { return b ? b->n_streams : FMD_E_ARG; }
int fmd_batch_math(const fmd_batch *b) { return b ? b->cfg.math : FMD_E_ARG; }
int fmd_batch_set_time_split(fmd_batch *b, int workers_per_cu) {
  if (!b) return fail(FMD_E_ARG, "NULL batch");
  b->time_split = workers_per_cu;
  return FMD_OK;
}
const char *fmd_batch_kernel_name(const fmd_batch *b) {
  return b ? fmdk_kernel_name(&b->kp, b->cfg.math) : "";
}

int fmd_batch_run_device_debug(fmd_batch *b, const void *d_iq, int n_blocks, void *d_pcm, void *d_lens,
                               void *hip_stream, const fmd_debug_taps *dbg) {
  if (!b || !d_iq || !d_pcm || !d_lens) return fail(FMD_E_ARG, "NULL argument");
  if (n_blocks < 0) return fail(FMD_E_ARG, "n_blocks < 0");
  if (n_blocks == 0) return FMD_OK;
  if (((uintptr_t)d_iq & 15) != 0) return fail(FMD_E_ARG, "d_iq must be 16-byte aligned");
  if ((long long)b->cfg.block_len * n_blocks >= (1LL << 32))   /* one raw buffer (32-bit size) per stream */
    return fail(FMD_E_ARG, "n_blocks too large: block_len * n_blocks must stay below 2^32 bytes per stream");
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : b->stream;
  fmdk_params kp = b->kp;
  kp.n_blocks = n_blocks;
  /* Cut each stream's tiles into time chunks until the grid offers enough
   * workers (wavefronts) per CU; each chunk > 0 replays warm_tiles tiles first
   * (see the kernel), so keep chunks at least 4x longer than the replay.  (8x until round 3: a one-block launch of
   * 256 streams then ran as four chunks per stream = ONE wave per SIMD, which takes 7 us per tile with nobody to hide its
   * latencies under - 0.094 ms; eight chunks of 4 + 1 tiles, two waves per SIMD: profiles/archive/r03y_blocks_per_launch.txt) */
  kp.n_streams = b->n_streams;
  kp.warm_tiles = fmdk_warm_tiles(&kp, b->cfg.math);
  kp.n_chunks = 1;
  if (kp.warm_tiles > 0 && b->time_split >= 0) {
    const int per_cu = b->time_split > 0 ? b->time_split
                                         : fmdk_workers_per_cu_mode(b->cfg.math, b->cfg.rate_out2 > 0 ? b->cfg.mode : 0);
    const long long m = kp.block_len >> 4, tile = fmdk_tile();
    const long long tiles = ((m + tile - 1) / tile) * n_blocks;
    long long want = ((long long)per_cu * b->n_cus + b->n_streams - 1) / b->n_streams;
    /* short launches: when three workers per SIMD would leave chunks under six replays' length, two per SIMD with longer
     * chunks are faster (stereo, 2 blocks x 256 streams: 0.100 ms against 0.110) */
    if (b->time_split == 0 && per_cu >= 12 && tiles < 6LL * kp.warm_tiles * want) {   /* (kernels budgeted for two per SIMD already are) */
      const long long want2 = ((long long)(per_cu - per_cu / 3) * b->n_cus + b->n_streams - 1) / b->n_streams;
      if (want2 < want) want = want2;
    }
    const long long most = tiles / (4LL * kp.warm_tiles);
    if (want > most) want = most;
    if (want > 1) kp.n_chunks = (int)want;
  }
  /* The state is always ping-ponged (the kernel's in / out pointers never alias).  Launches on one
   * stream are ordered by the stream; when the stream changes between two launches an event makes
   * the new stream wait for the previous launch, whose output state this one reads. */
  /* A caller's stream that is being captured into a hipGraph: the launch becomes a node of the graph.  The timing events have no meaning there (the launch
   * carries none and fmd_batch_last_kernel_ms says so afterwards), and the event hand-over between streams would record an event of the batch on a stream
   * outside the capture - which invalidates the capture: refused, with what to do instead. */
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (st && hipStreamIsCapturing(st, &cap) != hipSuccess) { cap = hipStreamCaptureStatusNone; (void)hipGetLastError(); }
  const int capturing = cap != hipStreamCaptureStatusNone;
  if (b->launched && b->last_stream != st) {
    if (capturing)
      return fail(FMD_E_STATE, "the batch's previous launch ran on another stream: call fmd_batch_sync() before capturing this one into a graph (an event "
                               "hand-over between streams cannot be recorded inside a capture)");
    HIP_TRY(hipEventRecord(b->ev_order, b->last_stream));
    HIP_TRY(hipStreamWaitEvent(st, b->ev_order, 0));
  }
  const int nxt = b->cur ^ 1;
  const int with_events = !b->no_timing && !capturing;
  /* the timing events ride on the kernel's dispatch packet (fmdk_launch): no packets of their own */
  int e = fmdk_launch(&kp, b->cfg.math, b->n_streams, d_iq, d_pcm, d_lens, b->d_state[b->cur],
                      b->d_state[nxt], dbg, st, with_events ? (void *)b->ev0 : NULL, with_events ? (void *)b->ev1 : NULL);
  if (e) return fail(FMD_E_HIP, "kernel launch failed: %s (%d)", hipGetErrorString((hipError_t)e), e);
  b->cur = nxt;
  b->last_stream = st;
  b->launched = 1;
  b->timed = with_events;           /* (a captured launch has no events: fmd_batch_last_kernel_ms then reports FMD_E_STATE instead of a stale time) */
  return FMD_OK;
}

int fmd_batch_run_device(fmd_batch *b, const void *d_iq, int n_blocks, void *d_pcm, void *d_lens,
                         void *hip_stream) {
  return fmd_batch_run_device_debug(b, d_iq, n_blocks, d_pcm, d_lens, hip_stream, NULL);
}

int fmd_batch_sync(fmd_batch *b) {
  if (!b) return fail(FMD_E_ARG, "NULL batch");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(batch_quiesce(b));
  return FMD_OK;
}

int fmd_batch_wait_stream(fmd_batch *b, void *producer_stream) {
  if (!b) return fail(FMD_E_ARG, "NULL batch");
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t ps = (hipStream_t)producer_stream;
  if (ps == b->stream) return FMD_OK;
  hipEvent_t ev;
  HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, ps);
  if (e == hipSuccess) e = hipStreamWaitEvent(b->stream, ev, 0);
  hipEventDestroy(ev);                         /* (released by the runtime once the recorded work has completed) */
  if (e != hipSuccess) return fail(FMD_E_HIP, "fmd_batch_wait_stream: %s", hipGetErrorString(e));
  return FMD_OK;
}

int fmd_batch_last_kernel_ms(fmd_batch *b, float *ms) {
  if (!b || !ms) return fail(FMD_E_ARG, "NULL argument");
  if (!b->timed) return fail(FMD_E_STATE, "the most recent launch carries no timing events (none launched yet, timing off, or captured into a graph)");
  HIP_TRY(hipEventSynchronize(b->ev1));
  HIP_TRY(hipEventElapsedTime(ms, b->ev0, b->ev1));
  return FMD_OK;
}

int fmd_batch_set_timing(fmd_batch *b, int on) {
  if (!b) return fail(FMD_E_ARG, "NULL batch");
  b->no_timing = !on;
  if (!on) b->timed = 0;
  return FMD_OK;
}

static int ensure_staging(fmd_batch *b, int n_blocks) {
  if ((size_t)n_blocks <= b->cap_blocks) return FMD_OK;
  if (b->d_iq) hipFree(b->d_iq);
  if (b->d_pcm) hipFree(b->d_pcm);
  if (b->d_lens) hipFree(b->d_lens);
  b->d_iq = b->d_pcm = b->d_lens = NULL;
  b->cap_blocks = 0;
  const size_t slots = (size_t)b->n_streams * (size_t)n_blocks;
  HIP_TRY(hipMalloc(&b->d_iq, slots * (size_t)b->cfg.block_len));
  HIP_TRY(hipMalloc(&b->d_pcm, slots * (size_t)b->pcm_stride * sizeof(int16_t)));
  HIP_TRY(hipMalloc(&b->d_lens, slots * sizeof(int32_t)));
  b->cap_blocks = (size_t)n_blocks;
  return FMD_OK;
}

int fmd_batch_run_host(fmd_batch *b, const uint8_t *iq, int n_blocks, int16_t *pcm, int32_t *lens) {
  if (!b || !iq || !pcm || !lens) return fail(FMD_E_ARG, "NULL argument");
  if (n_blocks <= 0) return fail(FMD_E_ARG, "n_blocks must be positive");
  HIP_TRY(hipSetDevice(b->device));
  int rc = ensure_staging(b, n_blocks);
  if (rc) return rc;
  const size_t slots = (size_t)b->n_streams * (size_t)n_blocks;
  HIP_TRY(hipMemcpyAsync(b->d_iq, iq, slots * (size_t)b->cfg.block_len, hipMemcpyHostToDevice, b->stream));
  rc = fmd_batch_run_device(b, b->d_iq, n_blocks, b->d_pcm, b->d_lens, NULL);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(pcm, b->d_pcm, slots * (size_t)b->pcm_stride * sizeof(int16_t),
                         hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipMemcpyAsync(lens, b->d_lens, slots * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return FMD_OK;
}

int fmd_batch_get_state(fmd_batch *b, int stream, fmd_stream_state *out) {
  if (!b || !out || stream < 0 || stream >= b->n_streams) return fail(FMD_E_ARG, "bad argument");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(batch_quiesce(b));
  HIP_TRY(hipMemcpy(out, (char *)b->d_state[b->cur] + sizeof(*out) * (size_t)stream, sizeof(*out),
                    hipMemcpyDeviceToHost));
  return FMD_OK;
}

int fmd_batch_set_state(fmd_batch *b, int stream, const fmd_stream_state *in) {
  if (!b || !in || stream < 0 || stream >= b->n_streams) return fail(FMD_E_ARG, "bad argument");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(batch_quiesce(b));
  HIP_TRY(hipMemcpy((char *)b->d_state[b->cur] + sizeof(*in) * (size_t)stream, in, sizeof(*in),
                    hipMemcpyHostToDevice));
  return FMD_OK;
}

int fmd_batch_reset(fmd_batch *b) {
  if (!b) return fail(FMD_E_ARG, "NULL batch");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(batch_quiesce(b));
  /* on the batch's own stream and waited for: hipMemset on device memory may return before the fill has run, and the
   * batch's stream (non-blocking) does not order itself behind the null stream */
  HIP_TRY(hipMemsetAsync(b->d_state[b->cur], 0, sizeof(fmd_stream_state) * (size_t)b->n_streams, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return FMD_OK;
}

/* ---- reference-shaped surface --------------------------------------------- */
/*
 * One single-stream batch per demod_state, found through a small registry
 * keyed by the struct's address (the reference struct has no spare pointer
 * field).  The batch is (re)built when the parameters that shape the kernels
 * change.  State lives in the struct between calls, like in the reference.
 */
struct drop_in {
  struct demod_state *key;
  fmd_batch *batch;
  fmd_config cfg;
  int convert_mode;   /* 0: rotate_90_u8_f32, 1: u8_f32 */
  /* One synchronisation per block (round 4).  The carried state stays on the device; `shadow` is what the last call
   * mirrored into the struct (linear histories, as the device keeps them).  A call whose struct still holds exactly
   * that skips the state upload; a caller that edited the struct's state between calls (the reference allows it: it is
   * a plain struct) is noticed by the comparison and gets its values uploaded.  `pin` is one pinned block the PCM, the
   * block length and the new state land in, behind a single hipStreamSynchronize. */
  fmd_stream_state shadow;
  int shadow_valid;
  int shadow_pos;     /* lpr.pos the shadow's linear histories correspond to */
  struct dropin_pin { fmd_stream_state st; int32_t len; int32_t pad[3]; int16_t pcm[]; } *pin;
  size_t pin_pcm;     /* int16 capacity of pin->pcm */
};

/* The registry: heap nodes behind a growing array of pointers (a node's address is stable for as long as its struct is registered; rounds 1 - 5 had 64
 * fixed slots and aborted on the 65th struct). */
static struct drop_in **g_drop;
static int g_drop_n, g_drop_cap;
static pthread_mutex_t g_drop_m = PTHREAD_MUTEX_INITIALIZER;

static struct drop_in *drop_find(struct demod_state *d, int create) {
  struct drop_in *hit = NULL;
  int empty = -1;
  pthread_mutex_lock(&g_drop_m);
  for (int i = 0; i < g_drop_n; i++) {
    if (g_drop[i] && g_drop[i]->key == d) { hit = g_drop[i]; break; }
    if (!g_drop[i] && empty < 0) empty = i;
  }
  if (!hit && create) {
    if (empty < 0 && g_drop_n == g_drop_cap) {
      const int cap = g_drop_cap ? 2 * g_drop_cap : 16;
      struct drop_in **g = (struct drop_in **)realloc(g_drop, (size_t)cap * sizeof(*g));
      if (g) { g_drop = g; g_drop_cap = cap; }
    }
    if (empty < 0 && g_drop_n < g_drop_cap) { empty = g_drop_n++; g_drop[empty] = NULL; }
    if (empty >= 0 && (hit = (struct drop_in *)calloc(1, sizeof(*hit)))) {
      hit->key = d;
      g_drop[empty] = hit;
    }
  }
  pthread_mutex_unlock(&g_drop_m);
  return hit;
}
static void drop_forget(struct drop_in *di) {
  pthread_mutex_lock(&g_drop_m);
  for (int i = 0; i < g_drop_n; i++)
    if (g_drop[i] == di) g_drop[i] = NULL;
  pthread_mutex_unlock(&g_drop_m);
  free(di);
}

/* The arithmetic family of the reference-shaped calls, whose signatures have no room for it: fmd_dropin_set_math() if the caller said so, else FMD_MATH_FAST in
 * the environment selects the +-1 LSB kernels (read once, at the first full_demod), else the bit-exact ones. */
static int g_dropin_math = -1, g_dropin_math_set = 0;
static pthread_once_t g_dropin_once = PTHREAD_ONCE_INIT;
static void dropin_read_env(void) { if (!g_dropin_math_set) g_dropin_math = getenv("FMD_MATH_FAST") ? FMD_MATH_FAST : FMD_MATH_EXACT; }
int fmd_dropin_set_math(int math) {
  if (math < FMD_MATH_EXACT || math > FMD_MATH_FAST_MFMA_F) return fail(FMD_E_ARG, "fmd_dropin_set_math: not a math value");
  g_dropin_math = math;
  g_dropin_math_set = 1;
  return FMD_OK;
}

/* What a failure inside a void reference-shaped call does: the caller's handler if one is installed (the call then returns with result_len = 0), else a line
 * on stderr and abort() - a demodulator that silently stops producing audio is the worse failure for the program this drops into. */
static fmd_dropin_error_fn g_dropin_err;
static void *g_dropin_err_ctx;
void fmd_dropin_set_error_handler(fmd_dropin_error_fn fn, void *ctx) { g_dropin_err = fn; g_dropin_err_ctx = ctx; }
static void die(const char *what) {
  if (g_dropin_err) { g_dropin_err(what, fmd_last_error(), g_dropin_err_ctx); return; }
  fprintf(stderr, "fmdemod_mi355x: %s: %s\n", what, fmd_last_error());
  abort();
}
#define DIE(d, what) do { die(what); if (d) (d)->result_len = 0; return; } while (0)

void init_u8_f32_table(void) {}  /* the conversion is arithmetic on the device (exact, no table) */
void init_lp_f32(void) {}        /* taps are designed per batch in fmd_design_taps              */

void demod_init(struct demod_state *s) {   /* src/rtl_fm_player.c:1156-1195 */
  s->rate_in = 240000;
  s->rate_out = 240000;
  s->squelch_level = 0;
  s->conseq_squelch = 10;
  s->terminate_on_squelch = 0;
  s->squelch_hits = 11;
  s->downsample_passes = 0;
  s->comp_fir_size = 0;
  s->prev_index = 0;
  s->post_downsample = 1;
  s->custom_atan = 1;
  s->deemph = 0.000050;
  s->offset_tuning = 0;
  s->rate_out2 = 48000;
  s->pre_j = s->pre_r = s->now_r = s->now_j = 0;
  s->pre_j_f32 = s->pre_r_f32 = 0;
  s->prev_lpr_index = 0;
  s->deemph_a = 0;
  s->deemph_l = 0;
  s->deemph_r = 0;
  s->deemph_l_f32 = 0;
  s->deemph_r_f32 = 0;
  s->volume = 0.4f;
  s->now_lpr = 0;
  s->lpr.mode = 2;
  s->lpr.size = 90;
  s->lpr.br = s->lpr.bm = s->lpr.bs = NULL;
  s->lpr.fm = s->lpr.fp = s->lpr.fs = NULL;
  pthread_rwlock_init(&s->rw, NULL);
  pthread_cond_init(&s->ready, NULL);
  pthread_mutex_init(&s->ready_m, NULL);
  s->output_target = NULL;
}

void init_lp_real_f32(struct demod_state *fm) {   /* src/rtl_fm_player.c:413-453 */
  struct lp_real *l = &fm->lpr;
  l->rsize = l->size >> 1;
  l->pp = 0;
  l->pos = 0;
  l->br = (float *)calloc((size_t)l->size, 4);
  l->bm = (float *)calloc((size_t)l->size, 4);
  l->bs = (float *)calloc((size_t)l->size, 4);
  l->fm = (float *)calloc((size_t)l->rsize, 4);
  l->fp = (float *)calloc((size_t)l->rsize, 4);
  l->fs = (float *)calloc((size_t)l->rsize, 4);
  design_mpx(l->size, fm->rate_in, l->fm, l->fp, l->fs, &l->swf, &l->cwf);
}

void fmd_demod_release(struct demod_state *d) {
  struct drop_in *di = drop_find(d, 0);
  if (!di) return;
  fmd_batch_destroy(di->batch);
  if (di->pin) hipHostFree(di->pin);
  drop_forget(di);
}

void deinit_lp_real_f32(struct demod_state *fm) {   /* src/rtl_fm_player.c:455-470 */
  struct lp_real *l = &fm->lpr;
  fmd_demod_release(fm);
  l->rsize = 0;
  free(l->br); free(l->bm); free(l->bs); free(l->fm); free(l->fp); free(l->fs);
  l->br = l->bm = l->bs = l->fm = l->fp = l->fs = NULL;
}

void rotate_90_u8_f32(struct demod_state *d) {   /* src/rtl_fm_player.c:206-226 */
  struct drop_in *di = drop_find(d, 1);
  if (!di) { fail(FMD_E_NOMEM, "out of host memory"); DIE(d, "rotate_90_u8_f32"); }
  di->convert_mode = 0;
  d->lp_len = (int)d->buf_len;
}

void u8_f32(struct demod_state *d) {             /* src/rtl_fm_player.c:228-239 */
  struct drop_in *di = drop_find(d, 1);
  if (!di) { fail(FMD_E_NOMEM, "out of host memory"); DIE(d, "u8_f32"); }
  di->convert_mode = 1;
  d->lp_len = (int)d->buf_len;
}

/* ring (write index pos) -> linear oldest-first */
static void ring_to_linear(const float *ring, int size, int pos, float *lin) {
  for (int i = 0; i < size; i++) lin[i] = ring[(pos + i) % size];
}
static void linear_to_ring(const float *lin, int size, int pos, float *ring) {
  for (int i = 0; i < size; i++) ring[(pos + i) % size] = lin[i];
}

void full_demod(struct demod_state *d) {         /* src/rtl_fm_player.c:758-788 */
  struct drop_in *di = drop_find(d, 1);
  if (!di) { fail(FMD_E_NOMEM, "out of host memory"); DIE(d, "full_demod"); }
  if (!d->lpr.br || !d->lpr.fm) { fail(FMD_E_STATE, "init_lp_real_f32 was not called"); DIE(d, "full_demod"); }
  pthread_once(&g_dropin_once, dropin_read_env);
  const int math = g_dropin_math;
  fmd_config c = {d->rate_in, d->rate_out, d->rate_out2, d->lpr.mode, d->lpr.size, d->deemph != 0.0,
                  di->convert_mode, d->deemph_lambda, d->volume, (int32_t)d->buf_len, math};
  if (!di->batch || memcmp(&c, &di->cfg, sizeof(c)) != 0) {
    fmd_batch_destroy(di->batch);
    di->batch = NULL;
    fmd_taps t;
    memset(&t, 0, sizeof(t));
    design_fb(t.fb);
    memcpy(t.fm, d->lpr.fm, sizeof(float) * (size_t)(d->lpr.size >> 1));   /* the caller's own tables */
    memcpy(t.fp, d->lpr.fp, sizeof(float) * (size_t)(d->lpr.size >> 1));
    memcpy(t.fs, d->lpr.fs, sizeof(float) * (size_t)(d->lpr.size >> 1));
    t.swf = d->lpr.swf;
    t.cwf = d->lpr.cwf;
    if (fmd_batch_create(&di->batch, &c, &t, 1, -1)) DIE(d, "full_demod: fmd_batch_create");
    di->cfg = c;
    /* a new batch starts from zeroed device state: what the struct holds (the stream so far - the reference keeps running across a
     * change of volume, buf_len, rates or de-emphasis, all of which are in fmd_config) must be uploaded whatever the shadow says */
    di->shadow_valid = 0;
  }
  fmd_batch *b = di->batch;
  const int size = d->lpr.size;

  /* struct -> device state: only when the struct does not hold what the last call left in it */
  fmd_stream_state st;
  memset(&st, 0, sizeof(st));
  memcpy(st.tb, d->lowpass_tb, sizeof(st.tb));
  st.pre_r = d->pre_r_f32;
  st.pre_j = d->pre_j_f32;
  st.pp = d->lpr.pp;
  st.deemph_l = d->deemph_l_f32;
  st.deemph_r = d->deemph_r_f32;
  st.acc = d->prev_lpr_index;
  ring_to_linear(d->lpr.br, size, d->lpr.pos, st.br);
  ring_to_linear(d->lpr.bm, size, d->lpr.pos, st.bm);
  ring_to_linear(d->lpr.bs, size, d->lpr.pos, st.bs);
  const int upload = !(di->shadow_valid && di->shadow_pos == d->lpr.pos && memcmp(&st, &di->shadow, sizeof(st)) == 0);

  if (hipSetDevice(b->device) != hipSuccess) { fail(FMD_E_HIP, "hipSetDevice failed"); DIE(d, "full_demod"); }
  if (ensure_staging(b, 1)) DIE(d, "full_demod: staging");
  if (!di->pin || di->pin_pcm < (size_t)b->pcm_stride) {
    if (di->pin) hipHostFree(di->pin);
    di->pin = NULL;
    if (hipHostMalloc((void **)&di->pin, sizeof(*di->pin) + sizeof(int16_t) * (size_t)b->pcm_stride, hipHostMallocDefault) != hipSuccess) {
      fail(FMD_E_NOMEM, "pinned staging for the drop-in surface");
      DIE(d, "full_demod");
    }
    di->pin_pcm = (size_t)b->pcm_stride;
  }
  hipError_t e = hipSuccess;
  if (upload) {
    /* (a caller-edited state, or the first block: everything queued on the batch's own stream, in order) */
    if (batch_quiesce(b) != hipSuccess) { fail(FMD_E_HIP, "device busy"); DIE(d, "full_demod"); }
    di->pin->st = st;
    e = hipMemcpyAsync(b->d_state[b->cur], &di->pin->st, sizeof(st), hipMemcpyHostToDevice, b->stream);
    if (e != hipSuccess) { fail(FMD_E_HIP, "state upload: %s", hipGetErrorString(e)); DIE(d, "full_demod"); }
    if (hipStreamSynchronize(b->stream) != hipSuccess) { fail(FMD_E_HIP, "state upload"); DIE(d, "full_demod"); }   /* pin->st is reused below */
  }
  e = hipMemcpyAsync(b->d_iq, d->buf, (size_t)d->buf_len, hipMemcpyHostToDevice, b->stream);
  if (e != hipSuccess) { fail(FMD_E_HIP, "IQ upload: %s", hipGetErrorString(e)); DIE(d, "full_demod"); }
  if (fmd_batch_run_device(b, b->d_iq, 1, b->d_pcm, b->d_lens, NULL)) DIE(d, "full_demod: run");
  if ((e = hipMemcpyAsync(di->pin->pcm, b->d_pcm, sizeof(int16_t) * (size_t)b->pcm_stride, hipMemcpyDeviceToHost, b->stream)) != hipSuccess ||
      (e = hipMemcpyAsync(&di->pin->len, b->d_lens, sizeof(int32_t), hipMemcpyDeviceToHost, b->stream)) != hipSuccess ||
      (e = hipMemcpyAsync(&di->pin->st, b->d_state[b->cur], sizeof(st), hipMemcpyDeviceToHost, b->stream)) != hipSuccess ||
      (e = hipStreamSynchronize(b->stream)) != hipSuccess) {                       /* the one wait of the block */
    fail(FMD_E_HIP, "full_demod: %s", hipGetErrorString(e));
    DIE(d, "full_demod");
  }
  const int32_t len = di->pin->len;
  memcpy(d->result, di->pin->pcm, sizeof(int16_t) * (size_t)(len > 0 ? len : 0));
  d->result_len = len;
  d->lp_len = (int)d->buf_len >> 3;                       /* :410 */

  /* device state -> struct (the reference keeps its state there; callers may read it) */
  st = di->pin->st;
  memcpy(d->lowpass_tb, st.tb, sizeof(st.tb));
  d->pre_r_f32 = st.pre_r;
  d->pre_j_f32 = st.pre_j;
  d->deemph_l_f32 = st.deemph_l;
  d->deemph_r_f32 = st.deemph_r;
  d->prev_lpr_index = st.acc;
  if (d->rate_out2 > 0 && d->lpr.mode != 0) {
    const int m = (int)(d->buf_len >> 4);
    const int pos = (d->lpr.pos + m) % size;
    linear_to_ring(st.br, size, pos, d->lpr.br);
    if (d->lpr.mode == 2) {
      linear_to_ring(st.bm, size, pos, d->lpr.bm);
      linear_to_ring(st.bs, size, pos, d->lpr.bs);
      d->lpr.pp = st.pp;
    }
    d->lpr.pos = pos;
  }
  /* what the struct holds now, as the next call will read it back: the fields a mode does not mirror keep the struct's values */
  {
    fmd_stream_state sh;
    memset(&sh, 0, sizeof(sh));
    memcpy(sh.tb, d->lowpass_tb, sizeof(sh.tb));
    sh.pre_r = d->pre_r_f32; sh.pre_j = d->pre_j_f32; sh.pp = d->lpr.pp;
    sh.deemph_l = d->deemph_l_f32; sh.deemph_r = d->deemph_r_f32; sh.acc = d->prev_lpr_index;
    ring_to_linear(d->lpr.br, size, d->lpr.pos, sh.br);
    ring_to_linear(d->lpr.bm, size, d->lpr.pos, sh.bm);
    ring_to_linear(d->lpr.bs, size, d->lpr.pos, sh.bs);
    /* the device's state and the struct's view of it agree exactly when every mirrored field went both ways */
    di->shadow = sh;
    di->shadow_pos = d->lpr.pos;
    di->shadow_valid = memcmp(&sh, &st, sizeof(sh)) == 0;
  }
}

/* ---- ingest ---------------------------------------------------------------- */
/*
 * One pinned ring per stream, written by fmd_ingest_callback (any thread: librtlsdr's USB event
 * thread in the reference, src/rtl_fm_player.c:839-853) and drained by the pump (the demod thread's
 * role, :855-933).  The ring IS the H2D source: a job's bytes are copied to the device straight from
 * the ring (no second host copy) and stay accounted as buffered until that copy has finished.
 *
 * Accounting (all under g->m):
 *   rpos      oldest byte not yet released        size      bytes in [rpos, rpos + size) (mod cap)
 *   inflight  leading bytes of that range handed to jobs whose H2D may still be reading them
 *   wpos      next write position
 * Two overflow behaviours (fmd_ingest_set_overflow):
 *   FMD_OVERFLOW_DROP_OLDEST (default)  the copy wraps at the end of the ring; bytes beyond the capacity
 *       push rpos forward (the oldest data is lost, counted in `dropped`).  A clean loss.
 *   FMD_OVERFLOW_REFERENCE  rtlsdr_callback to the letter (:813-834): a transfer that does not fit
 *       before the end of the ring restarts at offset 0 (no split copy; whatever lies between wpos and
 *       the end is left as it is), and on overflow only the byte count is clamped - rpos stays, so the
 *       reader next sees new data where it expected old.  Kept for identical behaviour
 *       (tests/test_ring_ref.py holds it against the reference's own callback).
 */
struct fmd_ingest {
  fmd_batch *batch;       /* NULL once the batch has been destroyed */
  int stream;
  uint8_t *ring;          /* pinned host memory */
  uint32_t cap, rpos, wpos, size, inflight;
  uint32_t debt;          /* in-flight bytes an overflow has already released (drop-oldest) */
  uint64_t dropped;
  int mute;
  int overflow_mode;
  int unbound;            /* created without a batch: ring in pageable memory */
  pthread_mutex_t m;
};

int fmd_ingest_create(fmd_ingest **out, fmd_batch *b, int stream, uint32_t ring_bytes) {
  if (!out) return fail(FMD_E_ARG, "bad argument");
  *out = NULL;
  if (ring_bytes == 0) ring_bytes = 16u * FMD_MAXIMUM_BUF_LENGTH;   /* include/rtl_fm_player.h:65 */
  fmd_ingest *g;
  if (!b) {
    /* unbound ring: plain host memory, no device involved; drained with fmd_ingest_pop */
    g = (fmd_ingest *)calloc(1, sizeof(*g));
    if (!g) return fail(FMD_E_NOMEM, "out of host memory");
    g->ring = (uint8_t *)calloc(ring_bytes, 1);       /* zero like the reference's static _input_buffer */
    if (!g->ring) { free(g); return fail(FMD_E_NOMEM, "out of host memory"); }
    g->unbound = 1;
    g->stream = -1;
    g->cap = ring_bytes;
    g->overflow_mode = FMD_OVERFLOW_DROP_OLDEST;
    pthread_mutex_init(&g->m, NULL);
    *out = g;
    return FMD_OK;
  }
  if (stream < 0 || stream >= b->n_streams) return fail(FMD_E_ARG, "bad argument");
  if (b->ingest[stream]) return fail(FMD_E_STATE, "stream %d already has an ingest ring", stream);
  if (ring_bytes < (uint32_t)b->cfg.block_len) return fail(FMD_E_ARG, "ring smaller than one block");
  if (hipSetDevice(b->device) != hipSuccess) return fail(FMD_E_HIP, "hipSetDevice(%d) failed", b->device);
  g = (fmd_ingest *)calloc(1, sizeof(*g));
  if (!g) return fail(FMD_E_NOMEM, "out of host memory");
  if (hipHostMalloc((void **)&g->ring, ring_bytes, hipHostMallocDefault) != hipSuccess) {
    free(g);
    return fail(FMD_E_NOMEM, "pinned allocation of %u bytes failed", ring_bytes);
  }
  memset(g->ring, 0, ring_bytes);                     /* zero like the reference's static _input_buffer */
  g->batch = b;
  g->stream = stream;
  g->cap = ring_bytes;
  g->overflow_mode = FMD_OVERFLOW_DROP_OLDEST;
  pthread_mutex_init(&g->m, NULL);
  b->ingest[stream] = g;
  *out = g;
  return FMD_OK;
}

/* the batch is going away (fmd_batch_destroy, after it has waited for every copy out of the ring) */
static void ingest_detach(struct fmd_ingest *g) {
  pthread_mutex_lock(&g->m);
  g->batch = NULL;
  g->inflight = 0;
  g->debt = 0;
  pthread_mutex_unlock(&g->m);
}

void fmd_ingest_destroy(fmd_ingest *g) {
  if (!g) return;
  pthread_mutex_lock(&g->m);
  fmd_batch *b = g->batch;
  const int busy = g->inflight != 0 || g->debt != 0;     /* debt: in-flight bytes an overflow has moved out of `inflight` */
  pthread_mutex_unlock(&g->m);
  if (b) {
    /* a queued H2D copy may still read this pinned ring: any job of the batch that holds ring bytes, whatever
     * this ring's own counters say after an overflow - let it finish before the memory goes away */
    if (busy || b->pump[0].ring_held || b->pump[1].ring_held) {
      hipSetDevice(b->device);
      batch_quiesce(b);
    }
    if (b->ingest) b->ingest[g->stream] = NULL;
  }
  if (g->unbound) free(g->ring);
  else hipHostFree(g->ring);
  pthread_mutex_destroy(&g->m);
  free(g);
}

int fmd_ingest_set_overflow(fmd_ingest *g, int mode) {
  if (!g || (mode != FMD_OVERFLOW_DROP_OLDEST && mode != FMD_OVERFLOW_REFERENCE)) return fail(FMD_E_ARG, "bad argument");
  pthread_mutex_lock(&g->m);
  g->overflow_mode = mode;
  pthread_mutex_unlock(&g->m);
  return FMD_OK;
}

void fmd_ingest_mute(fmd_ingest *g, int n_bytes) {
  if (!g) return;
  pthread_mutex_lock(&g->m);
  g->mute = n_bytes;
  pthread_mutex_unlock(&g->m);
}

/* rtlsdr_read_async_cb_t; the role of rtlsdr_callback (src/rtl_fm_player.c:790-837): optional mute
 * fill (:805-810), copy into the ring under the lock, overflow accounting.  Never blocks on the GPU. */
void fmd_ingest_callback(unsigned char *buf, uint32_t len, void *ctx) {
  fmd_ingest *g = (fmd_ingest *)ctx;
  if (!g || !buf || len == 0) return;
  pthread_mutex_lock(&g->m);
  if (g->mute) {
    uint32_t n = (uint32_t)g->mute < len ? (uint32_t)g->mute : len;
    memset(buf, 127, n);               /* the reference fills the USB buffer itself too (:807-808) */
    g->mute = 0;
  }
  if (g->overflow_mode == FMD_OVERFLOW_REFERENCE) {
    if (len > g->cap) { buf += len - g->cap; g->dropped += len - g->cap; len = g->cap; }   /* cannot happen with USB transfers */
    if (g->wpos + len <= g->cap) {                                  /* :813-820 */
      memcpy(g->ring + g->wpos, buf, len);
      g->wpos += len;
      if (g->wpos == g->cap) g->wpos = 0;
    } else {                                                        /* :821-827: restart at zero */
      memcpy(g->ring, buf, len);
      g->wpos = len;
    }
    if ((uint64_t)g->size + len > g->cap) {                         /* :829-834: clamp the count, rpos stays */
      g->dropped += (uint64_t)g->size + len - g->cap;
      g->size = g->cap;
    } else {
      g->size += len;
    }
    pthread_mutex_unlock(&g->m);
    return;
  }
  if (len > g->cap) {            /* keep the newest cap bytes */
    g->dropped += len - g->cap;
    buf += len - g->cap;
    len = g->cap;
  }
  uint32_t first = g->cap - g->wpos;
  if (first > len) first = len;
  memcpy(g->ring + g->wpos, buf, first);
  memcpy(g->ring, buf + first, len - first);
  g->wpos = (g->wpos + len) % g->cap;
  if ((uint64_t)g->size + len > g->cap) {        /* overwrote the oldest data */
    const uint32_t over = (uint32_t)((uint64_t)g->size + len - g->cap);
    g->dropped += over;
    g->rpos = (g->rpos + over) % g->cap;
    g->size = g->cap;
    /* bytes a job was still reading have been overwritten (that job's block is damaged, as any
     * overflow damages the stream): they are released here, not again when the job ends */
    const uint32_t eaten = over < g->inflight ? over : g->inflight;
    g->inflight -= eaten;
    g->debt += eaten;
  } else {
    g->size += len;
  }
  pthread_mutex_unlock(&g->m);
}

/* The dequeue of demod_thread_fn (src/rtl_fm_player.c:863-876) for callers that drain a ring
 * themselves: when at least len bytes are buffered, copies them out and returns len, else 0. */
uint32_t fmd_ingest_pop(fmd_ingest *g, uint8_t *out, uint32_t len) {
  if (!g || !out || len == 0 || len > g->cap) return 0;
  pthread_mutex_lock(&g->m);
  /* jobs hold the bytes in front of these (they cannot be released out of order): nothing is copied then */
  if (g->inflight != 0 || g->debt != 0 || g->size < len) { pthread_mutex_unlock(&g->m); return 0; }
  const uint32_t from = g->rpos;
  uint32_t first = g->cap - from;
  if (first > len) first = len;
  memcpy(out, g->ring + from, first);
  memcpy(out + first, g->ring, len - first);
  g->rpos = (g->rpos + len) % g->cap;
  g->size -= len;
  pthread_mutex_unlock(&g->m);
  return len;
}

uint32_t fmd_ingest_buffered(const fmd_ingest *gc) {
  fmd_ingest *g = (fmd_ingest *)gc;
  if (!g) return 0;
  pthread_mutex_lock(&g->m);
  const uint32_t n = g->size - g->inflight;     /* bytes no job has taken yet */
  pthread_mutex_unlock(&g->m);
  return n;
}

uint64_t fmd_ingest_dropped(const fmd_ingest *gc) {
  fmd_ingest *g = (fmd_ingest *)gc;
  if (!g) return 0;
  pthread_mutex_lock(&g->m);
  const uint64_t n = g->dropped;
  pthread_mutex_unlock(&g->m);
  return n;
}

static int pump_slot_reserve(fmd_batch *b, struct pump_slot *p, int nb) {
  if (!p->done) {
    HIP_TRY(hipEventCreateWithFlags(&p->h2d_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p->done, hipEventDisableTiming));
  }
  if ((size_t)nb <= p->cap_blocks) return FMD_OK;
  if (p->h_pcm) hipHostFree(p->h_pcm);
  if (p->h_lens) hipHostFree(p->h_lens);
  if (p->d_iq) hipFree(p->d_iq);
  if (p->d_pcm) hipFree(p->d_pcm);
  if (p->d_lens) hipFree(p->d_lens);
  p->h_pcm = NULL; p->h_lens = NULL; p->d_iq = p->d_pcm = p->d_lens = NULL;
  p->cap_blocks = 0;
  const size_t slots = (size_t)b->n_streams * (size_t)nb;
  HIP_TRY(hipHostMalloc((void **)&p->h_pcm, slots * (size_t)b->pcm_stride * sizeof(int16_t), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void **)&p->h_lens, slots * sizeof(int32_t), hipHostMallocDefault));
  HIP_TRY(hipMalloc(&p->d_iq, slots * (size_t)b->cfg.block_len));
  HIP_TRY(hipMalloc(&p->d_pcm, slots * (size_t)b->pcm_stride * sizeof(int16_t)));
  HIP_TRY(hipMalloc(&p->d_lens, slots * sizeof(int32_t)));
  p->cap_blocks = (size_t)nb;
  return FMD_OK;
}

/* Hand a job's bytes back to the rings' writers: its H2D copies have finished. */
static void pump_release_ring(fmd_batch *b, struct pump_slot *p) {
  if (!p->ring_held) return;
  const uint32_t take = (uint32_t)p->n_blocks * (uint32_t)b->cfg.block_len;
  for (int s = 0; s < b->n_streams; s++) {
    fmd_ingest *g = b->ingest[s];
    if (!g) continue;
    pthread_mutex_lock(&g->m);
    uint32_t r = take;
    const uint32_t d = g->debt < r ? g->debt : r;    /* part an overflow has released already */
    g->debt -= d;
    r -= d;
    if (r > g->inflight) r = g->inflight;
    g->rpos = (g->rpos + r) % g->cap;
    g->size -= r;
    g->inflight -= r;
    pthread_mutex_unlock(&g->m);
  }
  p->ring_held = 0;
}

/* Release the ring space of jobs whose H2D has completed, without waiting. */
static void pump_release_completed(fmd_batch *b) {
  for (int i = 0; i < 2; i++) {
    struct pump_slot *p = &b->pump[i];
    if (p->n_blocks > 0 && p->ring_held && hipEventQuery(p->h2d_done) == hipSuccess) pump_release_ring(b, p);
  }
}

/* Start one job: whole blocks that every bound stream has buffered (at most max_blocks) are copied
 * to the device STRAIGHT FROM THE PINNED RINGS on the copy stream (one or two asynchronous copies
 * per stream), then kernel and D2H are queued on the batch stream and the call returns.  The bytes
 * stay accounted in the rings until their copy has finished (released by the next _begin / _end that
 * finds the copy done), so nothing is lost if a later step of this call fails.  Up to two jobs may be
 * in flight: the H2D of job k+1 runs beside the kernel of job k. */
int fmd_batch_pump_begin(fmd_batch *b, int max_blocks) {
  if (!b || max_blocks <= 0) return fail(FMD_E_ARG, "bad argument");
  struct pump_slot *p = &b->pump[b->pump_head];
  if (p->n_blocks > 0) return fail(FMD_E_STATE, "two jobs already in flight: call fmd_batch_pump_end first");
  HIP_TRY(hipSetDevice(b->device));
  pump_release_completed(b);
  const uint32_t bl = (uint32_t)b->cfg.block_len;
  int nb = max_blocks;
  for (int s = 0; s < b->n_streams; s++) {
    fmd_ingest *g = b->ingest[s];
    if (!g) return fail(FMD_E_STATE, "stream %d has no ingest ring", s);
    pthread_mutex_lock(&g->m);
    int have = (int)((g->size - g->inflight) / bl);
    pthread_mutex_unlock(&g->m);
    if (have < nb) nb = have;
  }
  if (nb == 0) return 0;
  if (!b->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
  int rc = pump_slot_reserve(b, p, nb);              /* every buffer the job needs, before a byte is taken */
  if (rc) return rc;
  const uint32_t take = (uint32_t)nb * bl;
  int taken = 0;
  hipError_t e = hipSuccess;
  for (int s = 0; s < b->n_streams && e == hipSuccess; s++) {
    fmd_ingest *g = b->ingest[s];
    uint8_t *dst = (uint8_t *)p->d_iq + (size_t)s * take;
    pthread_mutex_lock(&g->m);
    const uint32_t from = (g->rpos + g->inflight) % g->cap;
    g->inflight += take;
    pthread_mutex_unlock(&g->m);
    taken = s + 1;
    uint32_t first = g->cap - from;
    if (first > take) first = take;
    e = hipMemcpyAsync(dst, g->ring + from, first, hipMemcpyHostToDevice, b->copy_stream);
    if (e == hipSuccess && take > first)
      e = hipMemcpyAsync(dst + first, g->ring, take - first, hipMemcpyHostToDevice, b->copy_stream);
  }
  const size_t slots = (size_t)b->n_streams * (size_t)nb;
  if (e == hipSuccess) e = hipEventRecord(p->h2d_done, b->copy_stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(b->stream, p->h2d_done, 0);
  int launched = 0;
  if (e == hipSuccess) {
    rc = fmd_batch_run_device(b, p->d_iq, nb, p->d_pcm, p->d_lens, NULL);
    if (rc == FMD_OK) {
      launched = 1;                                    /* the streams' state has advanced by nb blocks from here on */
      e = hipMemcpyAsync(p->h_pcm, p->d_pcm, slots * (size_t)b->pcm_stride * sizeof(int16_t), hipMemcpyDeviceToHost,
                         b->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(p->h_lens, p->d_lens, slots * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream);
      if (e == hipSuccess) e = hipEventRecord(p->done, b->stream);
    }
  }
  if (!launched) {
    /* nothing has been demodulated: give the bytes back.  Wait for the copies already queued, then un-take them
     * (they are still in the rings, so the next call sees them again); a part an overflow has meanwhile released
     * (moved from inflight to debt) is not taken back a second time */
    hipStreamSynchronize(b->copy_stream);
    /* debt = bytes an overflow has eaten from the OLDEST end of the in-flight region: they belong to the other job first
     * (if one still holds ring space), and only what exceeds that job's take was eaten from this one.  That part is already
     * released (rpos and size moved on when the overflow happened); the rest of this job's take goes back to "buffered",
     * and the other job's share of the debt stays for its own release. */
    const struct pump_slot *other = &b->pump[b->pump_head ^ 1];
    const uint32_t old_take = (other->n_blocks > 0 && other->ring_held) ? (uint32_t)other->n_blocks * bl : 0;
    for (int s = 0; s < taken; s++) {
      fmd_ingest *g = b->ingest[s];
      pthread_mutex_lock(&g->m);
      const uint32_t mine = g->debt > old_take ? g->debt - old_take : 0;      /* eaten from this job's bytes */
      const uint32_t eaten = mine < take ? mine : take;
      g->debt -= eaten;
      const uint32_t r = take - eaten;
      g->inflight -= r < g->inflight ? r : g->inflight;
      pthread_mutex_unlock(&g->m);
    }
    if (e != hipSuccess) return fail(FMD_E_HIP, "pump: %s (%d)", hipGetErrorString(e), (int)e);
    return rc;
  }
  /* the kernel is queued: the job exists whatever happened to its D2H (un-taking the bytes now would demodulate
   * the same blocks twice); a failed D2H is reported by fmd_batch_pump_end, which still releases the ring */
  p->n_blocks = nb;
  p->ring_held = 1;
  p->failed = (e != hipSuccess) ? (int)e : 0;
  b->pump_head ^= 1;
  return nb;
}

/* Finish the oldest job begun: waits for it and copies its PCM and lengths out (layout as
 * fmd_batch_run_host for that job's block count).  Returns the block count, 0 if none. */
int fmd_batch_pump_end(fmd_batch *b, int16_t *pcm, int32_t *lens) {
  if (!b || !pcm || !lens) return fail(FMD_E_ARG, "bad argument");
  struct pump_slot *p = &b->pump[b->pump_tail];
  if (p->n_blocks <= 0) return 0;
  HIP_TRY(hipSetDevice(b->device));
  if (p->failed) {
    /* the job ran (state advanced) but its PCM never left the device: wait for the kernel, free the slot and the
     * ring space, report */
    const int err = p->failed;
    hipStreamSynchronize(b->stream);
    pump_release_ring(b, p);
    p->n_blocks = 0;
    p->failed = 0;
    b->pump_tail ^= 1;
    return fail(FMD_E_HIP, "pump: the job's device-to-host copy could not be queued: %s (%d); its %s", hipGetErrorString((hipError_t)err),
                err, "blocks were demodulated and are lost");
  }
  HIP_TRY(hipEventSynchronize(p->done));
  pump_release_ring(b, p);                           /* done implies its H2D is done */
  pump_release_completed(b);
  const size_t slots = (size_t)b->n_streams * (size_t)p->n_blocks;
  memcpy(pcm, p->h_pcm, slots * (size_t)b->pcm_stride * sizeof(int16_t));
  memcpy(lens, p->h_lens, slots * sizeof(int32_t));
  const int nb = p->n_blocks;
  p->n_blocks = 0;
  b->pump_tail ^= 1;
  return nb;
}

int fmd_batch_pump(fmd_batch *b, int max_blocks, int16_t *pcm, int32_t *lens) {
  if (!b || !pcm || !lens || max_blocks <= 0) return fail(FMD_E_ARG, "bad argument");
  if (b->pump[b->pump_tail].n_blocks > 0)
    return fail(FMD_E_STATE, "jobs in flight: finish them with fmd_batch_pump_end");
  const int nb = fmd_batch_pump_begin(b, max_blocks);
  if (nb <= 0) return nb;
  const int got = fmd_batch_pump_end(b, pcm, lens);
  return got < 0 ? got : nb;
}
