/*
 * fm_oracle.c - CPU restatement of the rtl_fm_player IQ->PCM hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see fm_oracle.h).  Written from the verified
 * closed-form description of each stage; every function
 * cites the reference lines whose behaviour it follows.  All arithmetic is
 * float32 with separate (unfused) multiplies and adds, sums evaluated
 * left-to-right in k, exactly as the reference's Release build does; compile
 * with -ffp-contract=off.
 */
#include "fm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FMO_PI 3.14159265f    /* PI_F   include/rtl_fm_player.h:40 */
#define FMO_2PI 6.28318531f   /* PI2_F  include/rtl_fm_player.h:39 */
#define FMO_PI_2 1.5707963f   /* PI_2_F include/rtl_fm_player.h:41 */
#define FMO_PI_4 0.78539816f  /* PI_4_F include/rtl_fm_player.h:42 */

#define FMO_MAX_SIZE 256

struct fmo_stream {
  fmo_config cfg;
  int half;               /* lpr.rsize */
  float lut_pos[256];     /* u8_f32_table[0] */
  float lut_neg[256];     /* u8_f32_table[1] */
  float fb[16];           /* lp_filter_f32 */
  float fm[FMO_MAX_SIZE / 2], fp[FMO_MAX_SIZE / 2], fs[FMO_MAX_SIZE / 2];
  float swf, cwf;
  fmo_state st;
  /* work buffers, grown on demand */
  uint32_t cap_len;
  float *c;               /* [48 + len]   history + rotated samples            */
  float *y;               /* [len / 8]    decimated IQ                          */
  float *res;             /* [len / 16]   discriminator output / in-place mpx   */
  float *hv, *hbm, *hbs;  /* [size + len/16] linear histories for the MPX FIRs  */
};

/* ---------------------------------------------------------------- tables */

/* src/rtl_fm_player.c:195-204 */
static void build_u8_tables(fmo_stream *s) {
  for (int i = 0; i < 256; i++) {
    s->lut_pos[i] = ((float)i - 127.5f) / 128.0f;
    s->lut_neg[i] = ((float)i - 127.5f) / -128.0f;
  }
}

/* src/rtl_fm_player.c:241-251 */
static void build_iq_taps(fmo_stream *s) {
  for (int i = 0; i < 16; i++) {
    float j = (float)i - 15.5f;
    s->fb[i] = (sinf(0.125f * FMO_PI * j) / (FMO_PI * j)) *
               (0.54f - 0.46f * cosf(FMO_PI * (float)i / 15.5f));
  }
}

/* src/rtl_fm_player.c:413-453 (filter design only; the rings live in st) */
static void build_mpx_taps(fmo_stream *s) {
  const int size = s->cfg.size;
  const float rate = (float)s->cfg.rate_in;
  float wf = FMO_2PI * 19000.0f / rate;
  s->swf = sinf(wf);
  s->cwf = cosf(wf);
  float fmh = 16000.0f / rate;
  float fpl = 18000.0f / rate;
  float fph = 20000.0f / rate;
  float fsl = 21000.0f / rate;
  float fsh = 55000.0f / rate;
  s->half = size >> 1;
  for (int i = 0; i < s->half; i++) {
    float fi = (float)i - (float)(size - 1) / 2.0f;
    float fh = 0.54f - 0.46f * cosf(FMO_2PI * (float)i / (float)(size - 1));
    float fv;
    fv = (fi == 0) ? 2.0f * fmh : sinf(FMO_2PI * fmh * fi) / (FMO_PI * fi);
    s->fm[i] = fv * fh;
    fv = (fi == 0) ? 2.0f * (fph - fpl)
                   : (sinf(FMO_2PI * fph * fi) - sinf(FMO_2PI * fpl * fi)) / (FMO_PI * fi);
    s->fp[i] = fv * fh;
    fv = (fi == 0) ? 2.0f * (fsh - fsl)
                   : (sinf(FMO_2PI * fsh * fi) - sinf(FMO_2PI * fsl * fi)) / (FMO_PI * fi);
    s->fs[i] = fv * fh;
  }
}

float fmo_deemph_lambda(int output_rate, double tau) {
  /* src/rtl_fm_player.c:1575-1578 */
  return (float)exp(-1.0 / ((double)output_rate * tau));
}

/* ---------------------------------------------------------------- open/close */

fmo_stream *fmo_open(const fmo_config *cfg) {
  if (!cfg || cfg->size < 2 || cfg->size > FMO_MAX_SIZE || (cfg->size & 1)) return NULL;
  if (cfg->mode < 0 || cfg->mode > 2 || cfg->rate_in <= 0) return NULL;
  fmo_stream *s = (fmo_stream *)calloc(1, sizeof(*s));
  if (!s) return NULL;
  s->cfg = *cfg;
  build_u8_tables(s);
  build_iq_taps(s);
  build_mpx_taps(s);
  s->st.size = cfg->size;
  return s;
}

void fmo_close(fmo_stream *s) {
  if (!s) return;
  free(s->c);
  free(s->y);
  free(s->res);
  free(s->hv);
  free(s->hbm);
  free(s->hbs);
  free(s);
}

static int ensure_work(fmo_stream *s, uint32_t len) {
  if (len <= s->cap_len) return 0;
  free(s->c); free(s->y); free(s->res); free(s->hv); free(s->hbm); free(s->hbs);
  size_t m = len / 16;
  s->c = (float *)malloc(sizeof(float) * (48 + (size_t)len));
  s->y = (float *)malloc(sizeof(float) * (len / 8));
  s->res = (float *)malloc(sizeof(float) * (m + 4));
  s->hv = (float *)malloc(sizeof(float) * (FMO_MAX_SIZE + m));
  s->hbm = (float *)malloc(sizeof(float) * (FMO_MAX_SIZE + m));
  s->hbs = (float *)malloc(sizeof(float) * (FMO_MAX_SIZE + m));
  if (!s->c || !s->y || !s->res || !s->hv || !s->hbm || !s->hbs) {
    s->cap_len = 0;
    return -1;
  }
  s->cap_len = len;
  return 0;
}

/* ---------------------------------------------------------------- stages */

/* Stage a2/a2': u8 -> f32 with the j^n rotation (src/rtl_fm_player.c:206-226)
 * or without it (:228-239).  Sample n of the block (blocks are multiples of
 * four samples) is multiplied by j^(n mod 4):
 *   (I,Q), (-Q,I), (-I,-Q), (Q,-I). */
static void stage_convert(const fmo_stream *s, const uint8_t *iq, uint32_t len, float *c) {
  const float *P = s->lut_pos, *N = s->lut_neg;
  if (s->cfg.offset_tuning) {
    for (uint32_t i = 0; i < len; i++) c[i] = P[iq[i]];
    return;
  }
  for (uint32_t i = 0; i < len; i += 8) {
    const uint8_t *b = iq + i;
    float *o = c + i;
    o[0] = P[b[0]]; o[1] = P[b[1]];   /* n%4==0:  I,  Q */
    o[2] = N[b[3]]; o[3] = P[b[2]];   /* n%4==1: -Q,  I */
    o[4] = N[b[4]]; o[5] = N[b[5]];   /* n%4==2: -I, -Q */
    o[6] = P[b[7]]; o[7] = N[b[6]];   /* n%4==3:  Q, -I */
  }
}

/* Stage a4: 32-tap symmetric FIR, decimate by 8 (src/rtl_fm_player.c:253-411).
 * With c = [24 complex of history | block], output m is
 *   sum_{k=0..15} (c[8m-24+k] + c[8m+7-k]) * fb[k]
 * per component, products summed left to right.  `c` here already has the 48
 * history floats in front, so output m reads c[16m .. 16m+63]. */
static void stage_decimate(const fmo_stream *s, const float *c, uint32_t n_out, float *y) {
  const float *fb = s->fb;
  for (uint32_t m = 0; m < n_out; m++) {
    const float *w = c + 16 * (size_t)m;
    float ai = (w[0] + w[62]) * fb[0];
    float aq = (w[1] + w[63]) * fb[0];
    for (int k = 1; k < 16; k++) {
      ai += (w[2 * k] + w[62 - 2 * k]) * fb[k];
      aq += (w[2 * k + 1] + w[63 - 2 * k]) * fb[k];
    }
    y[2 * m] = ai;
    y[2 * m + 1] = aq;
  }
}

/* Polynomial atan2 (src/rtl_fm_player.c:606-667), restated through the
 * magnitude ratio a = min(|x|,|y|)/max(|x|,|y|) and
 *   r0 = a * (pi/4 - (a - 1) * (0.2447 + 0.0663 a)).
 * IEEE add/mul/div are sign-symmetric, so each of the reference's eight octant
 * expressions equals one of  +-r0 + {0, +-pi/2, +-pi}  bit for bit. */
static inline float poly_atan2(float y, float x) {
  if (x == 0.f) {
    if (y < 0.f) return -FMO_PI_2;
    if (y > 0.f) return FMO_PI_2;
    return 0.f;
  }
  if (y == 0.f) return (x < 0.f) ? FMO_PI : 0.f;
  float ax = fabsf(x), ay = fabsf(y);
  int x_major;  /* reference picks z = y/x in this case, else z = x/y */
  if (x < 0.f) x_major = (y < 0.f) ? (x <= y) : (-x >= y);
  else         x_major = (y < 0.f) ? (x >= -y) : (x >= y);
  float a = x_major ? ay / ax : ax / ay;
  float r0 = a * (FMO_PI_4 - (a - 1.f) * (0.2447f + 0.0663f * a));
  if (x < 0.f) {
    if (y < 0.f) return x_major ? r0 - FMO_PI : -r0 - FMO_PI_2;   /* third quadrant  */
    return x_major ? -r0 + FMO_PI : FMO_PI_2 + r0;                /* second quadrant */
  }
  if (y < 0.f) return x_major ? -r0 : r0 - FMO_PI_2;              /* fourth quadrant */
  return x_major ? r0 : FMO_PI_2 - r0;                            /* first quadrant  */
}

/* Stage a6: quadrature discriminator (src/rtl_fm_player.c:669-685). */
static void stage_discriminate(fmo_stream *s, const float *y, uint32_t n, float *v) {
  float pr = s->st.pre_r, pj = s->st.pre_j;
  for (uint32_t m = 0; m < n; m++) {
    float re = y[2 * m], im = y[2 * m + 1];
    v[m] = poly_atan2(pr * im - pj * re, re * pr + im * pj);
    pr = re;
    pj = im;
  }
  s->st.pre_r = pr;
  s->st.pre_j = pj;
}

/* src/rtl_fm_player.c:472-481: sin(2 atan(y/x)) = 2z / (1 + z^2), z = y/x */
static inline float carrier38(float x, float y) {
  if (x == 0.f) return 0.f;
  float z = y / x;
  return (z + z) / (1.f + (z * z));
}

static inline int emit_step(int *acc, int slow, int fast) {
  /* resampler accumulator, src/rtl_fm_player.c:493-496 / :507-509 / :570-572 */
  if ((*acc += slow) >= fast) {
    *acc -= fast;
    return 1;
  }
  return 0;
}

/* Stage a9/a10/a11: rational resampler + MPX decode (src/rtl_fm_player.c:483-604).
 * `res` is used in place exactly like the reference uses d->result: sample i is
 * read at step i, outputs are written at o (and o+1), so an output written
 * ahead of the read index is seen by later reads.
 * The FIR windows are kept as linear histories [size old values | new values]
 * instead of rings; window of step n = h[n+1 .. n+size], pairs
 * (oldest + k, newest - k), k ascending. */
static int stage_resample(fmo_stream *s, float *res, int n) {
  const int fast = s->cfg.rate_out, slow = s->cfg.rate_out2;
  const int size = s->cfg.size, half = s->half;
  fmo_state *st = &s->st;
  int acc = st->acc, o = 0;

  if (s->cfg.mode == 0) {                       /* :490-499 */
    for (int i = 0; i < n; i++)
      if (emit_step(&acc, slow, fast)) res[o++] = res[i];
    st->acc = acc;
    return o;
  }

  float *hv = s->hv;
  memcpy(hv, st->br, sizeof(float) * size);

  if (s->cfg.mode == 1) {                       /* :500-532 */
    const float *fm = s->fm;
    for (int i = 0; i < n; i++) {
      hv[size + i] = res[i];
      if (emit_step(&acc, slow, fast)) {
        const float *w = hv + i + 1;            /* w[0] oldest, w[size-1] newest */
        float vm = 0;
        for (int k = 0; k < half; k++) vm += (w[k] + w[size - 1 - k]) * fm[k];
        res[o++] = vm;
      }
    }
    memcpy(st->br, hv + n, sizeof(float) * size);
    st->pos = (st->pos + n) % size;
    st->acc = acc;
    return o;
  }

  /* mode 2, stereo: :533-600 */
  const float *fm = s->fm, *fp = s->fp, *fs = s->fs;
  float *hbm = s->hbm, *hbs = s->hbs;
  memcpy(hbm, st->bm, sizeof(float) * size);
  memcpy(hbs, st->bs, sizeof(float) * size);
  float pp = st->pp;
  for (int i = 0; i < n; i++) {
    hv[size + i] = res[i];
    const float *w = hv + i + 1;
    float vm = 0, vp = 0, vs = 0;
    for (int k = 0; k < half; k++) {
      float p = w[k] + w[size - 1 - k];
      vm += p * fm[k];
      vp += p * fp[k];
      vs += p * fs[k];
    }
    hbm[size + i] = vm;
    hbs[size + i] = vs * carrier38(vp * s->swf, vp * s->cwf - pp);
    pp = vp;
    if (emit_step(&acc, slow, fast)) {
      const float *wm = hbm + i + 1, *ws = hbs + i + 1;
      float om = 0, os = 0;
      for (int k = 0; k < half; k++) {
        om += (wm[k] + wm[size - 1 - k]) * fm[k];
        os += (ws[k] + ws[size - 1 - k]) * fm[k];
      }
      res[o] = om + os;
      res[o + 1] = om - os;
      o += 2;
    }
  }
  memcpy(st->br, hv + n, sizeof(float) * size);
  memcpy(st->bm, hbm + n, sizeof(float) * size);
  memcpy(st->bs, hbs + n, sizeof(float) * size);
  st->pp = pp;
  st->pos = (st->pos + n) % size;
  st->acc = acc;
  return o;
}

/* Stage a12: one-pole de-emphasis y += lambda * (y_prev - y)
 * (src/rtl_fm_player.c:687-709): sub, mul, add. */
static void stage_deemph(fmo_stream *s, float *x, int n) {
  const float lam = s->cfg.deemph_lambda;
  float l = s->st.deemph_l, r = s->st.deemph_r;
  if (s->cfg.mode == 2) {
    for (int i = 0; i < n; i += 2) {
      l = (x[i] += lam * (l - x[i]));
      r = (x[i + 1] += lam * (r - x[i + 1]));
    }
  } else {
    for (int i = 0; i < n; i++) l = (x[i] += lam * (l - x[i]));
  }
  s->st.deemph_l = l;
  s->st.deemph_r = r;
}

/* Stage a13: scale, clip, round-half-even (src/rtl_fm_player.c:711-735). */
static void stage_to_s16(const fmo_stream *s, const float *x, int n, int16_t *pcm) {
  const float coef = s->cfg.volume * 32768.0f;
  for (int i = 0; i < n; i++) {
    float t = x[i] * coef;
    if (t > 32767.0f) pcm[i] = 32767;
    else if (t < -32768.0f) pcm[i] = -32768;
    else pcm[i] = (int16_t)lrintf(t);
  }
}

/* ---------------------------------------------------------------- block */

/* demod_thread_fn's per-block sequence: rotate/convert, then full_demod
 * (src/rtl_fm_player.c:879-889, :758-788). */
int fmo_block_trace(fmo_stream *s, const uint8_t *iq, uint32_t len, int16_t *pcm,
                    const fmo_trace *tr) {
  if (!s || !iq || !pcm || len < 64 || (len & 15)) return -1;
  if (ensure_work(s, len)) return -2;
  const uint32_t n_y = len / 16;

  memcpy(s->c, s->st.tb, sizeof(float) * 48);
  stage_convert(s, iq, len, s->c + 48);
  memcpy(s->st.tb, s->c + len, sizeof(float) * 48);     /* :366 */
  stage_decimate(s, s->c, n_y, s->y);
  if (tr && tr->y) memcpy(tr->y, s->y, sizeof(float) * 2 * n_y);

  stage_discriminate(s, s->y, n_y, s->res);
  if (tr && tr->v) memcpy(tr->v, s->res, sizeof(float) * n_y);

  int n = (int)n_y;
  if (s->cfg.rate_out2 > 0) n = stage_resample(s, s->res, n);   /* :781-782 */
  if (tr && tr->mpx) memcpy(tr->mpx, s->res, sizeof(float) * n);
  if (s->cfg.deemph) stage_deemph(s, s->res, n);                /* :784-785 */
  stage_to_s16(s, s->res, n, pcm);                              /* :787 */
  return n;
}

int fmo_block(fmo_stream *s, const uint8_t *iq, uint32_t len, int16_t *pcm) {
  return fmo_block_trace(s, iq, len, pcm, NULL);
}

long fmo_run(fmo_stream *s, const uint8_t *iq, uint32_t len, int n_blocks, int16_t *pcm,
             int32_t *lens) {
  long total = 0;
  for (int b = 0; b < n_blocks; b++) {
    int n = fmo_block(s, iq + (size_t)b * len, len, pcm + total);
    if (n < 0) return n;
    if (lens) lens[b] = n;
    total += n;
  }
  return total;
}

void fmo_get_state(const fmo_stream *s, fmo_state *out) { *out = s->st; }
void fmo_set_state(fmo_stream *s, const fmo_state *in) {
  s->st = *in;
  s->st.size = s->cfg.size;
}

void fmo_get_taps(const fmo_stream *s, float *fb, float *fm, float *fp, float *fs, float *swf,
                  float *cwf) {
  if (fb) memcpy(fb, s->fb, sizeof(s->fb));
  if (fm) memcpy(fm, s->fm, sizeof(float) * s->half);
  if (fp) memcpy(fp, s->fp, sizeof(float) * s->half);
  if (fs) memcpy(fs, s->fs, sizeof(float) * s->half);
  if (swf) *swf = s->swf;
  if (cwf) *cwf = s->cwf;
}

/* ---------------------------------------------------------------- synth + hash */

void fmo_lcg_fill(uint32_t *state, uint8_t *buf, size_t n) {
  uint32_t s = *state;
  for (size_t i = 0; i < n; i++) {
    s = s * 1664525u + 1013904223u;
    buf[i] = (uint8_t)(s >> 24);
  }
  *state = s;
}

uint64_t fmo_hash16(uint64_t h, const int16_t *x, size_t n) {
  for (size_t i = 0; i < n; i++) {
    h ^= (uint64_t)(uint16_t)x[i];
    h *= 1099511628211ULL;
  }
  return h;
}

/* Fixed-point sine, Q15 out, phase = full circle over 2^32.  Parabola with
 * one refinement step (|error| < 0.1 %), integer only. */
static int32_t isin_q15(uint32_t phase) {
  int neg = (phase >> 31) & 1;
  uint32_t t = (phase & 0x7fffffffu) >> 16;          /* 0..32767 : half circle */
  int32_t y = (int32_t)((4ull * t * (32768u - t)) >> 15);        /* 0..32768 */
  int32_t y2 = (int32_t)(((uint32_t)y * (uint32_t)y) >> 15);
  y = y + (int32_t)((7373 * (y2 - y)) / 32768);
  if (y > 32767) y = 32767;
  return neg ? -y : y;
}

void fmo_dds_init(fmo_dds *d, int fs, int f_left, int f_right, int amp, int stereo,
                  uint32_t seed) {
  memset(d, 0, sizeof(*d));
  d->step_l = (uint32_t)(((uint64_t)f_left << 32) / (uint64_t)fs);
  d->step_r = (uint32_t)(((uint64_t)f_right << 32) / (uint64_t)fs);
  d->step_pilot = (uint32_t)(((uint64_t)19000 << 32) / (uint64_t)fs);
  /* +-75 kHz for wide FM; scaled down when fs cannot carry it */
  int dev = (fs >= 1000000) ? 75000 : fs / 40;
  d->dev_q = (int32_t)(((uint64_t)dev << 32) / (uint64_t)fs);
  d->amp = amp;
  d->stereo = stereo;
  d->noise = seed;
}

void fmo_dds_fill(fmo_dds *d, uint8_t *buf, size_t n_bytes) {
  for (size_t i = 0; i + 1 < n_bytes; i += 2) {
    int32_t l = isin_q15(d->ph_l), r = isin_q15(d->ph_r);
    int32_t sum = (l + r) / 2, diff = (l - r) / 2, mpx;
    if (d->stereo) {
      int32_t sub = isin_q15(d->ph_pilot * 2u);
      mpx = (29491 * (sum + (diff * sub) / 32768)) / 32768 + (3277 * isin_q15(d->ph_pilot)) / 32768;
    } else {
      mpx = (29491 * sum) / 32768;
    }
    int64_t dphi = ((int64_t)d->dev_q * mpx) / 32768;
    d->ph_carrier += (uint32_t)(0xC0000000u + (uint32_t)(int32_t)dphi);   /* -fs/4 + deviation */
    d->ph_l += d->step_l;
    d->ph_r += d->step_r;
    d->ph_pilot += d->step_pilot;
    int32_t cs = isin_q15(d->ph_carrier + 0x40000000u), sn = isin_q15(d->ph_carrier);
    d->noise = d->noise * 1664525u + 1013904223u;
    uint32_t d1 = (d->noise >> 8) & 0x7fff, d2 = (d->noise >> 17) & 0x7fff;
    int32_t ui = (int32_t)((uint32_t)(d->amp * cs + (128 << 15)) + d1) >> 15;
    int32_t uq = (int32_t)((uint32_t)(d->amp * sn + (128 << 15)) + d2) >> 15;
    buf[i] = (uint8_t)(ui < 0 ? 0 : ui > 255 ? 255 : ui);
    buf[i + 1] = (uint8_t)(uq < 0 ? 0 : uq > 255 ? 255 : uq);
  }
}
