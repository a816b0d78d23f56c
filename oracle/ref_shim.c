/*
 * ref_shim.c - a thin handle API over the REFERENCE's own hot path.
 *
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/).  This file is never
 * compiled on its own: oracle/build_ref.py appends it to a translation unit
 * that it assembles IN MEMORY from line ranges of the reference sources where
 * they lie (/root/reference/src/rtl_fm_player.c:195-788 and the type / table /
 * constant lines of /root/reference/include/rtl_fm_player.h), and pipes the
 * whole to gcc.  Nothing of the reference is copied into this repository and no
 * stand-in header is written: the sliced lines need only <math.h>, <string.h>,
 * <stdint.h>, <stdlib.h>, <stdio.h> and <pthread.h>.  The result,
 * oracle/_ref/libref.so, is "the reference compiled here": it pins
 * oracle/fm_oracle.c (tests/test_ref_pin.py), generates tests/golden/*.npz
 * (tests/golden/make_ref_fixtures.py) and is the CPU baseline bench.py times
 * (cpu_baseline.kind == "reference").
 *
 * Everything below is this repository's code.  It drives the reference the way
 * its own main() / demod_thread_fn do (file:line cited per step) on a heap
 * struct demod_state, one per handle, so several configurations can be open at
 * once; the static tables (u8_f32_table, lp_filter_f32) are shared and
 * idempotent, exactly as in the program.
 */

#define REF_API __attribute__((visibility("default")))

typedef struct ref_handle {
  struct demod_state d;
  int offset_tuning;
} ref_handle;

/* Open a stream.  Field values follow demod_init (src/rtl_fm_player.c:1156-1195) with the
 * command-line overrides of main (:1412-1419 -s / -r, :1464-1488 -X / -Y, :1454-1457 -E offset,
 * :1575-1578 deemph_lambda); then the three init calls of :1601-1603. */
REF_API void *ref_open(int rate_in, int rate_out, int rate_out2, int mode, int size, int deemph_on,
                       float deemph_lambda, float volume, int offset_tuning) {
  ref_handle *h = calloc(1, sizeof(*h));       /* `demod` is a zero-initialised global in the program */
  if (!h) return NULL;
  struct demod_state *s = &h->d;
  s->rate_in = rate_in;
  s->rate_out = rate_out;
  s->rate_out2 = rate_out2;
  s->squelch_level = 0;
  s->conseq_squelch = 10;
  s->squelch_hits = 11;
  s->post_downsample = 1;
  s->custom_atan = 1;
  s->deemph = deemph_on ? DEEMPHASIS_FM_EU : DEEMPHASIS_NONE;   /* full_demod only tests it for non-zero (:784) */
  s->deemph_lambda = deemph_lambda;
  s->offset_tuning = offset_tuning;
  s->volume = volume;
  s->lpr.mode = mode;
  s->lpr.size = size;
  pthread_rwlock_init(&s->rw, NULL);
  h->offset_tuning = offset_tuning;
  init_u8_f32_table();
  init_lp_f32();
  init_lp_real_f32(s);
  return h;
}

REF_API void ref_close(void *hv) {
  ref_handle *h = hv;
  if (!h) return;
  deinit_lp_real_f32(&h->d);
  pthread_rwlock_destroy(&h->d.rw);
  free(h);
}

static void ref_feed(ref_handle *h, const uint8_t *iq, uint32_t len) {
  struct demod_state *d = &h->d;
  memcpy(d->buf, iq, len);                 /* demod_thread_fn :870-876 */
  d->buf_len = len;
  if (h->offset_tuning) u8_f32(d);         /* :879-886 */
  else rotate_90_u8_f32(d);
}

/* One block exactly as demod_thread_fn runs it (:879-889): conversion, then full_demod().
 * Returns result_len; pcm receives d->result[0..result_len). */
REF_API int ref_block(void *hv, const uint8_t *iq, uint32_t len, int16_t *pcm) {
  ref_handle *h = hv;
  if (!h || len > MAXIMUM_BUF_LENGTH || len < 64 || (len & 15)) return -1;
  ref_feed(h, iq, len);
  full_demod(&h->d);
  memcpy(pcm, h->d.result, (size_t)h->d.result_len * sizeof(int16_t));
  return h->d.result_len;
}

/* The same block with the stages of full_demod (:758-788) called one by one in its order,
 * so the intermediates the reference leaves in d->lowpassed / d->result can be copied out
 * between them: y = decimated IQ after lp_f32, v = discriminator output after fm_demod_f32,
 * mpx = resampler output after lp_real_f32 (before de-emphasis).  Any pointer may be NULL.
 * tests/test_ref_pin.py checks that this staged walk and full_demod() itself give the same PCM. */
REF_API int ref_block_staged(void *hv, const uint8_t *iq, uint32_t len, int16_t *pcm, float *y, float *v,
                             float *mpx) {
  ref_handle *h = hv;
  if (!h || len > MAXIMUM_BUF_LENGTH || len < 64 || (len & 15)) return -1;
  struct demod_state *d = &h->d;
  ref_feed(h, iq, len);
  lp_f32(d);                                                           /* :764 */
  if (y) memcpy(y, d->lowpassed, (size_t)d->lp_len * sizeof(float));
  fm_demod_f32(d);                                                     /* :771 */
  if (v) memcpy(v, d->result, (size_t)d->result_len * sizeof(float));
  if (d->rate_out2 > 0) lp_real_f32(d);                                /* :781-782 */
  if (mpx) memcpy(mpx, d->result, (size_t)d->result_len * sizeof(float));
  if (d->deemph) deemph_filter_f32(d);                                 /* :784-785 */
  convert_f32_s16(d);                                                  /* :787 */
  memcpy(pcm, d->result, (size_t)d->result_len * sizeof(int16_t));
  return d->result_len;
}

/* n_blocks consecutive blocks of len bytes each; PCM back to back, lens[b] = result_len.
 * Returns the number of int16 written.  (The loop bench.py times as the CPU baseline.) */
REF_API long ref_run(void *hv, const uint8_t *iq, uint32_t len, int n_blocks, int16_t *pcm, int32_t *lens) {
  long total = 0;
  for (int b = 0; b < n_blocks; b++) {
    const int n = ref_block(hv, iq + (size_t)b * len, len, pcm + total);
    if (n < 0) return -1;
    if (lens) lens[b] = n;
    total += n;
  }
  return total;
}

/* Carried state, ring buffers in the reference's own (ring) order plus lpr.pos. */
typedef struct ref_state {
  float tb[48];
  float pre_r, pre_j, pp, deemph_l, deemph_r;
  int32_t acc, pos, size;
  float br[256], bm[256], bs[256];
} ref_state;

REF_API void ref_get_state(void *hv, ref_state *o) {
  ref_handle *h = hv;
  const struct demod_state *d = &h->d;
  memset(o, 0, sizeof(*o));
  memcpy(o->tb, d->lowpass_tb, sizeof(o->tb));
  o->pre_r = d->pre_r_f32;
  o->pre_j = d->pre_j_f32;
  o->pp = d->lpr.pp;
  o->deemph_l = d->deemph_l_f32;
  o->deemph_r = d->deemph_r_f32;
  o->acc = d->prev_lpr_index;
  o->pos = d->lpr.pos;
  o->size = d->lpr.size;
  const int n = d->lpr.size < 256 ? d->lpr.size : 256;
  memcpy(o->br, d->lpr.br, (size_t)n * 4);
  memcpy(o->bm, d->lpr.bm, (size_t)n * 4);
  memcpy(o->bs, d->lpr.bs, (size_t)n * 4);
}

/* Filter tables as the reference's init functions left them. */
REF_API void ref_get_taps(void *hv, float *fb, float *fm, float *fp, float *fs, float *swf, float *cwf) {
  ref_handle *h = hv;
  const struct demod_state *d = &h->d;
  memcpy(fb, lp_filter_f32, 16 * sizeof(float));
  const int half = d->lpr.size >> 1;
  memcpy(fm, d->lpr.fm, (size_t)half * 4);
  memcpy(fp, d->lpr.fp, (size_t)half * 4);
  memcpy(fs, d->lpr.fs, (size_t)half * 4);
  *swf = d->lpr.swf;
  *cwf = d->lpr.cwf;
}

REF_API void ref_get_u8_table(float *t0, float *t1) {
  init_u8_f32_table();
  memcpy(t0, u8_f32_table[0], 256 * sizeof(float));
  memcpy(t1, u8_f32_table[1], 256 * sizeof(float));
}

REF_API size_t ref_sizeof_demod_state(void) { return sizeof(struct demod_state); }
REF_API size_t ref_offsetof_demod_state(int which) {
  switch (which) {
    case 0: return offsetof(struct demod_state, buf);
    case 1: return offsetof(struct demod_state, buf_len);
    case 2: return offsetof(struct demod_state, lowpassed);
    case 3: return offsetof(struct demod_state, lp_len);
    case 4: return offsetof(struct demod_state, lowpass_tb);
    case 5: return offsetof(struct demod_state, result);
    case 6: return offsetof(struct demod_state, result_len);
    case 7: return offsetof(struct demod_state, rate_in);
    case 8: return offsetof(struct demod_state, pre_r_f32);
    case 9: return offsetof(struct demod_state, deemph);
    case 10: return offsetof(struct demod_state, deemph_lambda);
    case 11: return offsetof(struct demod_state, volume);
    case 12: return offsetof(struct demod_state, prev_lpr_index);
    case 13: return offsetof(struct demod_state, lpr);
    case 14: return offsetof(struct demod_state, rw);
    case 15: return offsetof(struct demod_state, output_target);
    default: return (size_t)-1;
  }
}
