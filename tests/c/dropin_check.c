/*
 * dropin_check.c - a C caller of the reference-shaped surface, through the C ABI only.
 *
 * Includes include/fmdemod_mi355x.h WITH the reference type restatements (no
 * FMD_NO_REFERENCE_TYPES) and drives the library exactly as rtl_fm_player's main() and
 * demod_thread_fn do: demod_init (src/rtl_fm_player.c:1382), option overrides as -X -s 300000
 * would set them (:1412-1419, :1464-1476), lambda (:1575-1578), the three init calls (:1601-1603),
 * then per 262144-byte block: fill d->buf, rotate_90_u8_f32(d), full_demod(d), read
 * d->result_len values from d->result (:870-889, :904-908).
 *
 * Input = the survey's LCG byte stream (seed 12345) over 40 blocks; prints the number of PCM
 * values and their 64-bit FNV-style hash.  SURVEY.md section 8c records what the reference gives:
 * 209714 values, hash c3e7eda4bd16dfe1 (tests/test_gpu_dropin_c.py compares).
 *
 *   usage: dropin_check [blocks]          FMD_MATH_FAST=1 selects the +-1 LSB kernels
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>

#include "fmdemod_mi355x.h"

static struct demod_state demod;      /* zero-initialised global, like the program's */

int main(int argc, char **argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 40;
  struct demod_state *d = &demod;
  demod_init(d);
  d->rate_in = d->rate_out = 300000;                 /* -s 300000 */
  d->rate_out2 = 48000;                              /* output.rate */
  d->lpr.mode = 2;                                   /* -X */
  d->lpr.size = 90;
  d->deemph_lambda = fmd_deemph_lambda(48000, d->deemph);
  init_u8_f32_table();
  init_lp_f32();
  init_lp_real_f32(d);

  uint32_t s = 12345;
  uint64_t h = 1469598103934665603ULL;
  long total = 0;
  for (int b = 0; b < blocks; b++) {
    for (uint32_t i = 0; i < FMD_MAXIMUM_BUF_LENGTH; i++) {
      s = s * 1664525u + 1013904223u;
      d->buf[i] = (uint8_t)(s >> 24);
    }
    d->buf_len = FMD_MAXIMUM_BUF_LENGTH;
    if (!d->offset_tuning) rotate_90_u8_f32(d);
    else u8_f32(d);
    full_demod(d);
    for (int i = 0; i < d->result_len; i++) {
      h ^= (uint64_t)(uint16_t)d->result[i];
      h *= 1099511628211ULL;
    }
    total += d->result_len;
  }
  printf("%ld %016" PRIx64 "\n", total, h);
  deinit_lp_real_f32(d);
  return 0;
}
