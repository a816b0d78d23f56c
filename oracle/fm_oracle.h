/*
 * fm_oracle.h - CPU restatement of the rtl_fm_player IQ->PCM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker /
 * the timed CPU baseline.  The product (rtl_fm_player_amd/csrc) never links
 * or calls it and fails loudly when its HIP library is missing.
 *
 * What it restates (reference = /root/reference, file:line):
 *   u8 -> f32 tables            src/rtl_fm_player.c:195-204
 *   fs/4 rotation / plain u8    src/rtl_fm_player.c:206-239
 *   /8 IQ low-pass taps         src/rtl_fm_player.c:241-251
 *   /8 IQ low-pass              src/rtl_fm_player.c:253-411
 *   MPX filter design           src/rtl_fm_player.c:413-453
 *   38 kHz carrier regeneration src/rtl_fm_player.c:472-481
 *   resampler (modes 0/1/2)     src/rtl_fm_player.c:483-604
 *   polynomial atan2            src/rtl_fm_player.c:606-667
 *   FM discriminator            src/rtl_fm_player.c:669-685
 *   de-emphasis                 src/rtl_fm_player.c:687-709
 *   f32 -> s16                  src/rtl_fm_player.c:711-735
 *   chain order (full_demod)    src/rtl_fm_player.c:758-788
 *
 * Parity pin: PINNED TO THE REFERENCE ITSELF.  oracle/build_ref.py compiles the
 * reference's hot path (src/rtl_fm_player.c:195-788 and the type / table lines of
 * include/rtl_fm_player.h it needs) from the sources where they lie into
 * oracle/_ref/libref.so - no stand-in headers, nothing of the reference in this
 * repository - and tests/test_ref_pin.py holds this restatement against it bit
 * for bit: PCM, block lengths, per-stage intermediates, carried state and filter
 * tables on the five survey configurations (whose recorded hashes libref.so also
 * reproduces), nine further configurations, FM-broadcast input and a 100-case
 * configuration fuzz with ragged block lengths.  tests/golden/ref_vectors.npz
 * holds reference outputs (tests/golden/make_ref_fixtures.py) for boxes without
 * /root/reference; tests/test_golden_vectors.py checks the oracle and, directly,
 * the HIP path against them.
 *
 * Build: -O3 -ffp-contract=off, no -ffast-math, no -march (the reference's
 * CMake Release build has no FMA contraction).
 */
#ifndef FM_ORACLE_H
#define FM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Parameters the reference keeps in struct demod_state
 * (include/rtl_fm_player.h:127-175). */
typedef struct fmo_config {
  int32_t rate_in;        /* demod_state.rate_in: filter design rate            */
  int32_t rate_out;       /* demod_state.rate_out: "fast" in the resampler      */
  int32_t rate_out2;      /* demod_state.rate_out2: "slow"; <= 0 skips resample */
  int32_t mode;           /* lpr.mode: 0 drop, 1 mono, 2 stereo                 */
  int32_t size;           /* lpr.size: 90 stereo / 128 mono                     */
  int32_t deemph;         /* nonzero: de-emphasis enabled (demod_state.deemph)  */
  int32_t offset_tuning;  /* nonzero: plain u8->f32, no fs/4 rotation           */
  float deemph_lambda;    /* demod_state.deemph_lambda                          */
  float volume;           /* demod_state.volume                                 */
} fmo_config;

/* State carried from one block to the next (the mutable fields of
 * struct demod_state / struct lp_real), in linear oldest->newest order. */
typedef struct fmo_state {
  float tb[48];           /* lowpass_tb: last 24 complex samples of the block   */
  float pre_r, pre_j;     /* pre_r_f32 / pre_j_f32                              */
  float pp;               /* lpr.pp: previous pilot band-pass output            */
  float deemph_l, deemph_r;
  int32_t acc;            /* prev_lpr_index                                     */
  int32_t pos;            /* lpr.pos (ring write index the reference would hold)*/
  int32_t size;           /* lpr.size                                           */
  float br[256];          /* last `size` discriminator samples, oldest first    */
  float bm[256];          /* last `size` L+R low-pass outputs, oldest first     */
  float bs[256];          /* last `size` demodulated L-R samples, oldest first  */
} fmo_state;

/* Optional per-block intermediates for stage-by-stage debugging. */
typedef struct fmo_trace {
  float *y;               /* [2*M] decimated IQ (I,Q interleaved)               */
  float *v;               /* [M]   discriminator output (before any overwrite)  */
  float *mpx;             /* [result_len] resampler output before de-emphasis   */
} fmo_trace;

typedef struct fmo_stream fmo_stream;

fmo_stream *fmo_open(const fmo_config *cfg);
void fmo_close(fmo_stream *s);

/* rotate_90_u8_f32 (or u8_f32) + full_demod on one block of `len` bytes of
 * interleaved u8 IQ.  `len` must be a multiple of 16 and >= 64.  Writes the
 * int16 PCM to pcm (capacity >= len/16 values) and returns result_len, or a
 * negative value on a bad argument. */
int fmo_block(fmo_stream *s, const uint8_t *iq, uint32_t len, int16_t *pcm);
int fmo_block_trace(fmo_stream *s, const uint8_t *iq, uint32_t len, int16_t *pcm,
                    const fmo_trace *tr);

/* n_blocks consecutive blocks; pcm blocks are written back to back, lens[b]
 * receives each result_len.  Returns the total number of int16 written. */
long fmo_run(fmo_stream *s, const uint8_t *iq, uint32_t len, int n_blocks,
             int16_t *pcm, int32_t *lens);

void fmo_get_state(const fmo_stream *s, fmo_state *out);
void fmo_set_state(fmo_stream *s, const fmo_state *in);

/* Filter tables (for fixtures and for handing identical taps to the device
 * path in tests).  fb: 16 floats; fm/fp/fs: size/2 floats each. */
void fmo_get_taps(const fmo_stream *s, float *fb, float *fm, float *fp, float *fs,
                  float *swf, float *cwf);

/* (float) exp(-1 / (rate * tau))  -- src/rtl_fm_player.c:1575-1578 */
float fmo_deemph_lambda(int output_rate, double tau);

/* ---- synthetic inputs and hashing ---- */
/* LCG bytes: s = s*1664525 + 1013904223 (uint32), byte = s >> 24. */
void fmo_lcg_fill(uint32_t *state, uint8_t *buf, size_t n);
/* 64-bit FNV-1a style hash over int16 units: h ^= (uint16)x; h *= prime. */
uint64_t fmo_hash16(uint64_t h, const int16_t *x, size_t n);
#define FMO_HASH_INIT 1469598103934665603ULL

/* Integer-only DDS FM multiplex generator: stereo WBFM (L/R tones, 19 kHz
 * pilot) FM-modulated on a carrier at -fs/4, quantised to u8 IQ.  Pure
 * integer arithmetic on a quarter-wave sine table built with integer
 * recurrences, so every platform regenerates identical bytes. */
typedef struct fmo_dds {
  uint32_t ph_l, ph_r, ph_pilot, ph_carrier, noise;
  uint32_t step_l, step_r, step_pilot;   /* phase steps per IQ sample */
  int32_t dev_q;      /* peak deviation as phase step (per unit mpx)   */
  int32_t amp;        /* IQ amplitude in LSB (<= 127)                  */
  int32_t stereo;     /* 0: mono programme only                        */
} fmo_dds;
void fmo_dds_init(fmo_dds *d, int fs, int f_left, int f_right, int amp, int stereo,
                  uint32_t seed);
void fmo_dds_fill(fmo_dds *d, uint8_t *buf, size_t n_bytes);

#ifdef __cplusplus
}
#endif
#endif
