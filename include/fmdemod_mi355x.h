/*
 * fmdemod_mi355x.h - C ABI of the MI355X-native FM demodulation path.
 *
 * Drop-in boundary for the IQ -> PCM hot path of rtl_fm_player
 * (rotate_90_u8_f32 + full_demod, reference src/rtl_fm_player.c:206-226,
 * :758-788).  Plain C, plain pointers and sizes; no HIP or torch types appear
 * in any signature.  Everything here is exported by libfmdemod_mi355x.so
 * (rtl_fm_player_amd/csrc).  The library runs every stage on the GPU with
 * hand-written gfx950 kernels; there is no CPU fallback and every entry point
 * fails (status < 0, or abort() for the void reference-shaped calls) when no
 * HIP device is usable.
 *
 * Three groups of entry points:
 *   1. the reference's own operator surface (same names, same struct layout)
 *      so rtl_fm_player.c can link against this library instead of its own
 *      definitions;
 *   2. a batch API (many independent streams x many blocks per launch) which
 *      is what the bench and the parity tests drive;
 *   3. an rtlsdr_read_async_cb_t-compatible ingest callback feeding a
 *      per-stream pinned staging ring.
 */
#ifndef FMDEMOD_MI355X_H
#define FMDEMOD_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------
 * 1. Reference-shaped surface
 * ------------------------------------------------------------------------
 * Layout-compatible restatement of struct lp_real / struct demod_state
 * (reference include/rtl_fm_player.h:95-110, :127-175; x86-64 glibc).  The
 * reference header cannot be included from a second translation unit (it
 * defines globals), so the layout is restated here and checked with static
 * assertions in csrc/fmd_host.c against the offsets recorded in SURVEY.md
 * section 8a (row a15).  Define FMD_NO_REFERENCE_TYPES before including this
 * header from a TU that already has the reference's own definitions.
 */
#define FMD_DEFAULT_BUF_LENGTH (1 * 16384)                 /* include/rtl_fm_player.h:31 */
#define FMD_MAXIMUM_OVERSAMPLE 16                          /* :32 */
#define FMD_MAXIMUM_BUF_LENGTH (FMD_MAXIMUM_OVERSAMPLE * FMD_DEFAULT_BUF_LENGTH) /* :33 */

#ifndef FMD_NO_REFERENCE_TYPES
/* pthread_rwlock_t is POSIX.1-2001: compile with -std=gnu11 (the reference's CMake default dialect)
 * or define _POSIX_C_SOURCE >= 200112L / _GNU_SOURCE before the first include under strict -std=c11 */
#include <pthread.h>

struct output_state;

struct lp_real {          /* include/rtl_fm_player.h:95-110 */
  float *br;              /* ring of discriminator samples                */
  float *bm;              /* ring of L+R low-pass outputs                 */
  float *bs;              /* ring of demodulated L-R samples              */
  float *fm;              /* half of the symmetric 0..16 kHz low-pass     */
  float *fp;              /* half of the 18..20 kHz pilot band-pass       */
  float *fs;              /* half of the 21..55 kHz L-R band-pass         */
  float swf;              /* sin(2 pi 19000 / rate_in)                    */
  float cwf;              /* cos(2 pi 19000 / rate_in)                    */
  float pp;               /* previous pilot band-pass output              */
  int pos;                /* ring write index                             */
  int size;               /* taps (90 stereo, 128 mono)                   */
  int rsize;              /* size / 2                                     */
  int mode;               /* 0 drop, 1 mono, 2 stereo                     */
};

struct demod_state {      /* include/rtl_fm_player.h:127-175 */
  int exit_flag;
  pthread_t thread;
  uint8_t buf[FMD_MAXIMUM_BUF_LENGTH];
  uint32_t buf_len;
  int16_t lowpassed[FMD_MAXIMUM_BUF_LENGTH << 1];
  int lp_len;
  float lowpass_tb[48];
  int16_t lp_i_hist[10][6];
  int16_t lp_q_hist[10][6];
  int16_t result[FMD_MAXIMUM_BUF_LENGTH];
  int result_len;
  int16_t droop_i_hist[9];
  int16_t droop_q_hist[9];
  int offset_tuning;
  int rate_in;
  int rate_out;
  int rate_out2;
  int now_r, now_j;
  int pre_r, pre_j;
  float pre_r_f32, pre_j_f32;
  int prev_index;
  int downsample;
  int post_downsample;
  int output_scale;
  int squelch_level, conseq_squelch, squelch_hits, terminate_on_squelch;
  int downsample_passes;
  int comp_fir_size;
  int custom_atan;
  double deemph;
  int deemph_a;
  int deemph_l;
  int deemph_r;
  float deemph_l_f32;
  float deemph_r_f32;
  float deemph_lambda;
  float volume;
  int now_lpr;
  int prev_lpr_index;
  struct lp_real lpr;
  pthread_rwlock_t rw;
  pthread_cond_t ready;
  pthread_mutex_t ready_m;
  struct output_state *output_target;
};

/* Same names, arguments and calling protocol as the reference:
 *   fill d->buf / d->buf_len, call rotate_90_u8_f32(d) (or u8_f32(d)), then
 *   full_demod(d), then read d->result_len int16 values from d->result
 *   (src/rtl_fm_player.c:870-889, :904-908).
 * rotate_90_u8_f32 / u8_f32 only record which conversion the next full_demod
 * applies (and set lp_len like the reference); full_demod uploads d->buf,
 * runs the whole chain on the GPU, downloads the PCM and mirrors every
 * mutable state field back into *d (lowpass_tb, pre_*_f32, lpr rings/pos/pp,
 * prev_lpr_index, deemph_*_f32), so CPU and GPU calls can be interleaved on
 * one struct.  They return void like the originals; an unusable device or an
 * unsupported configuration aborts with a message on stderr.
 *
 * One synchronisation per block (round 4): the carried state stays on the device between calls; the library keeps a
 * copy of what it last mirrored into the struct and uploads the struct's state only when the caller has changed it
 * (reset, restore, a CPU block in between) - IQ up, kernel, {PCM, length, new state} down into one pinned block, one
 * wait.  bench.py's `single_stream` leg: 0.10 ms per 262144-byte block against 1.04 ms for the reference on one host core.
 *
 * Limits of this surface (it serves one dongle, like the program it drops into; many streams
 * belong on the batch API below):
 *   - the device side of a struct is found through a registry keyed by the struct's address
 *     (the reference struct has no spare field); fmd_demod_release (or deinit_lp_real_f32) forgets a struct;
 *   - the arithmetic contract is read ONCE, at the first full_demod of the process, from the
 *     environment: FMD_MATH_FAST set -> +-1 LSB kernels, otherwise the bit-exact ones;
 *   - any HIP error is fatal (abort), because the signatures have no way to report it. */
void init_u8_f32_table(void);                       /* src/rtl_fm_player.c:195 */
void init_lp_f32(void);                             /* :241 */
void init_lp_real_f32(struct demod_state *fm);      /* :413 */
void deinit_lp_real_f32(struct demod_state *fm);    /* :455 */
void demod_init(struct demod_state *s);             /* :1156 */
void rotate_90_u8_f32(struct demod_state *d);       /* :206 */
void u8_f32(struct demod_state *d);                 /* :228 */
void full_demod(struct demod_state *d);             /* :758 */
/* Additive: release the device resources full_demod attached to *d. */
void fmd_demod_release(struct demod_state *d);

#endif /* FMD_NO_REFERENCE_TYPES */

/* Additive (declared whether or not FMD_NO_REFERENCE_TYPES is defined: a program that keeps its own struct definitions calls these too): what the void
 * reference-shaped calls cannot say through their signatures.
 *   fmd_dropin_set_math: the arithmetic family of the calls above (an fmd_config.math value), instead of the FMD_MATH_FAST environment variable; call it
 *     before the first full_demod (a struct's batch is rebuilt when the family changes).
 *   fmd_dropin_set_error_handler: by default a failure inside one of the calls above (no device, out of memory, a HIP error) prints a line and abort()s; with
 *     a handler installed it is told (which call, fmd_last_error()'s text) and the call returns with result_len = 0. */
int fmd_dropin_set_math(int math);
typedef void (*fmd_dropin_error_fn)(const char *where, const char *message, void *ctx);
void fmd_dropin_set_error_handler(fmd_dropin_error_fn fn, void *ctx);


/* ------------------------------------------------------------------------
 * 2. Batch API: n_streams independent demodulators, many blocks per launch
 * ------------------------------------------------------------------------ */

/* status codes (0 ok, < 0 error) */
#define FMD_OK 0
#define FMD_E_ARG (-1)          /* bad argument                              */
#define FMD_E_UNSUPPORTED (-2)  /* configuration outside the supported range */
#define FMD_E_NOMEM (-3)
#define FMD_E_HIP (-4)          /* HIP runtime error (see fmd_last_error)    */
#define FMD_E_NODEVICE (-5)     /* no usable gfx950 device                   */
#define FMD_E_STATE (-6)        /* call sequence error                       */

/* arithmetic contract */
#define FMD_MATH_EXACT 0  /* reference operation order, unfused mul/add: bit-exact PCM */
#define FMD_MATH_FAST 1   /* PCM within +-1 LSB: the fastest kernel family of this build for the configuration - FMD_MATH_FAST_MFMA_F where it
                             applies, FMD_MATH_FAST_MFMA otherwise (FMD_MATH_FAST_VALU for a caller's decimator taps beyond the 26-bit form);
                             a caller who wants a particular family names it here instead (the library reads no environment
                             variable for this).  fmd_batch_math() says what a batch runs.  Sharing the device with other
                             MFMA kernels: one packed-fp32 instruction form computed wrong results beside them (round 3); it is
                             gone from the kernels, and every build's device code is linted for it (tools/isa_lint.py over the
                             disassembly of the built library, run by the Makefile) - profiles/archive/r04_pk_opsel_hazard.md.  The
                             hazard is an empirical description of undocumented hardware behaviour: tests/test_gpu_neighbour.py
                             and tests/test_gpu_coresidency.py are the standing check. */
#define FMD_MATH_FAST_VALU 2   /* +-1 LSB, vector ALU only: fused multiply-adds in the reference's summation order */
#define FMD_MATH_FAST_MFMA 3   /* +-1 LSB, matrix pipe beside the vector ALU: the /8 decimator as exact int8 products
                                  of the IQ bytes with 26-bit fixed-point taps (v_mfma_i32_16x16x64_i8); any filter size, ragged tiles */
#define FMD_MATH_FAST_MFMA_F 7 /* +-1 LSB, every stage that has a matrix form there, in int8-limb fixed point with exact integer sums
                                  (samples round(v 2^20), taps round(h 2^qf)).  90-tap stereo: the pilot and L-R filters at full rate
                                  (src/rtl_fm_player.c:538-566), the L+R channel's two low-passes (:545 / :560, :588) as ONE 179-tap
                                  filter fm * fm and the second stage of L-R, both evaluated at the resampler's emit instants only, as
                                  the reference does (:570-598): a decimating banded product - rows = sixteen consecutive frames,
                                  columns = (group of sixteen frames, sample limb).  128-tap mono / narrow FM: the fm low-pass likewise
                                  (:500-532).  Needs whole tiles (block_len a multiple of 8192), sixteen frames a whole number P of
                                  samples, P a multiple of four (stereo: 64 .. 100 - 300 k, 240 k, 192 k -> 48 k; mono: 32 .. 128), and
                                  the fixed-point error estimates of DESIGN.md section 2a below 0.15 LSB (volume up to ~8 at 300 k; narrow
                                  FM up to ~1.7); other configurations run FMD_MATH_FAST_MFMA under this name */
/* retired in round 6 (tools/experiments/retired_round5_families.inc): the intermediate matrix-pipe families of rounds 4 and 5 - stage C alone
 * (_MFMA_C), the second stage at every sample (_MFMA_D), the composite L+R filter at every sample (_MFMA_E).  The names are accepted and mean
 * FMD_MATH_FAST. */
#define FMD_MATH_FAST_MFMA_C 4
#define FMD_MATH_FAST_MFMA_D 5
#define FMD_MATH_FAST_MFMA_E 6

typedef struct fmd_config {
  int32_t rate_in;        /* demod_state.rate_in                               */
  int32_t rate_out;       /* demod_state.rate_out  (resampler "fast")          */
  int32_t rate_out2;      /* demod_state.rate_out2 (resampler "slow"), <=0 off */
  int32_t mode;           /* lpr.mode 0/1/2                                    */
  int32_t size;           /* lpr.size (even, <= 256)                           */
  int32_t deemph;         /* nonzero: de-emphasis on                           */
  int32_t offset_tuning;  /* nonzero: no fs/4 rotation (u8_f32 path)           */
  float deemph_lambda;    /* demod_state.deemph_lambda                         */
  float volume;           /* demod_state.volume                                */
  int32_t block_len;      /* bytes of u8 IQ per block (reference: 262144);     */
                          /* multiple of 16, >= 64                             */
  int32_t math;           /* FMD_MATH_EXACT / FMD_MATH_FAST (/ _VALU / _MFMA / _MFMA_F) */
} fmd_config;

/* Filter tables; fmd_design_taps() fills them exactly as init_lp_f32 /
 * init_lp_real_f32 do (host libm).  A caller may override them (tests hand
 * the device path the very tables the oracle used). */
typedef struct fmd_taps {
  float fb[16];
  float fm[128], fp[128], fs[128];
  float swf, cwf;
} fmd_taps;

/* Carried state of one stream, linear oldest -> newest (mirrors the mutable
 * fields of struct demod_state / struct lp_real). */
typedef struct fmd_stream_state {
  float tb[48];
  float pre_r, pre_j;
  float pp;
  float deemph_l, deemph_r;
  int32_t acc;            /* prev_lpr_index */
  int32_t reserved[2];
  float br[256], bm[256], bs[256];   /* last `size` values, oldest first, at [0..size) */
} fmd_stream_state;

/* Optional stage taps for debugging / parity tests: device pointers or NULL.
 * Shapes per stream and block, M = block_len / 16:
 *   y [2*M] f32, v [M] f32 (before the Q1 overwrite), mpx [M] f32 (resampler
 *   output before de-emphasis, result_len entries used).  prof: see below.
 * A launch with at least one tap runs the debug build of its kernel (same arithmetic, plus the checks
 * and stores the taps need); launches without taps run the build that has none of them (~1 % faster). */
typedef struct fmd_debug_taps {
  void *y, *v, *mpx;
  void *prof;   /* i64 [n_streams][16]: shader-clock cycles per stage, summed over the launch
                   (0 load, 1 decimate, 2 discriminate, 3 q1, 4 mpx, 5 per-tile flush of the fast kernels, 6 resample,
                    7 roll, 8 flush, 9 state in/out, 15 total) */
} fmd_debug_taps;

typedef struct fmd_batch fmd_batch;

int fmd_design_taps(const fmd_config *cfg, fmd_taps *out);
float fmd_deemph_lambda(int output_rate, double tau);   /* src/rtl_fm_player.c:1577 */

/* device < 0: current device.  taps == NULL: fmd_design_taps(cfg). */
int fmd_batch_create(fmd_batch **out, const fmd_config *cfg, const fmd_taps *taps,
                     int n_streams, int device);
void fmd_batch_destroy(fmd_batch *b);

/* int16 slots per (stream, block) in the PCM buffer (multiple of 8). */
int fmd_batch_pcm_stride(const fmd_batch *b);
int fmd_batch_n_streams(const fmd_batch *b);
/* The kernel family this batch runs: FMD_MATH_EXACT, FMD_MATH_FAST_VALU, FMD_MATH_FAST_MFMA or FMD_MATH_FAST_MFMA_F (FMD_MATH_FAST
 * in the configuration resolves to one of the last three at creation; a named family the configuration cannot run resolves likewise). */
int fmd_batch_math(const fmd_batch *b);
/* The same question without a device or a batch: the family fmd_batch_create would run for this configuration (and these taps; NULL:
 * fmd_design_taps), or a negative status for a configuration it would refuse. */
int fmd_config_family(const fmd_config *cfg, const fmd_taps *taps);
/* What the fixed-point second stage of FMD_MATH_FAST_MFMA_F adds to a PCM value for this configuration, in LSB - whether or not the configuration runs it
 * (`family` says what it resolves to): per filter (stereo: [0] the composite L+R filter fm * fm, 179 taps, [1] fm over (L-R) x carrier; mono: [0] fm) the
 * rms ESTIMATE the family is gated on (limit_rms_lsb) and a worst-case BOUND with its three terms - samples rounded to 2^-20, taps rounded to 2^-qf, the
 * limb pairs left out (csrc/fmd_host.c, fixed_point_error).  The other stages' differences to the reference (fused multiply-adds,
 * v_rcp_f32, the exact redo of ill-conditioned samples) have no bound of this kind: tests/, the fuzz and the volume scans vouch for them.  No device needed. */
typedef struct fmd_error_estimate {
  int32_t family, filters;
  struct { int32_t taps, qf; float rms_lsb, worst_lsb, worst_samples_lsb, worst_taps_lsb, worst_dropped_lsb; } f[2];
  float limit_rms_lsb;
} fmd_error_estimate;
int fmd_config_error_estimate(const fmd_config *cfg, const fmd_taps *taps, fmd_error_estimate *out);
/* How a launch is cut into time chunks (one worker wavefront each; results do not depend on it - the tests hold the
 * library to that through this call): workers_per_cu > 0 = cut until the grid offers that many workers per CU,
 * 0 = the kernels' own figure (default), < 0 = never cut (one worker per stream). */
int fmd_batch_set_time_split(fmd_batch *b, int workers_per_cu);

/* fmd_batch_destroy waits for everything the batch has queued (on its own streams and on the
 * caller's stream of the most recent launch, which must therefore still exist) and detaches the
 * ingest rings bound to it: they stay valid and are destroyed by their owner with
 * fmd_ingest_destroy, before or after the batch. */

/* Device-resident run.  Layouts (all device pointers):
 *   d_iq   u8  [n_streams][n_blocks][block_len]
 *   d_pcm  s16 [n_streams][n_blocks][pcm_stride]
 *   d_lens i32 [n_streams][n_blocks]          (result_len of each block)
 * hip_stream: a hipStream_t passed as void* (NULL = the batch's own stream).
 * d_iq must be 16-byte aligned.  Asynchronous; state advances by n_blocks blocks
 * per stream.  Internally each stream's tiles are cut into time chunks so that
 * every CU holds 12 workers; results do not depend on that.
 * Stream rule: a batch's launches form ONE sequence (each reads the state the one before
 * wrote).  They may be queued on different streams - the library inserts the event wait when
 * the stream changes between two launches - but calls on one fmd_batch must come from one
 * thread at a time, and a caller's stream must outlive the work queued on it.  get_state /
 * set_state / reset / sync / destroy wait for the most recent launch whatever stream it is on. */
int fmd_batch_run_device(fmd_batch *b, const void *d_iq, int n_blocks, void *d_pcm,
                         void *d_lens, void *hip_stream);
int fmd_batch_run_device_debug(fmd_batch *b, const void *d_iq, int n_blocks, void *d_pcm,
                               void *d_lens, void *hip_stream, const fmd_debug_taps *dbg);
int fmd_batch_sync(fmd_batch *b);
/* The buffers of a launch must be READY on the stream it runs on: a caller that fills d_iq (or allocates d_pcm / d_lens from a
 * stream-ordered allocator) on another stream synchronises that stream first - or calls this: everything queued on
 * producer_stream so far happens before whatever this batch launches afterwards on its OWN stream (one event record + one wait,
 * no host synchronisation).  Launches with an explicit hip_stream are ordered by that stream and need none of it. */
int fmd_batch_wait_stream(fmd_batch *b, void *producer_stream);

/* Host-buffer convenience (H2D, run, D2H, sync); same layouts in host memory. */
int fmd_batch_run_host(fmd_batch *b, const uint8_t *iq, int n_blocks, int16_t *pcm,
                       int32_t *lens);

/* Per-stream carried state (synchronises first). */
int fmd_batch_get_state(fmd_batch *b, int stream, fmd_stream_state *out);
int fmd_batch_set_state(fmd_batch *b, int stream, const fmd_stream_state *in);
int fmd_batch_reset(fmd_batch *b);

/* Duration of the most recent fmd_batch_run_device kernel, measured with HIP
 * events recorded on the stream the kernel was launched on (synchronises). */
int fmd_batch_last_kernel_ms(fmd_batch *b, float *ms);
/* Every launch is bracketed by an event pair for fmd_batch_last_kernel_ms; a caller that
 * launches back to back and times the whole run itself can turn that off (on = 0): the two
 * event records cost about 10 us of GPU time per launch. */
int fmd_batch_set_timing(fmd_batch *b, int on);
/* Name of the dominant kernel (as rocprofv3 reports it) for this batch. */
const char *fmd_batch_kernel_name(const fmd_batch *b);

const char *fmd_last_error(void);
int fmd_device_count(void);

/* ------------------------------------------------------------------------
 * 3. Ingest: rtlsdr_read_async callback -> pinned staging ring -> batch
 * ------------------------------------------------------------------------
 * fmd_ingest_callback has the exact rtlsdr_read_async_cb_t signature
 * (reference include/rtl-sdr.h:340) and the contract of rtlsdr_callback
 * (src/rtl_fm_player.c:790-837): it copies `len` bytes before returning and
 * never blocks on the GPU.  ctx is an fmd_ingest* bound to one stream of a batch.
 * The ring is pinned memory and is itself the source of the H2D copies.
 *
 * Overflow.  DEFAULT IS NOT THE REFERENCE'S: FMD_OVERFLOW_DROP_OLDEST splits a copy at the end
 * of the ring and, when the ring is full, advances the read position over the oldest bytes (a clean
 * loss, counted).  The reference instead restarts a transfer that does not fit before the end of
 * the ring at offset 0 and, on overflow, clamps its byte count without moving the read position, so
 * the demod thread then reads new data where old was expected (src/rtl_fm_player.c:813-834).
 * FMD_OVERFLOW_REFERENCE reproduces exactly that, for callers that want identical behaviour. */
#define FMD_OVERFLOW_DROP_OLDEST 0
#define FMD_OVERFLOW_REFERENCE 1
typedef struct fmd_ingest fmd_ingest;
typedef void (*fmd_read_async_cb_t)(unsigned char *buf, uint32_t len, void *ctx);

/* ring_bytes 0: the reference's 4 MiB (16 blocks).  b == NULL makes an unbound ring in ordinary
 * host memory (no device needed; `stream` ignored) that its owner drains with fmd_ingest_pop. */
int fmd_ingest_create(fmd_ingest **out, fmd_batch *b, int stream, uint32_t ring_bytes);
void fmd_ingest_destroy(fmd_ingest *g);
int fmd_ingest_set_overflow(fmd_ingest *g, int mode);
void fmd_ingest_callback(unsigned char *buf, uint32_t len, void *ctx);
/* Bytes buffered that no job has taken yet / bytes dropped so far.  Thread-safe, like the callback
 * and fmd_ingest_mute: the callback may run on any thread (librtlsdr's event thread in the reference)
 * while another thread pumps. */
uint32_t fmd_ingest_buffered(const fmd_ingest *g);
uint64_t fmd_ingest_dropped(const fmd_ingest *g);
/* The dequeue of demod_thread_fn (src/rtl_fm_player.c:863-876): when at least len bytes are
 * buffered, copies them to out and returns len, else returns 0.  Not while pump jobs are in flight. */
uint32_t fmd_ingest_pop(fmd_ingest *g, uint8_t *out, uint32_t len);
/* Mute the first n bytes of the next callback buffer (retune, :805-810). */
void fmd_ingest_mute(fmd_ingest *g, int n_bytes);
/* Demodulate whole blocks that every bound stream has buffered: returns the
 * number of blocks processed per stream (>= 0) or an error.  pcm/lens as in
 * fmd_batch_run_host, for max_blocks blocks. */
int fmd_batch_pump(fmd_batch *b, int max_blocks, int16_t *pcm, int32_t *lens);
/* Pipelined form of the same: _begin queues the H2D of what is buffered (straight from the
 * pinned rings), the kernel and the D2H, and returns the job's block count at once (0: nothing
 * buffered); _end waits for the oldest job begun and hands out its PCM and lengths.
 * Two jobs may be in flight, so the H2D of one overlaps the kernel of the other (the demod
 * thread's copy / demodulate alternation, src/rtl_fm_player.c:871-889, without the
 * serialisation).  A job's bytes stay in the ring until their H2D has finished: size the ring
 * (fmd_ingest_create's ring_bytes) at twice the job when the producer must not wait. */
int fmd_batch_pump_begin(fmd_batch *b, int max_blocks);
int fmd_batch_pump_end(fmd_batch *b, int16_t *pcm, int32_t *lens);

/* ------------------------------------------------------------------------
 * 4. WAV output in the reference's format (InitWaveOut / CloseWaveOut,
 *    src/rtl_fm_player.c:1259-1328; headers include/rtl_fm_player.h:216-253)
 * ------------------------------------------------------------------------
 * 260-byte header (44-byte PCM header, 16 bit / 48000 Hz / mode == 2 ? stereo :
 * mono, then 216 zero bytes); close patches the sizes at offsets 4 and 40.
 * path "-" writes to stdout (sizes then stay at the reference's placeholders). */
#define FMD_WAV_HEADER_BYTES 260
typedef struct fmd_wav fmd_wav;
int fmd_wav_header(int mode, unsigned char out[FMD_WAV_HEADER_BYTES]);
int fmd_wav_open(fmd_wav **out, const char *path, int mode);
int fmd_wav_write(fmd_wav *w, const int16_t *pcm, size_t n_values);
int fmd_wav_close(fmd_wav *w);

#ifdef __cplusplus
}
#endif
#endif /* FMDEMOD_MI355X_H */
