/*
 * fmd_internal.h - private interface between the C host layer (fmd_host.c)
 * and the HIP kernel launchers (fmd_kernels.inc, one translation unit per kernel family).  Not installed, not exported
 * (csrc/fmdemod_mi355x.map).
 */
#ifndef FMD_INTERNAL_H
#define FMD_INTERNAL_H

#include <stdint.h>

#include "fmdemod_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FMDK_TILE 512         /* rate_in samples per tile: 64 lanes x 8 outputs           */
#define FMDK_WAVES 4          /* workers (wavefronts) per workgroup                       */
#define FMDK_DG_N 416         /* bytes per (byte phase, limb) of the decimating second stage's tap tables: composite L+R filter, */
#define FMDK_DF_N 288         /* ... fm (90 taps), */
#define FMDK_DM_N 368         /* ... fm (128 taps, mono) */
#define FMDK_DEC_K0G 180      /* the window's first sample before the tile's first, per filter (resample_tile_dec / resample_mono_dec) */
#define FMDK_DEC_K0F 92
#define FMDK_DEC_K0M 128
#ifndef FMDK_FRAME_CAP
#define FMDK_FRAME_CAP 512    /* pending resampler outputs (floats) per worker before a flush */
#endif

/* Uniform launch parameters, passed by value in the kernarg segment so that
 * tap reads with constant indices become scalar loads. */
typedef struct fmdk_params {
  float fb[16];           /* /8 IQ low-pass, half (reference lp_filter_f32)      */
  float fbs[16];          /* fast path: fb / 128 (exact scaling)                 */
  float c_i, c_q;         /* fast path: constant terms of the folded offset      */
  float fm[128], fp[128], fs[128];
  float fm_sh[128];       /* fm shifted by one tap (fm_sh[i] = fm[i + 1]): tap pairs that start on an odd tap as
                             aligned scalar pairs (resample_mono_2to1) */
  float swf, cwf, lambda, coef;
  float lam_pow[16];        /* lambda^(j+1), j = 0..15: the fast kernels' blocked de-emphasis */
  float lam_scan[8];        /* fast kernels' per-tile flush: lambda^(flush_g 2^k), k = 0..7 (zero with de-emphasis off) */
  float lam_eff;            /* lambda, or 0 with de-emphasis off (the fast flush then passes x through)  */
  float log2_a;             /* log2(lambda^flush_g), -1e30 with de-emphasis off: a^(i+1) = exp2((i+1) log2_a) */
  float car_inv_k2;         /* fast stereo: 1 / K^2, K = radius of (x, y) per unit |vs| below which the
                               regenerated 38 kHz carrier is redone exactly (fmd_kernels.inc, carrier_fast) */
  float inv_slow;           /* 1.0f / (float)slow: the generic emit-index form's estimate (made by the host: as a division in the kernel's prologue it held a register for the kernel's whole life) */
  float inv_fast;           /* 1.0f / (float)fast: tile_frames' estimate, likewise */
  float car_inv_k2_l2;      /* ... and 1 / (L K)^2: below L K the redo recomputes the window from the IQ words, between L K and K it
                               sums the worker's own window in the reference's order (redo_carrier, step 0)                 */
  int32_t size, half, mode;
  int32_t slow, fast;     /* rate_out2, rate_out                                 */
  int32_t resample;       /* rate_out2 > 0                                       */
  int32_t deemph, offset_tuning;
  int32_t warm;           /* de-emphasis warm-up frames for restarted segments   */
  int32_t block_len;      /* bytes per block                                     */
  int32_t n_blocks;
  int32_t pcm_stride;     /* int16 per (stream, block)                           */
  int32_t n_streams;
  int32_t n_chunks;       /* time chunks (workers) per stream                    */
  int32_t perm4;          /* resampler: four frames are a whole, odd number of samples apart (lane map)   */
  int32_t warm_fast;      /* frames after which a zero de-emphasis state is right to 1e-9 */
  int32_t warm_tiles;     /* tiles a chunk > 0 replays before its first tile     */
  uint32_t emit_magic, emit_shift;   /* resampler emit index: floor(n / slow) = mulhi(n, magic) >> shift for n < 2^29
                                        (0: slow too small for a 32-bit magic number, the kernel divides by float estimate) */
  int32_t mono_2to1;      /* fast mono, 128 taps, rate_out == 2 rate_out2: the four-frames-per-lane resampler */
  uint32_t tf_magic, tf_shift;       /* frames of a tile: floor(x / fast) = mulhi(x, magic) >> shift for x < 2^30 (0: divide) */
  int32_t flush_g;        /* fast kernels: frames per lane in the per-tile flush (2 mono only, 4 or 8): the smallest with
                             ceil(frames per tile / flush_g) x channels <= 64 lanes */
  /* matrix-pipe form of stage A (FMD_MATH_FAST_MFMA, v_mfma_i32_16x16x64_i8): the 32 decimator taps with the j^n
   * rotation signs as 26-bit fixed point, E = sgn * round(fb * 2^26) = l0 2^16 + l1 2^8 + l2 with balanced int8 limbs.
   * a_tab[limb][comp][d][4 dwords] = the 16 window bytes (8 IQ samples, I and Q slots) that taps 8d .. 8d+7 occupy in
   * the sum of component comp (0 = I, 1 = Q): the A operand of lane (row = 2 r + comp, g) for K slice s is entry
   * d = 4 s + g - r, zero outside 0..3 (fmd_host.c, build_a_tab). */
  int32_t a_tab[3 * 2 * 4 * 4];
  float a_bias_i, a_bias_q;          /* 2^-34 * sum of E over the window: the (u - 127.5) offset of the reference's table */
  /* matrix-pipe form of stage C (FMD_MATH_FAST_MFMA_F, 90-tap stereo): taps of filter f (fm, fp, fs) as T = round(h 2^qf) in three
   * balanced int8 limbs (the kernel builds its byte tables from fm / fp / fs and qf), samples as round(v 2^20):
   * y = ci_scale[f] (A0 + A1 2^-8 + A2 2^-16 + A3 2^-24), ci_scale = 2^(32 - 20 - qf)  (fmd_kernels.inc, mpx_tile_i8) */
  int32_t ci_qf[3];
  float ci_scale[3];
  /* ... whose second stage reads limbs: stage C hands (L-R) x carrier over as round(x 2^20) in three int8 limbs,
   * so its sums are put together at 2^20 times their value: ci_scale_q = 2^20 ci_scale, and the carrier's margin r^2 / K^2 - vs^2 is
   * taken with the scaled vs: car_inv_k2_q = 2^40 / K^2 (stage_c.inc, mpx_tile_i8; stage_d.inc, resample_tile_dec) */
  float ci_scale_q[3];
  float car_inv_k2_q;
  /* fast kernels: decimated samples with |I| + |Q| <= org_thr are redone in the reference's arithmetic (stage B).  1e-3 where it was
   * validated (narrow FM at the reference's default volume and everything with a larger PCM step per unit of discriminator error);
   * grows with coef x (largest tap of the filter behind the discriminator) beyond that: the phase error of such a sample is
   * (decimator difference) / magnitude and reaches the PCM through one tap */
  /* FMD_MATH_FAST_MFMA_F stereo: the L+R chain as one filter g = fm * fm (179 taps, symmetric) over the discriminator output: T_g = round(g 2^g_qf)
   * in three balanced int8 limbs, gq[u] = T_g[u] = T_g[178 - u] for u <= 89; y = g_scale (A0 + A1 2^-8 + A2 2^-16), g_scale = 2^(12 - g_qf);
   * g_unit = 2^-g_qf turns a T_g into its tap for the cold paths (cold_paths.inc: q1_patch_i8, lr_head_fix) */
  int32_t gq[90];
  int32_t g_qf;
  float g_scale, g_unit;
  int32_t dec_p;                 /* FMD_MATH_FAST_MFMA_F: 16 rate_out / rate_out2 = samples per sixteen frames (a multiple of four in 64 .. 100), 0: the family does not apply (resample_tile_dec) */
  const void *dec_tables;        /* ... device memory, made once per batch by the host (fmd_host.c, build_dec_tables): the sixteen byte phases of the reversed, zero-padded
                                    limb tables the decimating second stage reads - stereo: 16 x 3 x FMDK_DG_N bytes of the composite filter, then 16 x 3 x FMDK_DF_N of
                                    fm; 128-tap mono: 16 x 3 x FMDK_DM_N of fm.  The kernel's prologue copies them into LDS (one 16-byte word per thread and step) */
  int32_t dec_wide;              /* ... mono: more than eight groups of sixteen frames per tile (rate_out < 4 rate_out2): a column per group (resample_mono_dec) */
  int32_t pilot_pairs8;          /* matrix-pipe stage C: the pilot filter's class-3 limb pairs too (volume >= 1: the carrier's accuracy in LSB scales with it) */
  float org_thr, org_thr15;      /* (and 1.5 x it: the lane-level pre-test on max(|cross|, |dot|)) */
} fmdk_params;

/* Launch the fused IQ->PCM kernel for n_streams streams.  Returns 0 or a
 * hipError_t (> 0).  All pointers are device pointers.  ev_start / ev_stop: hipEvent_t or NULL - recorded with the
 * kernel's own dispatch packet (hipExtLaunchKernelGGL), not as packets of their own around it. */
int fmdk_launch(const fmdk_params *p, int math, int n_streams, const void *d_iq, void *d_pcm,
                void *d_lens, const void *d_state_in, void *d_state_out, const fmd_debug_taps *dbg,
                void *hip_stream, void *ev_start, void *ev_stop);
/* Tiles a time chunk must replay so that every FIR history is exact and the
 * de-emphasis recurrence has converged (0: the launch must not be split). */
int fmdk_warm_tiles(const fmdk_params *p, int math);
int fmdk_tile(void);
int fmdk_workers_per_cu(int math);
int fmdk_workers_per_cu_mode(int math, int mode);   /* the same by lpr.mode (the mono kernels may be budgeted differently) */
/* Mangled-free kernel name as rocprofv3 prints it (prefix match). */
const char *fmdk_kernel_name(const fmdk_params *p, int math);
/* Static LDS bytes of the fused kernel (for DESIGN.md / diagnostics). */
int fmdk_lds_bytes(void);

#ifdef __cplusplus
}
#endif
#endif
