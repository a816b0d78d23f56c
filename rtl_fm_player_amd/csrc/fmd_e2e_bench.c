/*
 * fmd_e2e_bench.c - the H2D-inclusive rate of the path without any Python in the loop.
 *
 * The reference's two threads around full_demod (src/rtl_fm_player.c): the dongle thread's rtlsdr_callback copies each
 * 262144-byte USB transfer into a ring (:790-837), the demod thread takes whole blocks out of it and demodulates
 * (:855-933).  Here N feeder pthreads play the dongle threads of S streams - each calls fmd_ingest_callback, one
 * 262144-byte transfer at a time, from ordinary (pageable) host buffers, pacing itself on the ring's free space the way
 * a dongle paces itself by time - while the main thread is the demod thread of all S streams:
 * fmd_batch_pump_begin / _end with two jobs in flight (H2D straight from the pinned rings, kernel, PCM back to host).
 * Wall clock from the first callback to the last PCM.  Then the same threads measure what the host side alone can do:
 * plain memcpy of the same transfers into pinned memory and into ordinary memory.
 *
 *   fmd_e2e_bench [-S streams=64] [-B blocks per job=16] [-J jobs=20] [-T feeder threads=16] [-m mode 2|1] [-e]
 *                 [-i rate_in=300000] [-o rate_out2=48000] [-z lpr.size] [-M math code 0..5 (fmdemod_mi355x.h)] [-W stall seconds=60]
 * prints ONE JSON line on stdout (the configuration it really ran is part of it).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define FMD_NO_REFERENCE_TYPES
#include "fmdemod_mi355x.h"

#include <hip/hip_runtime_api.h>

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

enum { BL = FMD_MAXIMUM_BUF_LENGTH };

typedef struct {
  int tid, n_thr, n_streams, blocks_per_job;
  long long blocks_total;            /* per stream */
  fmd_ingest **rings;
  uint32_t ring_bytes;
  uint8_t **src;                     /* per stream: blocks_per_job * BL bytes of IQ, replayed round and round */
  atomic_int *stop;
  double busy_s;                     /* time inside fmd_ingest_callback */
} feeder_t;

static void *feeder(void *arg) {
  feeder_t *f = (feeder_t *)arg;
  const int cap = (f->n_streams + f->n_thr - 1) / f->n_thr;      /* every stream has a feeder, whatever S and T are */
  long long *done = (long long *)calloc((size_t)cap, sizeof(*done));
  int *mine = (int *)calloc((size_t)cap, sizeof(*mine)), n_mine = 0;
  if (!done || !mine) { atomic_store(f->stop, 1); free(done); free(mine); return NULL; }
  for (int s = f->tid; s < f->n_streams; s += f->n_thr) { mine[n_mine] = s; done[n_mine++] = 0; }
  int left = n_mine;
  while (left > 0 && !atomic_load(f->stop)) {
    int progressed = 0;
    for (int i = 0; i < n_mine; i++) {
      if (done[i] >= f->blocks_total) continue;
      fmd_ingest *g = f->rings[mine[i]];
      /* never overflow the ring: a dongle delivers at its own rate, this one as fast as there is room */
      if (fmd_ingest_buffered(g) + (uint32_t)BL > f->ring_bytes) continue;
      const double t0 = now_s();
      fmd_ingest_callback(f->src[mine[i]] + (size_t)(done[i] % f->blocks_per_job) * BL, BL, g);
      f->busy_s += now_s() - t0;
      if (++done[i] == f->blocks_total) left--;
      progressed = 1;
    }
    if (!progressed) sched_yield();
  }
  free(done); free(mine);
  return NULL;
}

typedef struct { uint8_t *dst, *src; size_t bytes; int reps; double s; } cp_t;
static void *copier(void *arg) {
  cp_t *c = (cp_t *)arg;
  const double t0 = now_s();
  for (int r = 0; r < c->reps; r++)
    for (size_t o = 0; o + BL <= c->bytes; o += BL) memcpy(c->dst + o, c->src + o, BL);
  c->s = now_s() - t0;
  return NULL;
}

/* aggregate GB/s of n_thr threads each copying `bytes` per repetition in 262144-byte pieces into dst[t] */
static double copy_rate(int n_thr, uint8_t **dst, uint8_t **src, size_t bytes, int reps) {
  pthread_t th[256];
  cp_t c[256];
  const double t0 = now_s();
  for (int t = 0; t < n_thr; t++) {
    c[t] = (cp_t){dst[t], src[t], bytes, reps, 0.0};
    pthread_create(&th[t], NULL, copier, &c[t]);
  }
  for (int t = 0; t < n_thr; t++) pthread_join(th[t], NULL);
  return (double)n_thr * (double)bytes * reps / (now_s() - t0) / 1e9;
}

int main(int argc, char **argv) {
  int S = 64, B = 16, J = 20, T = 16, mode = 2, exact = 0, opt;
  int rate_in = 300000, rate_out2 = 48000, fsize = 0, math = -1;
  double stall_s = 60.0;
  while ((opt = getopt(argc, argv, "S:B:J:T:m:ei:o:z:M:W:h")) != -1) {
    switch (opt) {
      case 'S': S = atoi(optarg); break;
      case 'B': B = atoi(optarg); break;
      case 'J': J = atoi(optarg); break;
      case 'T': T = atoi(optarg); break;
      case 'm': mode = atoi(optarg); break;
      case 'e': exact = 1; break;
      case 'i': rate_in = atoi(optarg); break;
      case 'o': rate_out2 = atoi(optarg); break;
      case 'z': fsize = atoi(optarg); break;
      case 'M': math = atoi(optarg); break;
      case 'W': stall_s = atof(optarg); break;
      default: fprintf(stderr, "usage: fmd_e2e_bench [-S streams] [-B blocks/job] [-J jobs] [-T feeder threads] [-m mode] [-e]\n"); return opt == 'h' ? 0 : 2;
    }
  }
  if (S < 1 || S > 16384 || B < 1 || J < 2 || T < 1 || T > 256) { fprintf(stderr, "fmd_e2e_bench: bad arguments\n"); return 2; }
  if (T > S) T = S;
  if (fsize <= 0) fsize = mode == 1 ? 128 : 90;
  if (math < 0) math = exact ? FMD_MATH_EXACT : FMD_MATH_FAST;
  fmd_config cfg = {rate_in, rate_in, rate_out2, mode, fsize, 1, 0, 0.f, 0.4f, BL, math};
  cfg.deemph_lambda = fmd_deemph_lambda(rate_out2 > 0 ? rate_out2 : rate_in, 50e-6);
  fmd_batch *b = NULL;
  if (fmd_batch_create(&b, &cfg, NULL, S, -1)) { fprintf(stderr, "fmd_e2e_bench: %s\n", fmd_last_error()); return 1; }
  const int stride = fmd_batch_pcm_stride(b);
  const uint32_t ring_bytes = 2u * (uint32_t)B * BL;          /* two jobs deep: one being copied to the device, one filling */
  fmd_ingest **rings = (fmd_ingest **)calloc((size_t)S, sizeof(*rings));
  uint8_t **src = (uint8_t **)calloc((size_t)S, sizeof(*src));
  int16_t *pcm = (int16_t *)malloc((size_t)S * B * stride * sizeof(int16_t));
  int32_t *lens = (int32_t *)malloc((size_t)S * B * sizeof(int32_t));
  if (!rings || !src || !pcm || !lens) return 1;
  uint32_t x = 12345u;
  for (int s = 0; s < S; s++) {
    if (fmd_ingest_create(&rings[s], b, s, ring_bytes)) { fprintf(stderr, "fmd_e2e_bench: %s\n", fmd_last_error()); return 1; }
    src[s] = (uint8_t *)malloc((size_t)B * BL);
    if (!src[s]) return 1;
    for (size_t i = 0; i < (size_t)B * BL; i++) { x = x * 1664525u + 1013904223u; src[s][i] = (uint8_t)(x >> 24); }
  }

  atomic_int stop = 0;
  pthread_t th[256];
  feeder_t fd[256];
  const long long blocks_total = (long long)B * (J + 2);       /* two untimed jobs first: buffers allocated, clocks up */
  for (int t = 0; t < T; t++) {
    fd[t] = (feeder_t){t, T, S, B, blocks_total, rings, ring_bytes, src, &stop, 0.0};
    pthread_create(&th[t], NULL, feeder, &fd[t]);
  }
  /* the demod thread: a job starts when every stream has B blocks buffered (:863-868 polls the same way) */
  int begun = 0, ended = 0, rc = 0;
  double t0 = 0, kernel_wait = 0, last_progress = now_s();
  unsigned long long pcm_values = 0;
  while (ended < J + 2) {
    if (atomic_load(&stop) || now_s() - last_progress > stall_s) {   /* a feeder gave up, or nothing has moved for stall_s seconds */
      fprintf(stderr, "fmd_e2e_bench: stalled (%d jobs begun, %d ended)\n", begun, ended);
      atomic_store(&stop, 1);
      break;
    }
    if (begun < J + 2 && begun - ended < 2) {
      int ready = 1;
      for (int s = 0; s < S && ready; s++) ready = fmd_ingest_buffered(rings[s]) >= (uint32_t)B * BL;
      if (ready) {
        rc = fmd_batch_pump_begin(b, B);
        if (rc != B) { fprintf(stderr, "fmd_e2e_bench: pump_begin -> %d %s\n", rc, fmd_last_error()); atomic_store(&stop, 1); break; }
        begun++;
        last_progress = now_s();
        continue;
      }
      if (begun == ended) { sched_yield(); continue; }          /* nothing in flight: wait for the feeders */
    }
    const double w0 = now_s();
    rc = fmd_batch_pump_end(b, pcm, lens);
    kernel_wait += now_s() - w0;
    if (rc != B) { fprintf(stderr, "fmd_e2e_bench: pump_end -> %d %s\n", rc, fmd_last_error()); atomic_store(&stop, 1); break; }
    ended++;
    last_progress = now_s();
    if (ended == 2) { t0 = now_s(); kernel_wait = 0; }
    if (ended > 2) for (int i = 0; i < S * B; i++) pcm_values += (unsigned long long)lens[i];
  }
  const double dt = now_s() - t0;
  for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
  if (ended < J + 2) return 1;
  double busy = 0;
  for (int t = 0; t < T; t++) busy += fd[t].busy_s;
  const double nbytes = (double)J * S * B * BL;

  /* what the host side alone can do with the same transfers */
  uint8_t *pin[256], *pag[256], *sr[256];
  const size_t per_thr = (size_t)128 * BL;                     /* 32 MiB per thread: beyond the last-level cache share of a core */
  int have_pin = 1;
  for (int t = 0; t < T; t++) {
    sr[t] = (uint8_t *)malloc(per_thr);
    if (sr[t]) memset(sr[t], 2, per_thr);
    pag[t] = (uint8_t *)malloc(per_thr);
    if (hipHostMalloc((void **)&pin[t], per_thr, hipHostMallocDefault) != hipSuccess) have_pin = 0;
    if (pag[t]) memset(pag[t], 1, per_thr);
    if (have_pin) memset(pin[t], 1, per_thr);
  }
  const double cp_pin = have_pin ? copy_rate(T, pin, sr, per_thr, 4) : 0.0;
  const double cp_pag = copy_rate(T, pag, sr, per_thr, 4);
  const double cp_one = copy_rate(1, pag, sr, per_thr, 4);
  long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
  cpu_set_t set;
  int affinity = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : -1;

  printf("{\"value\": %.1f, \"unit\": \"Msamples/s\", \"pcie_gbs\": %.2f, \"config\": {\"rate_in\": %d, \"rate_out2\": %d, \"mode\": %d, \"size\": %d, "
         "\"math_requested\": %d, \"math_run\": %d}, \"streams\": %d, \"blocks_per_job\": %d, \"jobs\": %d, "
         "\"feeder_threads\": %d, \"ms_per_job\": %.3f, \"pcm_values\": %llu, \"callback_gbs_per_thread\": %.2f, "
         "\"demod_thread_wait_s\": %.3f, \"memcpy_gbs\": {\"threads_to_pinned\": %.2f, \"threads_to_pageable\": %.2f, \"one_thread_to_pageable\": %.2f}, "
         "\"cpus_online\": %ld, \"cpus_in_affinity_mask\": %d, "
         "\"path\": \"C, no Python: %d pthreads call fmd_ingest_callback (262144-byte transfers from pageable memory) -> pinned rings "
         "(2 jobs deep) -> H2D straight from the rings, fmd_batch_pump_begin/_end with two jobs in flight -> PCM in host memory; wall clock\"}\n",
         nbytes / 2 / dt / 1e6, nbytes / dt / 1e9, rate_in, rate_out2, mode, fsize, math, fmd_batch_math(b), S, B, J, T, dt / J * 1e3, pcm_values, nbytes / busy / 1e9 , kernel_wait, cp_pin, cp_pag, cp_one,
         ncpu, affinity, T);
  for (int s = 0; s < S; s++) { fmd_ingest_destroy(rings[s]); free(src[s]); }
  fmd_batch_destroy(b);
  return 0;
}
