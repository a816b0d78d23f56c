/*
 * fmd_node_bench.c - the path on every MI355X of one node, host side in C: one process, one demod thread per device.
 *
 * north_star (BASELINE.json): "independent stations / IQ streams shard embarrassingly across the 8 GPUs of one node with RCCL over xGMI used only
 * to gather throughput counters", host code in C.  Streams are independent (no term of src/rtl_fm_player.c:195-788 couples two demod_states), so
 * stream s of S_total lives on device s / streams_per_device for good: its IQ, its carried state and its PCM never leave that device, and no
 * collective sits on the data path.  This tool is the node-level driver of that sharding:
 *
 *   per device d: a demod thread bound to the CPUs of the device's NUMA node (sysfs: /sys/bus/pci/devices/<bdf>/numa_node), a batch of
 *   streams_per_device streams created on that device (the library calls hipSetDevice per call), and
 *     leg 1 "resident": the IQ of every stream resident in HBM, W untimed + K timed launches of n_blocks blocks - all devices start the timed
 *                       region together (a pthread barrier) and the node's rate is (all devices' samples) / (the slowest device's time);
 *     leg 2 "h2d":      T feeder pthreads per device (bound to the same NUMA node) play the dongle threads - fmd_ingest_callback, one 262144-byte
 *                       transfer at a time into the device's pinned rings - while the demod thread pumps (fmd_batch_pump_begin / _end, two jobs in
 *                       flight): H2D-inclusive Msamples/s and PCIe GB/s per device and for the node;
 *   then the counters {samples, elapsed ns, PCM values} of every device are gathered with ncclAllGather - librccl directly, one communicator per
 *   device in this one process (ncclCommInitAll), no torch, no MPI - and the line is printed from the gathered copy (device 0's).
 *
 *   fmd_node_bench [-d devices (default: all)] [-s total streams (default 256 per device)] [-B blocks per launch=16] [-K timed launches=20]
 *                  [-W warm-up launches=5] [-J h2d jobs=6 (0: skip leg 2)] [-T feeder threads per device=8] [-m mode 2|1] [-i rate_in] [-o rate_out2]
 *                  [-n (no RCCL: print from the host's own copy)] [-P (plan only: the stream -> device map as JSON, no device touched)]
 * prints ONE JSON line.  Unmeasured on multi-GPU hardware so far: the pool this repository is built on hands out one device per box
 * (DESIGN.md section 6); with -d 1 it runs there and is what tests/test_gpu_node_bench.py checks.
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
#include <rccl/rccl.h>

enum { BL = FMD_MAXIMUM_BUF_LENGTH, MAXDEV = 16, MAXT = 64 };

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* ---- sharding: stream s -> device s / per, the last device takes what is left ---- */
typedef struct { int first, count; } shard_t;
static int shard_plan(int total, int ndev, shard_t *out) {
  if (total < ndev || ndev < 1) return -1;
  const int per = (total + ndev - 1) / ndev;
  int first = 0;
  for (int d = 0; d < ndev; d++) {
    int c = total - first < per ? total - first : per;
    if (c <= 0) return -1;                       /* (a device without streams: ask for fewer devices) */
    out[d] = (shard_t){first, c};
    first += c;
  }
  return first == total ? 0 : -1;
}

/* ---- NUMA: the CPUs next to a device ---- */
static int device_numa_node(int dev) {
  char bdf[64] = {0}, path[160];
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), dev) != hipSuccess) return -1;
  for (char *p = bdf; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE *f = fopen(path, "r");
  int node = -1;
  if (f) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
  return node;
}
/* "0-31,64-95" -> cpu set; 0 on success */
static int numa_cpus(int node, cpu_set_t *set) {
  char path[96], buf[4096];
  CPU_ZERO(set);
  if (node < 0) return -1;
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  if (!fgets(buf, sizeof(buf), f)) { fclose(f); return -1; }
  fclose(f);
  int n = 0;
  for (char *p = buf; *p && *p != '\n';) {
    char *e;
    long a = strtol(p, &e, 10), b = a;
    if (e == p) break;
    if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, set); n++; }
    p = *e == ',' ? e + 1 : e;
  }
  return n > 0 ? 0 : -1;
}
static int bind_to_node(int node) {
  cpu_set_t set, cur;
  if (numa_cpus(node, &set)) return 0;
  /* only CPUs this process may use at all (a container's mask) */
  if (sched_getaffinity(0, sizeof(cur), &cur) == 0) {
    cpu_set_t both;
    CPU_AND(&both, &set, &cur);
    if (CPU_COUNT(&both) == 0) return 0;
    set = both;
  }
  return pthread_setaffinity_np(pthread_self(), sizeof(set), &set) == 0 ? CPU_COUNT(&set) : 0;
}

/* ---- per-device context ---- */
typedef struct {
  int dev, numa, cpus_bound, n_streams, first_stream;
  int B, K, W, J, T, mode, rate_in, rate_out2;
  pthread_barrier_t *bar;
  atomic_int *fail;
  /* results */
  int math_run;
  double resident_s, h2d_s;
  unsigned long long resident_samples, resident_pcm, h2d_samples, h2d_pcm;
  float kernel_ms;
  char err[256];
} devctx;

typedef struct {
  int tid, n_thr, n_streams, blocks_per_job, numa;
  long long blocks_total;
  fmd_ingest **rings;
  uint32_t ring_bytes;
  uint8_t *src;                      /* blocks_per_job * BL bytes of IQ, replayed round and round (every stream the same bytes: a throughput tool) */
  atomic_int *stop;
} feeder_t;

static void *feeder(void *arg) {
  feeder_t *f = (feeder_t *)arg;
  bind_to_node(f->numa);
  const int cap = (f->n_streams + f->n_thr - 1) / f->n_thr;
  long long *done = (long long *)calloc((size_t)cap, sizeof(*done));
  int *mine = (int *)calloc((size_t)cap, sizeof(*mine)), n_mine = 0;
  if (!done || !mine) { atomic_store(f->stop, 1); free(done); free(mine); return NULL; }
  for (int s = f->tid; s < f->n_streams; s += f->n_thr) mine[n_mine++] = s;
  int left = n_mine;
  while (left > 0 && !atomic_load(f->stop)) {
    int progressed = 0;
    for (int i = 0; i < n_mine; i++) {
      if (done[i] >= f->blocks_total) continue;
      fmd_ingest *g = f->rings[mine[i]];
      if (fmd_ingest_buffered(g) + (uint32_t)BL > f->ring_bytes) continue;      /* never overflow: as fast as there is room */
      fmd_ingest_callback(f->src + (size_t)(done[i] % f->blocks_per_job) * BL, BL, g);
      if (++done[i] == f->blocks_total) left--;
      progressed = 1;
    }
    if (!progressed) sched_yield();
  }
  free(done); free(mine);
  return NULL;
}

#define DFAIL(c, ...) do { snprintf((c)->err, sizeof((c)->err), __VA_ARGS__); atomic_store((c)->fail, 1); } while (0)

static void *device_main(void *arg) {
  devctx *c = (devctx *)arg;
  c->cpus_bound = bind_to_node(c->numa);
  const int S = c->n_streams, B = c->B;
  fmd_batch *b = NULL;
  uint8_t *h_iq = NULL, *d_iq = NULL;
  int16_t *d_pcm = NULL, *h_pcm = NULL;
  int32_t *d_lens = NULL, *h_lens = NULL;
  fmd_ingest **rings = NULL;
  int ok = 0;
  do {
    if (hipSetDevice(c->dev) != hipSuccess) { DFAIL(c, "hipSetDevice(%d) failed", c->dev); break; }
    fmd_config cfg = {c->rate_in, c->rate_in, c->rate_out2, c->mode, c->mode == 1 ? 128 : 90, 1, 0, 0.f, 0.4f, BL, FMD_MATH_FAST};
    cfg.deemph_lambda = fmd_deemph_lambda(c->rate_out2 > 0 ? c->rate_out2 : c->rate_in, 50e-6);
    if (fmd_batch_create(&b, &cfg, NULL, S, c->dev)) { DFAIL(c, "device %d: %s", c->dev, fmd_last_error()); break; }
    c->math_run = fmd_batch_math(b);
    const int stride = fmd_batch_pcm_stride(b);
    /* one stream's worth of synthetic IQ (LCG bytes, seeded by the device's first stream), the same for every stream of the device */
    h_iq = (uint8_t *)malloc((size_t)B * BL);
    h_lens = (int32_t *)malloc((size_t)S * B * sizeof(int32_t));
    if (!h_iq || !h_lens) { DFAIL(c, "device %d: out of host memory", c->dev); break; }
    uint32_t x = 12345u + (uint32_t)c->first_stream;
    for (size_t i = 0; i < (size_t)B * BL; i++) { x = x * 1664525u + 1013904223u; h_iq[i] = (uint8_t)(x >> 24); }
    if (hipMalloc((void **)&d_iq, (size_t)S * B * BL) != hipSuccess || hipMalloc((void **)&d_pcm, (size_t)S * B * stride * sizeof(int16_t)) != hipSuccess ||
        hipMalloc((void **)&d_lens, (size_t)S * B * sizeof(int32_t)) != hipSuccess) { DFAIL(c, "device %d: hipMalloc failed", c->dev); break; }
    int bad = 0;
    for (int s = 0; s < S && !bad; s++) bad = hipMemcpy(d_iq + (size_t)s * B * BL, h_iq, (size_t)B * BL, hipMemcpyHostToDevice) != hipSuccess;
    if (bad) { DFAIL(c, "device %d: H2D of the synthetic IQ failed", c->dev); break; }
    ok = 1;
  } while (0);

  /* ---- leg 1: resident.  Every device thread passes the same barriers whether it is healthy or not ---- */
  pthread_barrier_wait(c->bar);                               /* everybody is set up */
  if (ok && !atomic_load(c->fail)) {
    for (int i = 0; i < c->W && ok; i++) ok = fmd_batch_run_device(b, d_iq, B, d_pcm, d_lens, NULL) == 0;
    if (ok) ok = fmd_batch_sync(b) == 0;
    if (!ok) DFAIL(c, "device %d: warm-up launch failed: %s", c->dev, fmd_last_error());
  }
  pthread_barrier_wait(c->bar);                               /* the timed region starts on every device together */
  const double t0 = now_s();
  if (ok && !atomic_load(c->fail)) {
    for (int i = 0; i < c->K && ok; i++) ok = fmd_batch_run_device(b, d_iq, B, d_pcm, d_lens, NULL) == 0;
    if (ok) ok = fmd_batch_sync(b) == 0;
    if (!ok) DFAIL(c, "device %d: timed launch failed: %s", c->dev, fmd_last_error());
  }
  c->resident_s = now_s() - t0;
  pthread_barrier_wait(c->bar);
  if (ok && !atomic_load(c->fail)) {
    (void)fmd_batch_last_kernel_ms(b, &c->kernel_ms);
    if (hipMemcpy(h_lens, d_lens, (size_t)S * B * sizeof(int32_t), hipMemcpyDeviceToHost) == hipSuccess) {
      unsigned long long v = 0;
      for (int i = 0; i < S * B; i++) v += (unsigned long long)h_lens[i];
      c->resident_pcm = v * (unsigned long long)c->K;
    }
    c->resident_samples = (unsigned long long)c->K * S * B * (BL / 2);
  }

  /* ---- leg 2: H2D-inclusive (feeders -> pinned rings -> pump) ---- */
  if (c->J > 0) {
    atomic_int stop = 0;
    pthread_t th[MAXT];
    feeder_t fd[MAXT];
    int T = c->T > S ? S : c->T, started = 0;
    const uint32_t ring_bytes = 2u * (uint32_t)B * BL;
    const long long blocks_total = (long long)B * (c->J + 2);
    if (ok && !atomic_load(c->fail)) {
      (void)fmd_batch_reset(b);
      const int stride = fmd_batch_pcm_stride(b);
      rings = (fmd_ingest **)calloc((size_t)S, sizeof(*rings));
      h_pcm = (int16_t *)malloc((size_t)S * B * stride * sizeof(int16_t));
      if (!rings || !h_pcm) { DFAIL(c, "device %d: out of host memory", c->dev); ok = 0; }
      for (int s = 0; s < S && ok; s++)
        if (fmd_ingest_create(&rings[s], b, s, ring_bytes)) { DFAIL(c, "device %d: %s", c->dev, fmd_last_error()); ok = 0; }
    }
    pthread_barrier_wait(c->bar);
    if (ok && !atomic_load(c->fail)) {
      for (int t = 0; t < T; t++) {
        fd[t] = (feeder_t){t, T, S, B, c->numa, blocks_total, rings, ring_bytes, h_iq, &stop};
        if (pthread_create(&th[t], NULL, feeder, &fd[t]) == 0) started++;
      }
      int begun = 0, ended = 0, rc = 0;
      double t1 = 0, last = now_s();
      unsigned long long pv = 0;
      while (ended < c->J + 2) {
        if (atomic_load(&stop) || atomic_load(c->fail) || now_s() - last > 60.0) { DFAIL(c, "device %d: h2d leg stalled (%d begun, %d ended)", c->dev, begun, ended); break; }
        if (begun < c->J + 2 && begun - ended < 2) {
          int ready = 1;
          for (int s = 0; s < S && ready; s++) ready = fmd_ingest_buffered(rings[s]) >= (uint32_t)B * BL;
          if (ready) {
            rc = fmd_batch_pump_begin(b, B);
            if (rc != B) { DFAIL(c, "device %d: pump_begin -> %d %s", c->dev, rc, fmd_last_error()); break; }
            begun++; last = now_s();
            continue;
          }
          if (begun == ended) { sched_yield(); continue; }
        }
        rc = fmd_batch_pump_end(b, h_pcm, h_lens);
        if (rc != B) { DFAIL(c, "device %d: pump_end -> %d %s", c->dev, rc, fmd_last_error()); break; }
        ended++; last = now_s();
        if (ended == 2) t1 = now_s();                          /* two untimed jobs first */
        if (ended > 2) for (int i = 0; i < S * B; i++) pv += (unsigned long long)h_lens[i];
      }
      c->h2d_s = now_s() - t1;
      atomic_store(&stop, ended < c->J + 2);
      for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
      if (ended == c->J + 2) { c->h2d_samples = (unsigned long long)c->J * S * B * (BL / 2); c->h2d_pcm = pv; }
    }
    pthread_barrier_wait(c->bar);
  }
  if (rings) { for (int s = 0; s < S; s++) if (rings[s]) fmd_ingest_destroy(rings[s]); free(rings); }
  if (b) fmd_batch_destroy(b);
  if (d_iq) (void)hipFree(d_iq);
  if (d_pcm) (void)hipFree(d_pcm);
  if (d_lens) (void)hipFree(d_lens);
  free(h_iq); free(h_pcm); free(h_lens);
  return NULL;
}

/* ---- the counter gather: ncclAllGather of {samples, elapsed ns, PCM values} x 2 legs per device, librccl directly ---- */
enum { NCNT = 6 };
static int gather_counters(int ndev, const int *devs, const unsigned long long (*mine)[NCNT], unsigned long long *all /* [ndev][NCNT], device 0's copy */,
                           char *err, size_t errlen) {
  ncclComm_t comms[MAXDEV];
  hipStream_t st[MAXDEV];
  unsigned long long *snd[MAXDEV] = {0}, *rcv[MAXDEV] = {0};
  int rc = -1, made = 0;
  ncclResult_t nr = ncclCommInitAll(comms, ndev, devs);
  if (nr != ncclSuccess) { snprintf(err, errlen, "ncclCommInitAll: %s", ncclGetErrorString(nr)); return -1; }
  do {
    int bad = 0;
    for (int d = 0; d < ndev && !bad; d++) {
      bad = hipSetDevice(devs[d]) != hipSuccess || hipStreamCreate(&st[d]) != hipSuccess;
      if (!bad) made = d + 1;
      bad = bad || hipMalloc((void **)&snd[d], NCNT * sizeof(unsigned long long)) != hipSuccess ||
            hipMalloc((void **)&rcv[d], (size_t)ndev * NCNT * sizeof(unsigned long long)) != hipSuccess ||
            hipMemcpy(snd[d], mine[d], NCNT * sizeof(unsigned long long), hipMemcpyHostToDevice) != hipSuccess;
    }
    if (bad) { snprintf(err, errlen, "gather: device buffers"); break; }
    nr = ncclGroupStart();
    for (int d = 0; d < ndev && nr == ncclSuccess; d++) nr = ncclAllGather(snd[d], rcv[d], NCNT, ncclUint64, comms[d], st[d]);
    if (nr == ncclSuccess) nr = ncclGroupEnd(); else (void)ncclGroupEnd();
    if (nr != ncclSuccess) { snprintf(err, errlen, "ncclAllGather: %s", ncclGetErrorString(nr)); break; }
    for (int d = 0; d < ndev && !bad; d++) bad = hipSetDevice(devs[d]) != hipSuccess || hipStreamSynchronize(st[d]) != hipSuccess;
    if (bad || hipSetDevice(devs[0]) != hipSuccess ||
        hipMemcpy(all, rcv[0], (size_t)ndev * NCNT * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) { snprintf(err, errlen, "gather: read-back"); break; }
    rc = 0;
  } while (0);
  for (int d = 0; d < ndev; d++) {
    (void)hipSetDevice(devs[d]);
    if (snd[d]) (void)hipFree(snd[d]);
    if (rcv[d]) (void)hipFree(rcv[d]);
    if (d < made) (void)hipStreamDestroy(st[d]);
    (void)ncclCommDestroy(comms[d]);
  }
  return rc;
}

int main(int argc, char **argv) {
  int ndev = 0, total = 0, B = 16, K = 20, W = 5, J = 6, T = 8, mode = 2, rate_in = 300000, rate_out2 = 48000, rccl = 1, plan = 0, opt;
  while ((opt = getopt(argc, argv, "d:s:B:K:W:J:T:m:i:o:nPh")) != -1) {
    switch (opt) {
      case 'd': ndev = atoi(optarg); break;
      case 's': total = atoi(optarg); break;
      case 'B': B = atoi(optarg); break;
      case 'K': K = atoi(optarg); break;
      case 'W': W = atoi(optarg); break;
      case 'J': J = atoi(optarg); break;
      case 'T': T = atoi(optarg); break;
      case 'm': mode = atoi(optarg); break;
      case 'i': rate_in = atoi(optarg); break;
      case 'o': rate_out2 = atoi(optarg); break;
      case 'n': rccl = 0; break;
      case 'P': plan = 1; break;
      default:
        fprintf(stderr, "usage: fmd_node_bench [-d devices] [-s total streams] [-B blocks] [-K launches] [-W warm-up] [-J h2d jobs] [-T feeders/device] [-m mode] [-i rate_in] [-o rate_out2] [-n] [-P]\n");
        return opt == 'h' ? 0 : 2;
    }
  }
  if (plan) {                                     /* the stream -> device map, no device touched (tests/test_dist_cpu.py) */
    if (ndev < 1) ndev = 8;
    if (total < 1) total = 256 * ndev;
    shard_t sh[MAXDEV];
    if (ndev > MAXDEV || shard_plan(total, ndev, sh)) { fprintf(stderr, "fmd_node_bench: cannot place %d streams on %d devices\n", total, ndev); return 2; }
    printf("{\"devices\": %d, \"streams\": %d, \"shards\": [", ndev, total);
    for (int d = 0; d < ndev; d++) printf("%s{\"device\": %d, \"first_stream\": %d, \"streams\": %d}", d ? ", " : "", d, sh[d].first, sh[d].count);
    printf("], \"data_path_collectives\": 0}\n");
    return 0;
  }
  const int have = fmd_device_count();
  if (have < 1) { fprintf(stderr, "fmd_node_bench: no HIP device: the MI355X path has no CPU fallback\n"); return 1; }
  if (ndev < 1) ndev = have;
  if (ndev > have || ndev > MAXDEV) { fprintf(stderr, "fmd_node_bench: %d devices asked for, %d present\n", ndev, have); return 2; }
  if (total < 1) total = 256 * ndev;
  if (B < 1 || K < 1 || W < 0 || J < 0 || T < 1 || T > MAXT) { fprintf(stderr, "fmd_node_bench: bad arguments\n"); return 2; }
  shard_t sh[MAXDEV];
  if (shard_plan(total, ndev, sh)) { fprintf(stderr, "fmd_node_bench: cannot place %d streams on %d devices\n", total, ndev); return 2; }

  pthread_barrier_t bar;
  pthread_barrier_init(&bar, NULL, (unsigned)ndev);
  atomic_int fail = 0;
  devctx ctx[MAXDEV];
  pthread_t th[MAXDEV];
  int devs[MAXDEV];
  memset(ctx, 0, sizeof(ctx));
  for (int d = 0; d < ndev; d++) {
    devs[d] = d;
    ctx[d] = (devctx){.dev = d, .numa = device_numa_node(d), .n_streams = sh[d].count, .first_stream = sh[d].first, .B = B, .K = K, .W = W, .J = J, .T = T,
                      .mode = mode, .rate_in = rate_in, .rate_out2 = rate_out2, .bar = &bar, .fail = &fail};
    pthread_create(&th[d], NULL, device_main, &ctx[d]);
  }
  for (int d = 0; d < ndev; d++) pthread_join(th[d], NULL);
  pthread_barrier_destroy(&bar);
  if (atomic_load(&fail)) {
    for (int d = 0; d < ndev; d++) if (ctx[d].err[0]) fprintf(stderr, "fmd_node_bench: %s\n", ctx[d].err);
    return 1;
  }

  unsigned long long mine[MAXDEV][NCNT], all[MAXDEV * NCNT];
  for (int d = 0; d < ndev; d++) {
    mine[d][0] = ctx[d].resident_samples; mine[d][1] = (unsigned long long)(ctx[d].resident_s * 1e9); mine[d][2] = ctx[d].resident_pcm;
    mine[d][3] = ctx[d].h2d_samples;      mine[d][4] = (unsigned long long)(ctx[d].h2d_s * 1e9);      mine[d][5] = ctx[d].h2d_pcm;
  }
  const char *gathered = "host copy (-n)";
  char gerr[200] = {0};
  if (rccl) {
    if (gather_counters(ndev, devs, (const unsigned long long (*)[NCNT])mine, all, gerr, sizeof(gerr))) { fprintf(stderr, "fmd_node_bench: %s\n", gerr); return 1; }
    for (int d = 0; d < ndev; d++)
      for (int k = 0; k < NCNT; k++)
        if (all[d * NCNT + k] != mine[d][k]) { fprintf(stderr, "fmd_node_bench: gathered counter %d of device %d differs from the device thread's own\n", k, d); return 1; }
    gathered = "ncclAllGather (librccl, one communicator per device, this process)";
  } else {
    memcpy(all, mine, sizeof(unsigned long long) * (size_t)ndev * NCNT);
  }

  /* the line, from the gathered copy: node rate = all samples / the slowest device's time */
  unsigned long long rs = 0, rmax = 0, hs = 0, hmax = 0;
  for (int d = 0; d < ndev; d++) {
    rs += all[d * NCNT + 0]; if (all[d * NCNT + 1] > rmax) rmax = all[d * NCNT + 1];
    hs += all[d * NCNT + 3]; if (all[d * NCNT + 4] > hmax) hmax = all[d * NCNT + 4];
  }
  printf("{\"metric\": \"IQ Msamples/s through full_demod\", \"unit\": \"Msamples/s\", \"n_gpus\": %d, \"streams\": %d, \"blocks_per_launch\": %d, \"launches\": %d, \"warmup\": %d, "
         "\"value\": %.1f, \"h2d_value\": %.1f, \"h2d_pcie_gbs\": %.2f, \"scaling\": \"weak\", \"mode\": %d, \"rate_in\": %d, \"rate_out2\": %d, \"per_device\": [",
         ndev, total, B, K, W, rmax ? (double)rs / ((double)rmax * 1e-9) / 1e6 : 0.0, hmax ? (double)hs / ((double)hmax * 1e-9) / 1e6 : 0.0,
         hmax ? 2.0 * (double)hs / ((double)hmax * 1e-9) / 1e9 : 0.0, mode, rate_in, rate_out2);
  for (int d = 0; d < ndev; d++) {
    const unsigned long long *a = &all[d * NCNT];
    printf("%s{\"device\": %d, \"numa_node\": %d, \"cpus_bound\": %d, \"first_stream\": %d, \"streams\": %d, \"math_run\": %d, \"value\": %.1f, \"ms_per_launch\": %.4f, "
           "\"last_kernel_ms\": %.4f, \"pcm_values\": %llu, \"h2d_value\": %.1f, \"h2d_pcie_gbs\": %.2f, \"h2d_pcm_values\": %llu}",
           d ? ", " : "", d, ctx[d].numa, ctx[d].cpus_bound, ctx[d].first_stream, ctx[d].n_streams, ctx[d].math_run,
           a[1] ? (double)a[0] / ((double)a[1] * 1e-9) / 1e6 : 0.0, (double)a[1] * 1e-6 / K, ctx[d].kernel_ms, a[2],
           a[4] ? (double)a[3] / ((double)a[4] * 1e-9) / 1e6 : 0.0, a[4] ? 2.0 * (double)a[3] / ((double)a[4] * 1e-9) / 1e9 : 0.0, a[5]);
  }
  printf("], \"counters_gathered_by\": \"%s\", \"data_path_collectives\": 0, "
         "\"path\": \"C, one process: per device a demod thread bound to the device's NUMA node; leg 1 IQ resident in HBM, all devices start together, "
         "node rate = all samples / slowest device; leg 2 feeder pthreads -> fmd_ingest_callback -> pinned rings -> fmd_batch_pump_begin/_end\"}\n", gathered);
  return 0;
}
