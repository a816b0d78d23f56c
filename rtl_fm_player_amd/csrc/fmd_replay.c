/*
 * fmd_replay.c - file-replay driver: the role of the reference's demod thread
 * (demod_thread_fn, src/rtl_fm_player.c:855-933) for recorded IQ.
 *
 * Reads rtl_sdr-style capture files (interleaved u8 I,Q: exactly what
 * rtlsdr_read_async hands to the callback) in blocks of MAXIMUM_BUF_LENGTH =
 * 262144 bytes, demodulates them on the GPU through the C ABI of
 * libfmdemod_mi355x.so and writes int16 PCM (raw, or WAV in the reference's
 * format).  Several input files are demodulated as independent streams in one
 * batch.  Like the reference's thread it only ever processes whole blocks: a
 * trailing partial block is left unread.
 *
 *   fmd_replay [options] in.u8 [in2.u8 ...]
 *     -s rate_in      demod rate (capture rate / 8), default 240000  (-s of the reference)
 *     -r rate_out2    output rate, default 48000                      (-r)
 *     -X / -Y         stereo / mono at 192 k like the reference's flags (before -s/-r to override)
 *     -M mode         lpr.mode 0/1/2 (default 2)
 *     -E              offset tuning: no fs/4 rotation
 *     -D              disable de-emphasis;  -t tau_us (default 50)
 *     -v volume       default 0.4
 *     -e              bit-exact kernels (default: fast, +-1 LSB)
 *     -n blocks       blocks per launch (default 16)
 *     -w              write WAV (reference header) instead of raw PCM
 *     -o prefix       output prefix: <prefix><stream>.pcm|.wav (default "out")
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define FMD_NO_REFERENCE_TYPES
#include "fmdemod_mi355x.h"

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char **argv) {
  fmd_config cfg = {240000, 240000, 48000, 2, 90, 1, 0, 0.f, 0.4f, FMD_MAXIMUM_BUF_LENGTH, FMD_MATH_FAST};
  double tau = 50e-6;
  int per_launch = 16, wav = 0, opt;
  const char *prefix = "out";
  while ((opt = getopt(argc, argv, "s:r:XYM:EDt:v:en:wo:h")) != -1) {
    switch (opt) {
      case 's': cfg.rate_in = cfg.rate_out = atoi(optarg); break;          /* :1412-1415 */
      case 'r': cfg.rate_out2 = atoi(optarg); break;                       /* :1416-1419 */
      case 'X': cfg.rate_in = cfg.rate_out = 192000; cfg.rate_out2 = 48000; cfg.mode = 2; cfg.size = 90; break;
      case 'Y': cfg.rate_in = cfg.rate_out = 192000; cfg.rate_out2 = 48000; cfg.mode = 1; cfg.size = 128; break;
      case 'M': cfg.mode = atoi(optarg); cfg.size = (cfg.mode == 1) ? 128 : 90; break;
      case 'E': cfg.offset_tuning = 1; break;
      case 'D': cfg.deemph = 0; break;
      case 't': tau = atof(optarg) * 1e-6; break;
      case 'v': cfg.volume = (float)atof(optarg); break;
      case 'e': cfg.math = FMD_MATH_EXACT; break;
      case 'n': per_launch = atoi(optarg); break;
      case 'w': wav = 1; break;
      case 'o': prefix = optarg; break;
      default:
        fprintf(stderr, "usage: fmd_replay [-s rate_in] [-r rate_out2] [-X|-Y] [-M mode] [-E] [-D] [-t tau_us]\n"
                        "                  [-v volume] [-e] [-n blocks] [-w] [-o prefix] in.u8 [in2.u8 ...]\n");
        return opt == 'h' ? 0 : 2;
    }
  }
  const int n = argc - optind;
  if (n < 1 || per_launch < 1) { fprintf(stderr, "fmd_replay: no input files\n"); return 2; }
  const int out_rate = cfg.rate_out2 > 0 ? cfg.rate_out2 : cfg.rate_out;
  cfg.deemph_lambda = fmd_deemph_lambda(out_rate, tau);                    /* :1577 */

  fmd_batch *b = NULL;
  if (fmd_batch_create(&b, &cfg, NULL, n, -1)) {
    fprintf(stderr, "fmd_replay: %s\n", fmd_last_error());
    return 1;
  }
  const int stride = fmd_batch_pcm_stride(b);
  const size_t bl = (size_t)cfg.block_len;
  FILE **in = (FILE **)calloc((size_t)n, sizeof(FILE *));
  FILE **raw = (FILE **)calloc((size_t)n, sizeof(FILE *));
  fmd_wav **wv = (fmd_wav **)calloc((size_t)n, sizeof(fmd_wav *));
  uint8_t *iq = (uint8_t *)malloc((size_t)n * per_launch * bl);
  int16_t *pcm = (int16_t *)malloc((size_t)n * per_launch * stride * sizeof(int16_t));
  int32_t *lens = (int32_t *)malloc((size_t)n * per_launch * sizeof(int32_t));
  if (!in || !raw || !wv || !iq || !pcm || !lens) { fprintf(stderr, "fmd_replay: out of memory\n"); return 1; }
  for (int s = 0; s < n; s++) {
    char name[1024];
    in[s] = fopen(argv[optind + s], "rb");
    if (!in[s]) { perror(argv[optind + s]); return 1; }
    snprintf(name, sizeof(name), "%s%d.%s", prefix, s, wav ? "wav" : "pcm");
    if (wav) {
      if (fmd_wav_open(&wv[s], name, cfg.mode)) { fprintf(stderr, "fmd_replay: cannot open %s\n", name); return 1; }
    } else if (!(raw[s] = fopen(name, "wb"))) { perror(name); return 1; }
  }

  unsigned long long blocks_done = 0, pcm_done = 0;
  const double t0 = now_s();
  double t_gpu = 0;
  for (;;) {
    /* how many whole blocks can every stream deliver this round? */
    int nb = per_launch;
    for (int s = 0; s < n && nb > 0; s++) {
      uint8_t *dst = iq + (size_t)s * per_launch * bl;
      int got = 0;
      while (got < nb && fread(dst + (size_t)got * bl, 1, bl, in[s]) == bl) got++;
      if (got < nb) nb = got;             /* streams stay block-aligned with each other */
    }
    if (nb == 0) break;
    if (nb < per_launch) {                /* compact the per-stream slabs to nb blocks each */
      for (int s = 1; s < n; s++) memmove(iq + (size_t)s * nb * bl, iq + (size_t)s * per_launch * bl, (size_t)nb * bl);
    }
    const double g0 = now_s();
    if (fmd_batch_run_host(b, iq, nb, pcm, lens)) {
      fprintf(stderr, "fmd_replay: %s\n", fmd_last_error());
      return 1;
    }
    t_gpu += now_s() - g0;
    for (int s = 0; s < n; s++)
      for (int k = 0; k < nb; k++) {
        const int16_t *p = pcm + ((size_t)s * nb + k) * stride;
        const int len = lens[(size_t)s * nb + k];
        if (wav) fmd_wav_write(wv[s], p, (size_t)len);
        else fwrite(p, sizeof(int16_t), (size_t)len, raw[s]);           /* result_len << 1 bytes, :905-908 */
        pcm_done += (unsigned long long)len;
      }
    blocks_done += (unsigned long long)nb * n;
    if (nb < per_launch) break;
  }
  const double dt = now_s() - t0;
  for (int s = 0; s < n; s++) {
    fclose(in[s]);
    if (wav) fmd_wav_close(wv[s]);
    else fclose(raw[s]);
  }
  fprintf(stderr, "fmd_replay: %d stream(s), %llu blocks, %llu PCM values, %.3f s wall (%.3f s in fmd_batch_run_host), "
                  "%.1f M IQ samples/s including file I/O and PCIe\n",
          n, blocks_done, pcm_done, dt, t_gpu, blocks_done * (bl / 2) / dt / 1e6);
  fmd_batch_destroy(b);
  free(in); free(raw); free(wv); free(iq); free(pcm); free(lens);
  return 0;
}
