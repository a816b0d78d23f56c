/*
 * fmd_wav.c - WAV output compatible with the reference's InitWaveOut /
 * CloseWaveOut (src/rtl_fm_player.c:1259-1328).
 *
 * The reference writes a fixed 260-byte header (include/rtl_fm_player.h:216-253):
 * a canonical 44-byte PCM header (16 bit, 48000 Hz, 1 or 2 channels) followed by
 * 216 zero bytes that play as silence, and on close patches the RIFF size at
 * offset 4 (file size - 8) and the data size at offset 40 (file size - 44).  The
 * header always says 48000 Hz, whatever the output rate is.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FMD_NO_REFERENCE_TYPES
#include "fmdemod_mi355x.h"

struct fmd_wav {
  FILE *f;
  int is_stdout;
};

static void put_le32(unsigned char *p, uint32_t v) {
  p[0] = (unsigned char)(v & 0xff);
  p[1] = (unsigned char)((v >> 8) & 0xff);
  p[2] = (unsigned char)((v >> 16) & 0xff);
  p[3] = (unsigned char)((v >> 24) & 0xff);
}
static void put_le16(unsigned char *p, uint16_t v) {
  p[0] = (unsigned char)(v & 0xff);
  p[1] = (unsigned char)((v >> 8) & 0xff);
}

int fmd_wav_header(int mode, unsigned char out[FMD_WAV_HEADER_BYTES]) {
  const int channels = (mode == 2) ? 2 : 1;
  const uint32_t rate = 48000, byte_rate = rate * (uint32_t)channels * 2;
  memset(out, 0, FMD_WAV_HEADER_BYTES);
  memcpy(out, "RIFF", 4);
  put_le32(out + 4, byte_rate + 36);          /* placeholder: one second of audio, as in the reference */
  memcpy(out + 8, "WAVEfmt ", 8);
  put_le32(out + 16, 16);                     /* fmt chunk size */
  put_le16(out + 20, 1);                      /* PCM */
  put_le16(out + 22, (uint16_t)channels);
  put_le32(out + 24, rate);
  put_le32(out + 28, byte_rate);
  put_le16(out + 32, (uint16_t)(channels * 2));
  put_le16(out + 34, 16);
  memcpy(out + 36, "data", 4);
  put_le32(out + 40, byte_rate);              /* placeholder */
  return FMD_OK;
}

int fmd_wav_open(fmd_wav **out, const char *path, int mode) {
  if (!out || !path) return FMD_E_ARG;
  *out = NULL;
  fmd_wav *w = (fmd_wav *)calloc(1, sizeof(*w));
  if (!w) return FMD_E_NOMEM;
  if (strcmp(path, "-") == 0) {
    w->f = stdout;
    w->is_stdout = 1;
  } else {
    w->f = fopen(path, "wb");
    if (!w->f) { free(w); return FMD_E_ARG; }
  }
  unsigned char hdr[FMD_WAV_HEADER_BYTES];
  fmd_wav_header(mode, hdr);
  if (fwrite(hdr, 1, sizeof(hdr), w->f) != sizeof(hdr)) {
    if (!w->is_stdout) fclose(w->f);
    free(w);
    return FMD_E_STATE;
  }
  *out = w;
  return FMD_OK;
}

int fmd_wav_write(fmd_wav *w, const int16_t *pcm, size_t n_values) {
  if (!w || (!pcm && n_values)) return FMD_E_ARG;
  return fwrite(pcm, sizeof(int16_t), n_values, w->f) == n_values ? FMD_OK : FMD_E_STATE;
}

int fmd_wav_close(fmd_wav *w) {
  if (!w) return FMD_OK;
  int rc = FMD_OK;
  if (!w->is_stdout) {                        /* src/rtl_fm_player.c:1265-1279 */
    long size = ftell(w->f);
    unsigned char le[4];
    if (size < 0 || fseek(w->f, 4, SEEK_SET)) rc = FMD_E_STATE;
    else {
      put_le32(le, (uint32_t)(size - 8));
      fwrite(le, 1, 4, w->f);
      fseek(w->f, 40, SEEK_SET);
      put_le32(le, (uint32_t)(size - 44));
      fwrite(le, 1, 4, w->f);
    }
    fclose(w->f);
  } else {
    fflush(w->f);
  }
  free(w);
  return rc;
}
