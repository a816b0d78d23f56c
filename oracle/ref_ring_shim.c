/*
 * ref_ring_shim.c - handle-free API over the REFERENCE's own ingest callback.
 *
 * TEST INFRASTRUCTURE ONLY.  Appended by oracle/build_ref.py to a translation unit made of
 * the reference's rtlsdr_callback (src/rtl_fm_player.c:790-837) and the globals it uses
 * (include/rtl_fm_player.h:56-57, :64-74, :112-125, :127-175, :210-211), read where they lie.
 * Everything in this file is this repository's code: the producer side calls the
 * reference's callback as librtlsdr would (src/librtlsdr.c:1720-1724), the consumer side
 * restates the dequeue of demod_thread_fn (src/rtl_fm_player.c:863-876).  It exists so that
 * tests/test_ring_ref.py can hold fmd_ingest_callback's reference-overflow mode against
 * the reference's real behaviour, overflow included.
 *
 * rtlsdr_cancel_async (librtlsdr, absent from this image) is only reached when _do_exit is
 * set, which this shim never does: the symbol stays weak and undefined.
 */
#pragma weak rtlsdr_cancel_async

#define REF_API __attribute__((visibility("default")))

REF_API void refring_reset(void) {
  _input_buffer_rpos = _input_buffer_wpos = _input_buffer_size = 0;
  memset(_input_buffer, 0, sizeof(_input_buffer));
  memset(&dongle, 0, sizeof(dongle));
  pthread_rwlock_init(&demod.rw, NULL);
  dongle.demod_target = &demod;       /* dongle_init, src/rtl_fm_player.c:1153 */
  (void)_output_buffer;
}

/* One completed bulk transfer: librtlsdr hands (buffer, actual_length, ctx) to the callback
 * (src/librtlsdr.c:1720-1724).  `mute` > 0 arms the retune mute first, as the controller does
 * (src/rtl_fm_player.c:1109: dongle.mute = BUFFER_DUMP).  The callback may write the mute
 * fill into buf, as the reference's does. */
REF_API void refring_push(unsigned char *buf, uint32_t len, int mute) {
  if (mute > 0) dongle.mute = mute;
  rtlsdr_callback(buf, len, &dongle);
}

/* demod_thread_fn :863-876: one MAXIMUM_BUF_LENGTH block if buffered; returns its length or 0. */
REF_API uint32_t refring_pop(uint8_t *out) {
  const uint32_t len = MAXIMUM_BUF_LENGTH;
  if (_input_buffer_size < len) return 0;
  pthread_rwlock_wrlock(&demod.rw);
  memcpy(out, _input_buffer + _input_buffer_rpos, len);
  _input_buffer_rpos += len;
  _input_buffer_size -= len;
  if (_input_buffer_rpos == _input_buffer_size_max) _input_buffer_rpos = 0;
  pthread_rwlock_unlock(&demod.rw);
  return len;
}

REF_API void refring_counters(uint32_t *rpos, uint32_t *wpos, uint32_t *size, uint32_t *size_max) {
  *rpos = _input_buffer_rpos;
  *wpos = _input_buffer_wpos;
  *size = _input_buffer_size;
  *size_max = _input_buffer_size_max;
}
