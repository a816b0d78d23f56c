"""The ingest ring against the REFERENCE's own callback (row f3 of SURVEY.md section 8).

oracle/_ref/libref_ring.so holds rtlsdr_callback (src/rtl_fm_player.c:790-837) compiled from the
reference where it lies, with the demod thread's dequeue (:863-876) restated in
oracle/ref_ring_shim.c.  fmd_ingest_callback in FMD_OVERFLOW_REFERENCE mode must deliver the same
bytes in the same order for any interleaving of transfers and dequeues - including the two quirks of
the reference: a transfer that does not fit before the end of the ring restarts at offset 0, and an
overflow clamps the byte count without moving the read position.  The default mode (drop-oldest) is
checked for what the header promises instead.  Runs without a GPU (unbound rings).
"""
import ctypes as C

import numpy as np
import pytest

import rtl_fm_player_amd as R
from oracle import refbind

BL = 262144
RING = 16 * BL


def _ring(mode, cap=0):
    L = R.lib()
    h = C.c_void_p()
    assert L.fmd_ingest_create(C.byref(h), None, 0, cap) == 0
    assert L.fmd_ingest_set_overflow(h, mode) == 0
    return L, h


@pytest.mark.skipif(not refbind.have_ref(), reason="oracle/_ref/libref_ring.so not built (no /root/reference)")
@pytest.mark.parametrize("seed,lens", [
    (1, [BL]),                                   # the program's own cadence: 262144-byte transfers
    (2, [BL // 2, BL, BL // 4]),                 # other multiples (rtlsdr_read_async buf_len is a multiple of 512)
    (3, [512 * 37, 512 * 211, BL, 512]),         # lengths the ring size is NOT a multiple of: restart-at-zero quirk
])
def test_reference_overflow_mode_equals_the_reference_callback(seed, lens):
    Lr = refbind.ring_lib()
    Lr.refring_reset()
    L, h = _ring(1)
    rng = np.random.default_rng(seed)
    out_r = np.empty(BL, np.uint8)
    out_g = np.empty(BL, np.uint8)
    pushes = pops = overflows = 0
    for step in range(600):
        # phases: producer faster than consumer (overflow), then the consumer catches up
        p_push = 0.8 if (step // 100) % 2 == 0 else 0.3
        if rng.random() < p_push:
            n = int(rng.choice(lens))
            buf = rng.integers(0, 256, n, dtype=np.uint8)
            # BUFFER_DUMP after a retune (:1109).  Only on transfers that hold it: the reference writes its
            # 4096 mute bytes without looking at len (:805-810), this library stops at the end of the buffer
            mute = 4096 if (rng.random() < 0.05 and n >= 4096) else 0
            a, b = buf.copy(), buf.copy()
            Lr.refring_push(a.ctypes.data, n, mute)
            if mute:
                L.fmd_ingest_mute(h, mute)
            L.fmd_ingest_callback(b.ctypes.data, n, h)
            assert np.array_equal(a, b)                                   # the mute fill lands in the caller's buffer too
            pushes += 1
        else:
            nr = Lr.refring_pop(out_r.ctypes.data)
            ng = L.fmd_ingest_pop(h, out_g.ctypes.data, BL)
            assert nr == ng, (step, nr, ng)
            if nr:
                assert np.array_equal(out_r, out_g), "block differs at step %d" % step
                pops += 1
        c = [C.c_uint32() for _ in range(4)]
        Lr.refring_counters(*[C.byref(x) for x in c])
        assert L.fmd_ingest_buffered(h) == c[2].value, step              # _input_buffer_size
        overflows += c[2].value == c[3].value
    assert pushes > 100 and pops > 50 and overflows > 10                  # the run did overflow
    L.fmd_ingest_destroy(h)


def test_drop_oldest_mode_keeps_the_newest_bytes_in_order():
    L, h = _ring(0, cap=8 * 4096)
    data = (np.arange(40 * 4096) % 251).astype(np.uint8)
    for k in range(20):                                                   # 20 x 6000 bytes into a 32768-byte ring
        chunk = np.ascontiguousarray(data[k * 6000:(k + 1) * 6000])
        L.fmd_ingest_callback(chunk.ctypes.data, 6000, h)
    assert L.fmd_ingest_buffered(h) == 8 * 4096
    assert L.fmd_ingest_dropped(h) == 20 * 6000 - 8 * 4096
    out = np.empty(8 * 4096, np.uint8)
    assert L.fmd_ingest_pop(h, out.ctypes.data, out.size) == out.size
    assert np.array_equal(out, data[20 * 6000 - 8 * 4096:20 * 6000])      # exactly the newest, in order
    assert L.fmd_ingest_buffered(h) == 0 and L.fmd_ingest_pop(h, out.ctypes.data, 16) == 0
    L.fmd_ingest_destroy(h)


def test_callback_and_pop_from_two_threads():
    """The reference's threading (callback on the USB event thread, dequeue on the demod thread,
    src/rtl_fm_player.c:839-876): no byte lost, duplicated or reordered while the ring never overflows."""
    import threading
    L, h = _ring(0)
    total = 200 * BL // 4
    src = np.random.default_rng(7).integers(0, 256, total, dtype=np.uint8)
    got = []

    import time
    deadline = time.time() + 120

    def producer():
        pos = 0
        while pos < total and time.time() < deadline:
            n = min(BL // 4, total - pos)
            while L.fmd_ingest_buffered(h) + n > RING and time.time() < deadline:
                pass                                                      # this test is about ordering, not overflow
            L.fmd_ingest_callback(src[pos:pos + n].ctypes.data, n, h)
            pos += n

    t = threading.Thread(target=producer)
    t.start()
    out = np.empty(BL // 4, np.uint8)
    n_got = 0
    while n_got < total:
        assert time.time() < deadline, "ring made no progress"
        if L.fmd_ingest_pop(h, out.ctypes.data, out.size):
            got.append(out.copy())
            n_got += out.size
    t.join()
    assert np.array_equal(np.concatenate(got), src)
    assert L.fmd_ingest_dropped(h) == 0
    L.fmd_ingest_destroy(h)
