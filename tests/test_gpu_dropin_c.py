"""A plain C program through the C ABI, as rtl_fm_player.c itself would call the library
(tests/c/dropin_check.c: demod_init -> init_* -> per block rotate_90_u8_f32 + full_demod), and the
reference's threading on the ingest side (callback on one thread, pump on another)."""
import ctypes as C
import os
import subprocess
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BL = 262144


@pytest.fixture(scope="module")
def R():
    import rtl_fm_player_amd as R
    if R.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests need a real MI355X")
    return R


def _exe(R):
    return os.path.join(os.path.dirname(R.library_path()), "dropin_check")


def test_c_caller_reproduces_the_reference_hash(R):
    """40 blocks of the survey's LCG stream through demod_init / init_* / rotate_90_u8_f32 / full_demod
    called from C: 209714 PCM values hashing to c3e7eda4bd16dfe1 (reproduced by
    the reference compiled here: tests/test_ref_pin.py)."""
    env = {k: v for k, v in os.environ.items() if k != "FMD_MATH_FAST"}
    r = subprocess.run([_exe(R)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr
    n, h = r.stdout.split()
    assert int(n) == 209714 and h == "c3e7eda4bd16dfe1"


def test_c_caller_fast_kernels_via_environment(R):
    """FMD_MATH_FAST=1 selects the +-1 LSB kernels for the same C program: same count, another hash."""
    r = subprocess.run([_exe(R), "10"], capture_output=True, text=True, env=dict(os.environ, FMD_MATH_FAST="1"),
                       timeout=600)
    assert r.returncode == 0, r.stderr
    n, h = r.stdout.split()
    env = {k: v for k, v in os.environ.items() if k != "FMD_MATH_FAST"}
    r2 = subprocess.run([_exe(R), "10"], capture_output=True, text=True, env=env, timeout=600)
    n2, h2 = r2.stdout.split()
    assert n == n2 and h != h2


@pytest.mark.parametrize("overflow_mode", [0, 1])
def test_callback_thread_while_main_thread_pumps(R, overflow_mode):
    """The reference's threading (src/rtl_fm_player.c:839-876): fmd_ingest_callback runs on producer
    threads at the 262144-byte cadence of rtlsdr_read_async while the main thread pumps with two jobs in
    flight; rings sized so that nothing overflows: PCM == oracle for every stream."""
    from oracle import OracleStream, lcg_bytes
    L = R.lib()
    ns, nb_total, per_job = 4, 24, 4
    kw = dict(rate_in=300000, rate_out2=48000, mode=2)
    b = R.BatchDemod(R.wbfm_config(math=R.MATH_EXACT, **kw), ns)
    iqs = [lcg_bytes(nb_total * BL, 9000 + s)[0] for s in range(ns)]
    rings = []
    for s in range(ns):
        h = C.c_void_p()
        assert L.fmd_ingest_create(C.byref(h), b._h, s, 0) == 0            # 16 blocks, like _input_buffer
        assert L.fmd_ingest_set_overflow(h, overflow_mode) == 0
        rings.append(h)

    deadline = time.time() + 120

    def producer(s):
        for k in range(nb_total):
            # leave room: this test must not overflow.  fmd_ingest_buffered does not count the bytes of the (up
            # to two) jobs in flight, 2 x per_job blocks, which still occupy the ring: 3 + 1 + 8 <= 16 blocks
            while L.fmd_ingest_buffered(rings[s]) > 3 * BL and time.time() < deadline:
                time.sleep(0.0005)
            blk = np.ascontiguousarray(iqs[s][k * BL:(k + 1) * BL])
            L.fmd_ingest_callback(blk.ctypes.data, BL, rings[s])

    thr = [threading.Thread(target=producer, args=(s,)) for s in range(ns)]
    [t.start() for t in thr]
    outs, lens_all, done, in_flight = [[] for _ in range(ns)], [[] for _ in range(ns)], 0, []
    while done < nb_total:
        assert time.time() < deadline, "pump made no progress (%d of %d blocks)" % (done, nb_total)
        n = L.fmd_batch_pump_begin(b._h, per_job) if len(in_flight) < 2 else 0
        assert n >= 0, L.fmd_last_error()
        if n > 0:
            in_flight.append(n)
        if in_flight and (n == 0 or len(in_flight) == 2):
            nb = in_flight.pop(0)
            pcm = np.zeros((ns, nb, b.pcm_stride), dtype=np.int16)
            lens = np.zeros((ns, nb), dtype=np.int32)
            assert L.fmd_batch_pump_end(b._h, pcm.ctypes.data, lens.ctypes.data) == nb
            for s in range(ns):
                outs[s] += [pcm[s, k, :lens[s, k]] for k in range(nb)]
                lens_all[s] += list(lens[s])
            done += nb
    [t.join() for t in thr]
    for s in range(ns):
        want, wl = OracleStream(**kw).run(iqs[s], BL)
        assert lens_all[s] == list(wl)
        assert np.array_equal(np.concatenate(outs[s]), want), "stream %d" % s
        assert L.fmd_ingest_dropped(rings[s]) == 0 and L.fmd_ingest_buffered(rings[s]) == 0
    b.close()                                # batch first, rings after: the rings are detached, not freed
    for h in rings:
        L.fmd_ingest_destroy(h)
