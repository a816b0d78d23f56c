"""ctypes binding of oracle/_ref/libref.so - the REFERENCE's hot path compiled here by
oracle/build_ref.py (test infrastructure only; see oracle/ref_shim.c).

RefStream mirrors OracleStream (oracle/fmo.py) so the pin tests read as
"reference vs restatement on the same call".  have_ref() is False on a checkout where
neither /root/reference nor a previously built oracle/_ref/libref.so exists; the tests
that need it skip with that reason.
"""
import ctypes as C
import os

import numpy as np

from .build_ref import build_ref, ref_paths
from .fmo import deemph_lambda

_lib = None
_ring = None


class RefState(C.Structure):
    _fields_ = [
        ("tb", C.c_float * 48),
        ("pre_r", C.c_float),
        ("pre_j", C.c_float),
        ("pp", C.c_float),
        ("deemph_l", C.c_float),
        ("deemph_r", C.c_float),
        ("acc", C.c_int32),
        ("pos", C.c_int32),
        ("size", C.c_int32),
        ("br", C.c_float * 256),
        ("bm", C.c_float * 256),
        ("bs", C.c_float * 256),
    ]


def have_ref():
    try:
        return bool(build_ref(quiet=True)) and os.path.isfile(ref_paths()[0])
    except SystemExit:
        return False


def lib():
    global _lib
    if _lib is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libref.so is not built (no /root/reference here and no earlier build)")
        L = C.CDLL(ref_paths()[0])
        L.ref_open.restype = C.c_void_p
        L.ref_open.argtypes = [C.c_int] * 6 + [C.c_float, C.c_float, C.c_int]
        L.ref_close.argtypes = [C.c_void_p]
        L.ref_block.restype = C.c_int
        L.ref_block.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.ref_block_staged.restype = C.c_int
        L.ref_block_staged.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 4
        L.ref_run.restype = C.c_long
        L.ref_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_get_state.argtypes = [C.c_void_p, C.POINTER(RefState)]
        L.ref_get_taps.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.POINTER(C.c_float)] * 2
        L.ref_get_u8_table.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_sizeof_demod_state.restype = C.c_size_t
        L.ref_offsetof_demod_state.restype = C.c_size_t
        L.ref_offsetof_demod_state.argtypes = [C.c_int]
        _lib = L
    return _lib


def ring_lib():
    global _ring
    if _ring is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libref_ring.so is not built")
        L = C.CDLL(ref_paths()[1])
        L.refring_push.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.refring_pop.restype = C.c_uint32
        L.refring_pop.argtypes = [C.c_void_p]
        L.refring_counters.argtypes = [C.POINTER(C.c_uint32)] * 4
        _ring = L
    return _ring


class RefStream:
    """One demodulator stream of the reference itself (same keywords as OracleStream)."""

    def __init__(self, rate_in=300000, rate_out=None, rate_out2=48000, mode=2, size=None,
                 deemph=True, deemph_lambda_=None, volume=0.4, offset_tuning=False,
                 output_rate=None, tau=50e-6):
        if size is None:
            size = 128 if mode == 1 else 90
        if rate_out is None:
            rate_out = rate_in
        if output_rate is None:
            output_rate = rate_out2 if rate_out2 > 0 else rate_out
        if deemph_lambda_ is None:
            deemph_lambda_ = deemph_lambda(output_rate, tau)
        self.mode, self.size = mode, size
        self._h = lib().ref_open(rate_in, rate_out, rate_out2, mode, size, int(bool(deemph)),
                                 deemph_lambda_, volume, int(bool(offset_tuning)))
        if not self._h:
            raise MemoryError("ref_open")

    def close(self):
        if self._h:
            lib().ref_close(self._h)
            self._h = None

    __del__ = close

    def block(self, iq, trace=False):
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        n_y = iq.size // 16
        pcm = np.empty(max(n_y, 4), dtype=np.int16)
        if not trace:
            n = lib().ref_block(self._h, iq.ctypes.data, iq.size, pcm.ctypes.data)
            if n < 0:
                raise ValueError("ref_block error %d" % n)
            return pcm[:n].copy()
        y = np.empty(2 * n_y, dtype=np.float32)
        v = np.empty(n_y, dtype=np.float32)
        mpx = np.empty(max(n_y, 4), dtype=np.float32)
        n = lib().ref_block_staged(self._h, iq.ctypes.data, iq.size, pcm.ctypes.data, y.ctypes.data,
                                   v.ctypes.data, mpx.ctypes.data)
        if n < 0:
            raise ValueError("ref_block_staged error %d" % n)
        return pcm[:n].copy(), {"y": y, "v": v, "mpx": mpx[:n].copy()}

    def run(self, iq, block_len):
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        nb = iq.size // block_len
        pcm = np.empty(nb * (block_len // 16) + 4, dtype=np.int16)
        lens = np.empty(nb, dtype=np.int32)
        tot = lib().ref_run(self._h, iq.ctypes.data, block_len, nb, pcm.ctypes.data, lens.ctypes.data)
        if tot < 0:
            raise ValueError("ref_run error %d" % tot)
        return pcm[:tot].copy(), lens

    def taps(self):
        half = self.size // 2
        fb = np.empty(16, np.float32)
        fm = np.empty(half, np.float32)
        fp = np.empty(half, np.float32)
        fs = np.empty(half, np.float32)
        swf, cwf = C.c_float(), C.c_float()
        lib().ref_get_taps(self._h, fb.ctypes.data, fm.ctypes.data, fp.ctypes.data, fs.ctypes.data,
                           C.byref(swf), C.byref(cwf))
        return {"fb": fb, "fm": fm, "fp": fp, "fs": fs, "swf": swf.value, "cwf": cwf.value}

    def get_state(self):
        """Carried state with the rings unrolled to oldest -> newest (the oracle's order)."""
        st = RefState()
        lib().ref_get_state(self._h, C.byref(st))
        n, pos = st.size, st.pos
        out = {"tb": np.array(st.tb, np.float32), "pre_r": st.pre_r, "pre_j": st.pre_j, "pp": st.pp,
               "deemph_l": st.deemph_l, "deemph_r": st.deemph_r, "acc": st.acc, "pos": pos, "size": n}
        for name in ("br", "bm", "bs"):
            ring = np.array(getattr(st, name), np.float32)[:n]
            out[name] = np.roll(ring, -pos)       # ring[pos] is the oldest sample
        return out
