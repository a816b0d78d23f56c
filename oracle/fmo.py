"""ctypes binding of oracle/libfm_oracle.so (test infrastructure, see fm_oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfm_oracle.so")
HASH_INIT = 1469598103934665603


class FmoConfig(C.Structure):
    _fields_ = [
        ("rate_in", C.c_int32),
        ("rate_out", C.c_int32),
        ("rate_out2", C.c_int32),
        ("mode", C.c_int32),
        ("size", C.c_int32),
        ("deemph", C.c_int32),
        ("offset_tuning", C.c_int32),
        ("deemph_lambda", C.c_float),
        ("volume", C.c_float),
    ]


class FmoState(C.Structure):
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


class _Trace(C.Structure):
    _fields_ = [("y", C.c_void_p), ("v", C.c_void_p), ("mpx", C.c_void_p)]


class _Dds(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("ph_l", "ph_r", "ph_pilot", "ph_carrier", "noise", "step_l", "step_r", "step_pilot")] + \
               [("dev_q", C.c_int32), ("amp", C.c_int32), ("stereo", C.c_int32)]


def build_oracle(force=False):
    """Compile oracle/libfm_oracle.so with the committed Makefile (gcc)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("fm_oracle.c", "fm_oracle.h"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        subprocess.run(["make", "-C", _HERE, "-B", "libfm_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(_LIB_PATH)
        L.fmo_open.restype = C.c_void_p
        L.fmo_open.argtypes = [C.POINTER(FmoConfig)]
        L.fmo_close.argtypes = [C.c_void_p]
        L.fmo_block.restype = C.c_int
        L.fmo_block.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.fmo_block_trace.restype = C.c_int
        L.fmo_block_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(_Trace)]
        L.fmo_run.restype = C.c_long
        L.fmo_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.fmo_get_state.argtypes = [C.c_void_p, C.POINTER(FmoState)]
        L.fmo_set_state.argtypes = [C.c_void_p, C.POINTER(FmoState)]
        L.fmo_get_taps.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.POINTER(C.c_float)] * 2
        L.fmo_deemph_lambda.restype = C.c_float
        L.fmo_deemph_lambda.argtypes = [C.c_int, C.c_double]
        L.fmo_lcg_fill.argtypes = [C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t]
        L.fmo_hash16.restype = C.c_uint64
        L.fmo_hash16.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t]
        L.fmo_dds_init.argtypes = [C.POINTER(_Dds)] + [C.c_int] * 5 + [C.c_uint32]
        L.fmo_dds_fill.argtypes = [C.POINTER(_Dds), C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def deemph_lambda(output_rate, tau=50e-6):
    return float(lib().fmo_deemph_lambda(int(output_rate), float(tau)))


def lcg_bytes(n, seed=12345):
    """n bytes of the survey's LCG stream; returns (uint8 array, next state)."""
    st = C.c_uint32(seed)
    buf = np.empty(n, dtype=np.uint8)
    lib().fmo_lcg_fill(C.byref(st), buf.ctypes.data, n)
    return buf, st.value


def dds_bytes(n_bytes, fs=2400000, f_left=1000, f_right=3000, amp=100, stereo=1, seed=1):
    """Integer-DDS stereo FM multiplex as u8 IQ (identical on every platform)."""
    d = _Dds()
    lib().fmo_dds_init(C.byref(d), fs, f_left, f_right, amp, stereo, seed)
    buf = np.empty(n_bytes, dtype=np.uint8)
    lib().fmo_dds_fill(C.byref(d), buf.ctypes.data, n_bytes)
    return buf


def hash16(pcm, h=HASH_INIT):
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    return int(lib().fmo_hash16(h, pcm.ctypes.data, pcm.size))


class OracleStream:
    """One demodulator stream of the CPU oracle."""

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
        self.cfg = FmoConfig(rate_in, rate_out, rate_out2, mode, size, int(bool(deemph)),
                             int(bool(offset_tuning)), deemph_lambda_, volume)
        self._h = lib().fmo_open(C.byref(self.cfg))
        if not self._h:
            raise ValueError("fmo_open rejected the configuration")

    def close(self):
        if self._h:
            lib().fmo_close(self._h)
            self._h = None

    __del__ = close

    @property
    def channels(self):
        return 2 if self.cfg.mode == 2 else 1

    def block(self, iq, trace=False):
        """Demodulate one block; returns int16 PCM (and a dict of intermediates)."""
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        n_y = iq.size // 16
        pcm = np.empty(max(n_y, 4), dtype=np.int16)
        if not trace:
            n = lib().fmo_block(self._h, iq.ctypes.data, iq.size, pcm.ctypes.data)
            if n < 0:
                raise ValueError("fmo_block error %d" % n)
            return pcm[:n].copy()
        y = np.empty(2 * n_y, dtype=np.float32)
        v = np.empty(n_y, dtype=np.float32)
        mpx = np.empty(max(n_y, 4), dtype=np.float32)
        tr = _Trace(y.ctypes.data, v.ctypes.data, mpx.ctypes.data)
        n = lib().fmo_block_trace(self._h, iq.ctypes.data, iq.size, pcm.ctypes.data, C.byref(tr))
        if n < 0:
            raise ValueError("fmo_block_trace error %d" % n)
        return pcm[:n].copy(), {"y": y, "v": v, "mpx": mpx[:n].copy()}

    def run(self, iq, block_len):
        """Demodulate consecutive blocks; returns (pcm concatenated, lens)."""
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        nb = iq.size // block_len
        pcm = np.empty(nb * (block_len // 16) + 4, dtype=np.int16)
        lens = np.empty(nb, dtype=np.int32)
        tot = lib().fmo_run(self._h, iq.ctypes.data, block_len, nb, pcm.ctypes.data, lens.ctypes.data)
        if tot < 0:
            raise ValueError("fmo_run error %d" % tot)
        return pcm[:tot].copy(), lens

    def taps(self):
        half = self.cfg.size // 2
        fb = np.empty(16, np.float32)
        fm = np.empty(half, np.float32)
        fp = np.empty(half, np.float32)
        fs = np.empty(half, np.float32)
        swf, cwf = C.c_float(), C.c_float()
        lib().fmo_get_taps(self._h, fb.ctypes.data, fm.ctypes.data, fp.ctypes.data, fs.ctypes.data,
                           C.byref(swf), C.byref(cwf))
        return {"fb": fb, "fm": fm, "fp": fp, "fs": fs, "swf": swf.value, "cwf": cwf.value}

    def get_state(self):
        st = FmoState()
        lib().fmo_get_state(self._h, C.byref(st))
        return st

    def set_state(self, st):
        lib().fmo_set_state(self._h, C.byref(st))
