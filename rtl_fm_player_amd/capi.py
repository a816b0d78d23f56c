"""ctypes binding of libfmdemod_mi355x.so (see include/fmdemod_mi355x.h)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("FMD_LIB_PATH") or os.path.join(_HERE, "libfmdemod_mi355x.so")   # override: kernel experiments

MATH_EXACT = 0
MATH_FAST = 1          # the fastest +-1 LSB family of this build for the configuration
MATH_FAST_VALU = 2     # +-1 LSB, vector ALU only
MATH_FAST_MFMA = 3     # +-1 LSB, /8 decimator on the matrix pipe
MATH_FAST_MFMA_F = 7   # ... and every other stage that has a matrix form: 90-tap stereo (pilot / L-R filters at full rate, the composite L+R filter and the second
                       # stage at the emit instants), 128-tap mono (stage D at the emit instants); what MATH_FAST resolves to where it applies
MATH_FAST_MFMA_C, MATH_FAST_MFMA_D, MATH_FAST_MFMA_E = 4, 5, 6   # retired in round 6: accepted, mean MATH_FAST
FAST_MATHS = (MATH_FAST_VALU, MATH_FAST_MFMA, MATH_FAST_MFMA_F)
MAXIMUM_BUF_LENGTH = 16 * 16384


class FmdError(RuntimeError):
    pass


def library_path():
    return _LIB


def build_library(force=False):
    """Compile the HIP kernels and the C host layer for gfx950 (hipcc + gcc)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc]
    if force:
        cmd.append("-B")
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    return _LIB


class FmdConfig(C.Structure):
    _fields_ = [
        ("rate_in", C.c_int32), ("rate_out", C.c_int32), ("rate_out2", C.c_int32),
        ("mode", C.c_int32), ("size", C.c_int32), ("deemph", C.c_int32),
        ("offset_tuning", C.c_int32), ("deemph_lambda", C.c_float), ("volume", C.c_float),
        ("block_len", C.c_int32), ("math", C.c_int32),
    ]


class FmdTaps(C.Structure):
    _fields_ = [("fb", C.c_float * 16), ("fm", C.c_float * 128), ("fp", C.c_float * 128),
                ("fs", C.c_float * 128), ("swf", C.c_float), ("cwf", C.c_float)]


class _FmdFilterError(C.Structure):
    _fields_ = [("taps", C.c_int32), ("qf", C.c_int32), ("rms_lsb", C.c_float), ("worst_lsb", C.c_float),
                ("worst_samples_lsb", C.c_float), ("worst_taps_lsb", C.c_float), ("worst_dropped_lsb", C.c_float)]


class FmdErrorEstimate(C.Structure):
    _fields_ = [("family", C.c_int32), ("filters", C.c_int32), ("f", _FmdFilterError * 2), ("limit_rms_lsb", C.c_float)]


class FmdStreamState(C.Structure):
    _fields_ = [("tb", C.c_float * 48), ("pre_r", C.c_float), ("pre_j", C.c_float), ("pp", C.c_float),
                ("deemph_l", C.c_float), ("deemph_r", C.c_float), ("acc", C.c_int32),
                ("reserved", C.c_int32 * 2), ("br", C.c_float * 256), ("bm", C.c_float * 256),
                ("bs", C.c_float * 256)]


class FmdDebugTaps(C.Structure):
    _fields_ = [("y", C.c_void_p), ("v", C.c_void_p), ("mpx", C.c_void_p), ("prof", C.c_void_p)]


class LpReal(C.Structure):          # struct lp_real
    _fields_ = [("br", C.POINTER(C.c_float)), ("bm", C.POINTER(C.c_float)), ("bs", C.POINTER(C.c_float)),
                ("fm", C.POINTER(C.c_float)), ("fp", C.POINTER(C.c_float)), ("fs", C.POINTER(C.c_float)),
                ("swf", C.c_float), ("cwf", C.c_float), ("pp", C.c_float), ("pos", C.c_int),
                ("size", C.c_int), ("rsize", C.c_int), ("mode", C.c_int)]


class DemodState(C.Structure):      # struct demod_state (x86-64 glibc layout)
    _fields_ = [
        ("exit_flag", C.c_int), ("thread", C.c_ulong),
        ("buf", C.c_uint8 * MAXIMUM_BUF_LENGTH), ("buf_len", C.c_uint32),
        ("lowpassed", C.c_int16 * (MAXIMUM_BUF_LENGTH << 1)), ("lp_len", C.c_int),
        ("lowpass_tb", C.c_float * 48), ("lp_i_hist", C.c_int16 * 60), ("lp_q_hist", C.c_int16 * 60),
        ("result", C.c_int16 * MAXIMUM_BUF_LENGTH), ("result_len", C.c_int),
        ("droop_i_hist", C.c_int16 * 9), ("droop_q_hist", C.c_int16 * 9),
        ("offset_tuning", C.c_int), ("rate_in", C.c_int), ("rate_out", C.c_int), ("rate_out2", C.c_int),
        ("now_r", C.c_int), ("now_j", C.c_int), ("pre_r", C.c_int), ("pre_j", C.c_int),
        ("pre_r_f32", C.c_float), ("pre_j_f32", C.c_float), ("prev_index", C.c_int),
        ("downsample", C.c_int), ("post_downsample", C.c_int), ("output_scale", C.c_int),
        ("squelch_level", C.c_int), ("conseq_squelch", C.c_int), ("squelch_hits", C.c_int),
        ("terminate_on_squelch", C.c_int), ("downsample_passes", C.c_int), ("comp_fir_size", C.c_int),
        ("custom_atan", C.c_int), ("deemph", C.c_double), ("deemph_a", C.c_int), ("deemph_l", C.c_int),
        ("deemph_r", C.c_int), ("deemph_l_f32", C.c_float), ("deemph_r_f32", C.c_float),
        ("deemph_lambda", C.c_float), ("volume", C.c_float), ("now_lpr", C.c_int),
        ("prev_lpr_index", C.c_int), ("lpr", LpReal),
        ("rw", C.c_uint8 * 56), ("ready", C.c_uint8 * 48), ("ready_m", C.c_uint8 * 40),
        ("output_target", C.c_void_p),
    ]


_lib = None

_EXPORTS = [
    "init_u8_f32_table", "init_lp_f32", "init_lp_real_f32", "deinit_lp_real_f32", "demod_init",
    "rotate_90_u8_f32", "u8_f32", "full_demod", "fmd_demod_release", "fmd_dropin_set_math", "fmd_dropin_set_error_handler",
    "fmd_config_error_estimate", "fmd_design_taps", "fmd_deemph_lambda", "fmd_batch_create", "fmd_batch_destroy",
    "fmd_batch_pcm_stride", "fmd_batch_n_streams", "fmd_batch_math", "fmd_config_family", "fmd_batch_set_time_split", "fmd_batch_run_device", "fmd_batch_run_device_debug",
    "fmd_batch_sync", "fmd_batch_wait_stream", "fmd_batch_run_host", "fmd_batch_get_state", "fmd_batch_set_state",
    "fmd_batch_reset", "fmd_batch_last_kernel_ms", "fmd_batch_set_timing", "fmd_batch_kernel_name", "fmd_last_error",
    "fmd_device_count", "fmd_ingest_create", "fmd_ingest_destroy", "fmd_ingest_callback",
    "fmd_ingest_buffered", "fmd_ingest_dropped", "fmd_ingest_mute", "fmd_ingest_set_overflow", "fmd_ingest_pop",
    "fmd_batch_pump",
    "fmd_batch_pump_begin", "fmd_batch_pump_end",
    "fmd_wav_header", "fmd_wav_open", "fmd_wav_write", "fmd_wav_close",
]

INGEST_CB = C.CFUNCTYPE(None, C.POINTER(C.c_ubyte), C.c_uint32, C.c_void_p)
DROPIN_ERROR_CB = C.CFUNCTYPE(None, C.c_char_p, C.c_char_p, C.c_void_p)     # fmd_dropin_error_fn


def exported_symbols():
    return list(_EXPORTS)


def lib():
    """Load libfmdemod_mi355x.so; raises FmdError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise FmdError("%s is missing: run __graft_entry__.build() (there is no CPU fallback)" % _LIB)
    # torch bundles its own libamdhip64.so.7; two HIP runtimes in one process break the
    # second one, so when torch is installed let it load first and share its copy.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(_LIB)
    vp = C.c_void_p
    L.fmd_design_taps.argtypes = [C.POINTER(FmdConfig), C.POINTER(FmdTaps)]
    L.fmd_deemph_lambda.restype = C.c_float
    L.fmd_deemph_lambda.argtypes = [C.c_int, C.c_double]
    L.fmd_batch_create.argtypes = [C.POINTER(vp), C.POINTER(FmdConfig), C.POINTER(FmdTaps), C.c_int, C.c_int]
    L.fmd_batch_destroy.argtypes = [vp]
    L.fmd_batch_destroy.restype = None
    L.fmd_batch_pcm_stride.argtypes = [vp]
    L.fmd_batch_n_streams.argtypes = [vp]
    L.fmd_batch_math.argtypes = [vp]
    L.fmd_config_family.argtypes = [C.POINTER(FmdConfig), vp]
    L.fmd_batch_set_time_split.argtypes = [vp, C.c_int]
    L.fmd_batch_run_device.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.fmd_batch_run_device_debug.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.POINTER(FmdDebugTaps)]
    L.fmd_batch_sync.argtypes = [vp]
    L.fmd_batch_run_host.argtypes = [vp, vp, C.c_int, vp, vp]
    L.fmd_batch_get_state.argtypes = [vp, C.c_int, C.POINTER(FmdStreamState)]
    L.fmd_batch_set_state.argtypes = [vp, C.c_int, C.POINTER(FmdStreamState)]
    L.fmd_batch_reset.argtypes = [vp]
    L.fmd_batch_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.fmd_batch_set_timing.argtypes = [vp, C.c_int]
    L.fmd_batch_wait_stream.argtypes = [vp, vp]
    L.fmd_batch_kernel_name.argtypes = [vp]
    L.fmd_batch_kernel_name.restype = C.c_char_p
    L.fmd_last_error.restype = C.c_char_p
    L.fmd_ingest_create.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_uint32]
    L.fmd_ingest_destroy.argtypes = [vp]
    L.fmd_ingest_destroy.restype = None
    L.fmd_ingest_callback.argtypes = [vp, C.c_uint32, vp]
    L.fmd_ingest_callback.restype = None
    L.fmd_ingest_buffered.argtypes = [vp]
    L.fmd_ingest_buffered.restype = C.c_uint32
    L.fmd_ingest_dropped.argtypes = [vp]
    L.fmd_ingest_dropped.restype = C.c_uint64
    L.fmd_ingest_mute.argtypes = [vp, C.c_int]
    L.fmd_ingest_mute.restype = None
    L.fmd_ingest_set_overflow.argtypes = [vp, C.c_int]
    L.fmd_ingest_pop.argtypes = [vp, vp, C.c_uint32]
    L.fmd_ingest_pop.restype = C.c_uint32
    L.fmd_batch_pump.argtypes = [vp, C.c_int, vp, vp]
    L.fmd_batch_pump_begin.argtypes = [vp, C.c_int]
    L.fmd_batch_pump_end.argtypes = [vp, vp, vp]
    for name in ("init_lp_real_f32", "deinit_lp_real_f32", "demod_init", "rotate_90_u8_f32", "u8_f32",
                 "full_demod", "fmd_demod_release"):
        getattr(L, name).argtypes = [C.POINTER(DemodState)]
        getattr(L, name).restype = None
    _lib = L
    return L


def _check(rc, what):
    if rc < 0:
        raise FmdError("%s failed (%d): %s" % (what, rc, lib().fmd_last_error().decode()))
    return rc


def device_count():
    return int(lib().fmd_device_count())


def wbfm_config(rate_in=300000, rate_out=None, rate_out2=48000, mode=2, size=None, deemph=True,
                deemph_lambda=None, volume=0.4, offset_tuning=False, block_len=262144,
                math=MATH_EXACT, output_rate=None, tau=50e-6):
    """fmd_config with the reference's defaults (demod_init / -X / -Y, src/rtl_fm_player.c:1156-1195)."""
    if size is None:
        size = 128 if mode == 1 else 90
    if rate_out is None:
        rate_out = rate_in
    if output_rate is None:
        output_rate = rate_out2 if rate_out2 > 0 else rate_out
    if deemph_lambda is None:
        deemph_lambda = float(lib().fmd_deemph_lambda(int(output_rate), float(tau)))
    return FmdConfig(rate_in, rate_out, rate_out2, mode, size, int(bool(deemph)), int(bool(offset_tuning)),
                     deemph_lambda, volume, block_len, math)


def design_taps(cfg):
    t = FmdTaps()
    _check(lib().fmd_design_taps(C.byref(cfg), C.byref(t)), "fmd_design_taps")
    return t


def config_family(cfg, taps=None):
    """The kernel family fmd_batch_create would run for this configuration (needs no device); raises FmdError for one it refuses."""
    rc = lib().fmd_config_family(C.byref(cfg), C.byref(taps) if taps is not None else None)
    _check(rc if rc < 0 else 0, "fmd_config_family")
    return rc


def config_error_estimate(cfg, taps=None):
    """fmd_config_error_estimate as a dict: what the fixed-point second stage adds to a PCM value (rms estimate, worst-case bound and its terms, per
    filter) and the family the configuration resolves to.  Needs no device."""
    e = FmdErrorEstimate()
    _check(lib().fmd_config_error_estimate(C.byref(cfg), C.byref(taps) if taps is not None else None, C.byref(e)), "fmd_config_error_estimate")
    return {"family": e.family, "limit_rms_lsb": e.limit_rms_lsb,
            "filters": [{k: getattr(e.f[i], k) for k, _ in _FmdFilterError._fields_} for i in range(e.filters)]}


def _ptr(x):
    """Device pointer of a torch tensor, or a raw integer address."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


class BatchDemod:
    """n_streams independent demodulators; thin wrapper of the fmd_batch_* C API."""

    def __init__(self, cfg, n_streams, taps=None, device=-1):
        self.cfg = cfg
        self.n_streams = int(n_streams)
        self._h = C.c_void_p()
        _check(lib().fmd_batch_create(C.byref(self._h), C.byref(cfg), C.byref(taps) if taps else None,
                                      self.n_streams, device), "fmd_batch_create")
        self.pcm_stride = lib().fmd_batch_pcm_stride(self._h)
        self.channels = 2 if cfg.mode == 2 else 1
        self.math = lib().fmd_batch_math(self._h)      # the kernel family MATH_FAST resolved to

    def close(self):
        if getattr(self, "_h", None):
            lib().fmd_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def run_device(self, d_iq, n_blocks, d_pcm, d_lens, hip_stream=None, debug=None):
        """Asynchronous device-resident run (tensors or raw device addresses) on `hip_stream` (None: the batch's own stream - the
        buffers must be ready there: torch.cuda.synchronize(), or wait_stream(), after producing them on torch's stream)."""
        if debug is None:
            rc = lib().fmd_batch_run_device(self._h, _ptr(d_iq), n_blocks, _ptr(d_pcm), _ptr(d_lens),
                                            _ptr(hip_stream))
        else:
            dbg = FmdDebugTaps(*[(_ptr(debug.get(k)).value if debug.get(k) is not None else None)
                                 for k in ("y", "v", "mpx", "prof")])
            rc = lib().fmd_batch_run_device_debug(self._h, _ptr(d_iq), n_blocks, _ptr(d_pcm), _ptr(d_lens),
                                                  _ptr(hip_stream), C.byref(dbg))
        _check(rc, "fmd_batch_run_device")

    def sync(self):
        _check(lib().fmd_batch_sync(self._h), "fmd_batch_sync")

    def wait_stream(self, producer_stream=None):
        """Order this batch's own stream behind what is queued on `producer_stream` (a hipStream_t address; None: torch's current
        stream).  run_device() with hip_stream=None launches on the batch's OWN stream: tensors another stream is still filling -
        or memory torch's allocator handed out in its stream's order - are only safe after this or a synchronisation."""
        if producer_stream is None:
            import torch
            producer_stream = torch.cuda.current_stream().cuda_stream
        _check(lib().fmd_batch_wait_stream(self._h, _ptr(producer_stream)), "fmd_batch_wait_stream")

    def run_host(self, iq, n_blocks):
        """iq: uint8 [n_streams, n_blocks, block_len] -> (pcm [S, B, stride] int16, lens [S, B])."""
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        assert iq.size == self.n_streams * n_blocks * self.cfg.block_len
        pcm = np.zeros((self.n_streams, n_blocks, self.pcm_stride), dtype=np.int16)
        lens = np.zeros((self.n_streams, n_blocks), dtype=np.int32)
        _check(lib().fmd_batch_run_host(self._h, iq.ctypes.data, n_blocks, pcm.ctypes.data, lens.ctypes.data),
               "fmd_batch_run_host")
        return pcm, lens

    def run_host_concat(self, iq, n_blocks):
        """Like run_host but returns, per stream, the PCM of all blocks concatenated."""
        pcm, lens = self.run_host(iq, n_blocks)
        out = [np.concatenate([pcm[s, b, :lens[s, b]] for b in range(n_blocks)]) for s in range(self.n_streams)]
        return out, lens

    def set_time_split(self, workers_per_cu):
        """Time chunks per launch: > 0 workers per CU to cut for, 0 default, < 0 never cut; see fmd_batch_set_time_split."""
        _check(lib().fmd_batch_set_time_split(self._h, int(workers_per_cu)), "fmd_batch_set_time_split")

    def set_timing(self, on):
        """Bracket every launch with an event pair (default) or not; see fmd_batch_set_timing."""
        _check(lib().fmd_batch_set_timing(self._h, int(bool(on))), "fmd_batch_set_timing")

    def last_kernel_ms(self):
        ms = C.c_float()
        _check(lib().fmd_batch_last_kernel_ms(self._h, C.byref(ms)), "fmd_batch_last_kernel_ms")
        return ms.value

    def kernel_name(self):
        return lib().fmd_batch_kernel_name(self._h).decode()

    def get_state(self, stream=0):
        st = FmdStreamState()
        _check(lib().fmd_batch_get_state(self._h, stream, C.byref(st)), "fmd_batch_get_state")
        return st

    def set_state(self, stream, st):
        _check(lib().fmd_batch_set_state(self._h, stream, C.byref(st)), "fmd_batch_set_state")

    def reset(self):
        _check(lib().fmd_batch_reset(self._h), "fmd_batch_reset")
