"""MI355X-native FM demodulation path (the IQ -> PCM hot path of rtl_fm_player).

The product is the C-ABI library ``libfmdemod_mi355x.so`` (include/fmdemod_mi355x.h):
C host code + hand-written gfx950 kernels.  This package is a thin ctypes view of
that ABI for the tests and the benchmark; it contains no arithmetic of its own
and has no CPU fallback: importing :mod:`rtl_fm_player_amd.capi` raises if the
library has not been built.
"""
from .capi import (  # noqa: F401
    MATH_EXACT,
    MATH_FAST,
    FAST_MATHS,
    MATH_FAST_MFMA,
    MATH_FAST_MFMA_C,
    MATH_FAST_MFMA_D,
    MATH_FAST_MFMA_E,
    MATH_FAST_MFMA_F,
    MATH_FAST_VALU,
    BatchDemod,
    DemodState,
    FmdConfig,
    FmdError,
    FmdStreamState,
    FmdTaps,
    build_library,
    config_error_estimate,
    config_family,
    design_taps,
    device_count,
    lib,
    library_path,
    wbfm_config,
)
