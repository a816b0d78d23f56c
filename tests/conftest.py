import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lcg40():
    """40 blocks of the survey's LCG byte stream (seed 12345), SURVEY.md section 8c."""
    from oracle import lcg_bytes
    buf, _ = lcg_bytes(40 * 262144, 12345)
    return buf


@pytest.fixture(params=["valu", "mfma", "mfma_c", "mfma_d", "mfma_e", "mfma_f"])
def fast_math(request):
    """The +-1 LSB kernel families of the library: vector ALU only, stage A on the matrix pipe (what MATH_FAST resolves to for
    mono / NFM / generic filter sizes), and stages A + C on the matrix pipe (what it resolves to for 90-tap stereo with whole tiles;
    other configurations run the stage-A family under that name), and stages A + C + D (the default for 90-tap stereo at
    rate_out >= 4 rate_out2 up to round 5's first half), and the same with the L+R chain as one composite filter (mfma_e: today's default there;
    mono and everything else run what mfma_d runs under that name), and that with the second stage at the emit instants only (mfma_f: round 6's
    default where sixteen frames are a whole number of samples - 300 k, 240 k, 192 k -> 48 k; the rest runs mfma_e's kernels under that name)."""
    import rtl_fm_player_amd as R
    return {"valu": R.MATH_FAST_VALU, "mfma": R.MATH_FAST_MFMA, "mfma_c": R.MATH_FAST_MFMA_C, "mfma_d": R.MATH_FAST_MFMA_D, "mfma_e": R.MATH_FAST_MFMA_E, "mfma_f": R.MATH_FAST_MFMA_F}[request.param]
