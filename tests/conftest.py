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


@pytest.fixture(params=["valu", "mfma", "mfma_f"])
def fast_math(request):
    """The +-1 LSB kernel families of the library: vector ALU only; stage A on the matrix pipe (what MATH_FAST resolves to for generic filter sizes, ragged
    tiles and rates the decimating second stage does not cover); and every stage that has a matrix form (mfma_f: 90-tap stereo and 128-tap mono where
    sixteen frames are a whole number of samples - 300 k, 240 k, 192 k -> 48 k, 25 k -> 12.5 k; other configurations run the stage-A family under that name)."""
    import rtl_fm_player_amd as R
    return {"valu": R.MATH_FAST_VALU, "mfma": R.MATH_FAST_MFMA, "mfma_f": R.MATH_FAST_MFMA_F}[request.param]
