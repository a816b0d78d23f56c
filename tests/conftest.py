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
