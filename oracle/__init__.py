"""CPU oracle for the rtl_fm_player IQ->PCM path: TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product package (rtl_fm_player_amd) never does.

oracle/_ref/ (built by build_ref.py from /root/reference where it lies, git-ignored) holds the
reference's own hot path as a shared library; refbind.py binds it for the pin tests, the fixture
generator and bench.py's cpu_baseline leg.
"""
from .fmo import (  # noqa: F401
    FmoConfig,
    FmoState,
    OracleStream,
    build_oracle,
    dds_bytes,
    deemph_lambda,
    hash16,
    lcg_bytes,
    HASH_INIT,
)
