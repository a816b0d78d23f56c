"""The +-1 LSB families beside a neighbour wave that issues double-rate MFMAs on every SIMD.

Round 3 measured wrong PCM here (every stream of every launch under v_mfma_f32_16x16x32_bf16, profiles/archive/r03m_*);
round 4 found the one instruction form behind it and took it out of the kernels (profiles/archive/r04_pk_opsel_hazard.md,
tools/isa_lint.py).  This is the short form of tools/diag/coburst.py: 256 streams fed the same IQ, launches made while
tools/diag/coburst.hip keeps one MFMA-only wave per SIMD busy on its own stream; every stream of every launch must equal
the PCM of a launch made alone, and that PCM is held against the oracle (the reference's arithmetic,
src/rtl_fm_player.c:758-788) within 1 LSB."""
import ctypes
import os
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BL = 262144


@pytest.fixture(scope="module")
def R():
    import rtl_fm_player_amd as R
    if R.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests need a real MI355X")
    return R


@pytest.fixture(scope="module")
def neighbour():
    so = os.path.join(ROOT, "tools", "diag", "libcoburst.so")
    if not os.path.exists(so):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                        os.path.join(ROOT, "tools", "diag", "coburst.hip")], check=True)
    return ctypes.CDLL(so)


@pytest.mark.parametrize("kind", [0, 1], ids=["bf16_16x16x32", "i8_16x16x64"])
@pytest.mark.parametrize("mode", [2, 1], ids=["stereo", "mono"])
@pytest.mark.parametrize("family", ["valu", "mfma", "mfma_f"])
def test_fast_family_beside_mfma_neighbour(R, neighbour, family, mode, kind):
    import torch
    from oracle import OracleStream, lcg_bytes
    S, B, NL = 256, 16, 12
    math = {"valu": R.MATH_FAST_VALU, "mfma": R.MATH_FAST_MFMA, "mfma_f": R.MATH_FAST_MFMA_F}[family]
    kw = dict(rate_in=300000, rate_out2=48000, mode=mode)
    dev = torch.device("cuda:0")
    host = lcg_bytes(B * BL, 2024)[0]
    iq = torch.from_numpy(host).to(dev).view(1, B * BL).expand(S, B * BL).contiguous()
    b = R.BatchDemod(R.wbfm_config(math=math, **kw), S)
    pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
    lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    b.reset(); b.run_device(iq, B, pcm, lens); b.sync()
    alone = pcm.view(S, -1)[0].clone()
    l0 = lens[0].cpu().numpy()
    want, wl = OracleStream(**kw).run(host, BL)
    assert np.array_equal(l0, wl)
    got = np.concatenate([pcm[0, k, :wl[k]].cpu().numpy() for k in range(B)])
    assert int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max()) <= 1
    assert neighbour.coburst_start(kind, 256, 0) == 0
    try:
        time.sleep(0.05)
        bad = 0
        for _ in range(NL):
            b.reset(); b.run_device(iq, B, pcm, lens); b.sync()
            bad += int((pcm.view(S, -1) != alone.unsqueeze(0)).any(dim=1).sum().item())
    finally:
        assert neighbour.coburst_stop() == 0
    b.close()
    assert bad == 0, "%d of %d stream-launches deviate beside the neighbour (family %d, mode %d)" % (bad, NL * S, math, mode)
