"""The committed reference outputs (tests/golden/ref_vectors.npz, made from the reference
compiled here: tests/golden/make_ref_fixtures.py) against

  * the CPU oracle (runs everywhere, also where /root/reference does not exist), and
  * the HIP path through the C ABI, DIRECTLY, without the oracle in between (-m gpu):
    exact kernels bit for bit, fast kernels within 1 LSB (BASELINE.json north_star).
"""
import numpy as np
import pytest

from golden_util import BL, DDS_CONFIGS, LCG_CONFIGS, bits, check_run, vectors


@pytest.mark.parametrize("name", sorted(LCG_CONFIGS))
def test_oracle_reproduces_reference_vectors(name, lcg40):
    from oracle import OracleStream, hash16
    g = vectors()
    cfg = LCG_CONFIGS[name]
    pcm, lens = OracleStream(**cfg).run(lcg40, BL)
    check_run(name, pcm, lens, hash16)
    s = OracleStream(**cfg)
    for b in range(3):
        p, tr = s.block(lcg40[b * BL:(b + 1) * BL], trace=True)
        if b in (0, 2):
            assert np.array_equal(bits(tr["y"][:64]), bits(g[name + "/y_head%d" % b]))
            assert np.array_equal(bits(tr["y"][-64:]), bits(g[name + "/y_tail%d" % b]))
            assert np.array_equal(bits(tr["v"][:64]), bits(g[name + "/v_head%d" % b]))
            assert np.array_equal(bits(tr["v"][-64:]), bits(g[name + "/v_tail%d" % b]))
            assert np.array_equal(bits(tr["mpx"][:96]), bits(g[name + "/mpx_head%d" % b]))
    t = s.taps()
    for k in ("fb", "fm", "fp", "fs"):
        assert np.array_equal(bits(t[k]), bits(g[name + "/" + k])), k
    assert np.array_equal(bits([t["swf"], t["cwf"], s.cfg.deemph_lambda]), bits(g[name + "/scalars"]))
    s40 = OracleStream(**cfg)
    s40.run(lcg40, BL)
    st = s40.get_state()
    n = st.size
    assert st.acc == int(g[name + "/state_acc"][0])
    assert np.array_equal(bits(list(st.tb)), bits(g[name + "/state_tb"]))
    assert np.array_equal(bits([st.pre_r, st.pre_j]), bits(g[name + "/state_f"][:2]))
    assert np.array_equal(bits([st.deemph_l, st.deemph_r]), bits(g[name + "/state_f"][3:]))
    assert np.array_equal(bits(list(st.br)[:n]), bits(g[name + "/state_br"]))
    if cfg["mode"] == 2:
        assert np.array_equal(bits([st.pp]), bits(g[name + "/state_f"][2:3]))
        assert np.array_equal(bits(list(st.bm)[:n]), bits(g[name + "/state_bm"]))
        assert np.array_equal(bits(list(st.bs)[:n]), bits(g[name + "/state_bs"]))


@pytest.mark.parametrize("name", sorted(DDS_CONFIGS))
def test_oracle_reproduces_reference_vectors_fm_broadcast(name):
    from oracle import OracleStream, dds_bytes, hash16
    iq = dds_bytes(10 * BL, fs=2400000)
    assert hash16(iq[:4096].view(np.int16)) == int(vectors()["dds/iq_hash_first4k"][0])   # same bytes on every box
    pcm, lens = OracleStream(**DDS_CONFIGS[name]).run(iq, BL)
    check_run(name, pcm, lens, hash16)


# ---------------------------------------------------------------- HIP path vs the vectors

def _gpu_run(R, cfg_kw, iq, nb, math):
    b = R.BatchDemod(R.wbfm_config(block_len=BL, math=math, **cfg_kw), 1)
    out, lens = b.run_host_concat(np.ascontiguousarray(iq).reshape(1, nb, BL), nb)
    return out[0], lens[0], b


@pytest.fixture(scope="module")
def R():
    import rtl_fm_player_amd as R
    if R.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests need a real MI355X")
    return R


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(LCG_CONFIGS))
def test_hip_path_reproduces_reference_vectors(R, name, lcg40, fast_math):
    from oracle import hash16            # only the hash function (an FNV loop), not the demodulator
    g = vectors()
    cfg = LCG_CONFIGS[name]
    pcm, lens, b = _gpu_run(R, cfg, lcg40, 40, R.MATH_EXACT)
    check_run(name, pcm, lens, hash16, tol=0)
    st = b.get_state(0)
    n = cfg.get("size", 128 if cfg["mode"] == 1 else 90)
    assert st.acc == int(g[name + "/state_acc"][0])
    assert np.array_equal(bits(list(st.tb)), bits(g[name + "/state_tb"]))
    assert np.array_equal(bits([st.pre_r, st.pre_j]), bits(g[name + "/state_f"][:2]))
    assert np.array_equal(bits([st.deemph_l, st.deemph_r]), bits(g[name + "/state_f"][3:]))
    assert np.array_equal(bits(list(st.br)[:n]), bits(g[name + "/state_br"]))
    if cfg["mode"] == 2:
        assert np.array_equal(bits(list(st.bm)[:n]), bits(g[name + "/state_bm"]))
        assert np.array_equal(bits(list(st.bs)[:n]), bits(g[name + "/state_bs"]))
    b.close()
    pcm, lens, b = _gpu_run(R, cfg, lcg40, 40, fast_math)
    check_run(name, pcm, lens, hash16, tol=1)
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(DDS_CONFIGS))
def test_hip_path_reproduces_reference_vectors_fm_broadcast(R, name):
    from oracle import dds_bytes, hash16
    iq = dds_bytes(10 * BL, fs=2400000)
    for math, tol in [(R.MATH_EXACT, 0)] + [(m, 1) for m in R.FAST_MATHS]:
        pcm, lens, b = _gpu_run(R, DDS_CONFIGS[name], iq, 10, math)
        check_run(name, pcm, lens, hash16, tol=tol)
        b.close()


@pytest.mark.gpu
def test_hip_stage_taps_reproduce_reference_vectors(R, lcg40):
    """Decimated IQ / discriminator / resampler excerpts of blocks 0 and 2 (the Q1 block)."""
    import torch
    g = vectors()
    name, nb, M = "stereo_300k", 3, BL // 16
    b = R.BatchDemod(R.wbfm_config(math=R.MATH_EXACT, **LCG_CONFIGS[name]), 1)
    dev = torch.device("cuda:0")
    iq = torch.from_numpy(lcg40[: nb * BL].copy()).to(dev)
    pcm = torch.zeros(nb * b.pcm_stride, dtype=torch.int16, device=dev)
    lens = torch.zeros(nb, dtype=torch.int32, device=dev)
    y = torch.zeros(nb * 2 * M, dtype=torch.float32, device=dev)
    v = torch.zeros(nb * M, dtype=torch.float32, device=dev)
    mpx = torch.zeros(nb * M, dtype=torch.float32, device=dev)
    b.run_device(iq, nb, pcm, lens, debug={"y": y, "v": v, "mpx": mpx})
    b.sync()
    torch.cuda.synchronize()
    y, v, mpx = y.cpu().numpy(), v.cpu().numpy(), mpx.cpu().numpy()
    for k in (0, 2):
        yk, vk, mk = y[k * 2 * M:(k + 1) * 2 * M], v[k * M:(k + 1) * M], mpx[k * M:(k + 1) * M]
        assert np.array_equal(bits(yk[:64]), bits(g[name + "/y_head%d" % k]))
        assert np.array_equal(bits(yk[-64:]), bits(g[name + "/y_tail%d" % k]))
        assert np.array_equal(bits(vk[:64]), bits(g[name + "/v_head%d" % k]))
        assert np.array_equal(bits(vk[-64:]), bits(g[name + "/v_tail%d" % k]))
        assert np.array_equal(bits(mk[:96]), bits(g[name + "/mpx_head%d" % k]))
