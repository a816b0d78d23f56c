"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Exact-math kernels must be bit-identical to the oracle (and therefore hash to
the reference's known answers); fast-math kernels must be
within +-1 LSB (BASELINE.json north_star tolerance).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BL = 262144

CONFIGS = {
    "stereo_300k": dict(rate_in=300000, rate_out2=48000, mode=2),
    "mono_300k": dict(rate_in=300000, rate_out2=48000, mode=1),
    "nfm_25k": dict(rate_in=25000, rate_out2=12500, mode=1),
    "stereo_240k": dict(rate_in=240000, rate_out2=48000, mode=2),
    "stereo_192k": dict(rate_in=192000, rate_out2=48000, mode=2),
}
KNOWN_HASH = {
    "stereo_300k": 0xC3E7EDA4BD16DFE1, "mono_300k": 0x2109FE431B558355, "nfm_25k": 0x3E6F57574F3156AA,
    "stereo_240k": 0x8E0413ED2BF00E75, "stereo_192k": 0x6E145D091E77DBC9,
}


@pytest.fixture(scope="module")
def R():
    import rtl_fm_player_amd as R
    if R.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests need a real MI355X")
    return R


def oracle_run(cfg_kw, iq, block_len=BL):
    from oracle import OracleStream
    s = OracleStream(**cfg_kw)
    pcm, lens = s.run(iq, block_len)
    return pcm, lens, s


def gpu_run(R, cfg_kw, iq, n_blocks, math, block_len=BL, launches=1, n_streams=1, time_split=0):
    cfg = R.wbfm_config(block_len=block_len, math=math, **cfg_kw)
    b = R.BatchDemod(cfg, n_streams)
    b.set_time_split(time_split)
    per = n_blocks // launches
    iq = iq.reshape(n_streams, n_blocks, block_len)
    outs = [[] for _ in range(n_streams)]
    lens_all = []
    for l in range(launches):
        part = np.ascontiguousarray(iq[:, l * per:(l + 1) * per])
        out, lens = b.run_host_concat(part, per)
        for s in range(n_streams):
            outs[s].append(out[s])
        lens_all.append(lens)
    return [np.concatenate(o) for o in outs], np.concatenate(lens_all, axis=1), b


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_exact_bit_identical_40_blocks(R, lcg40, name):
    from oracle import hash16
    want, wlens, _ = oracle_run(CONFIGS[name], lcg40)
    got, lens, _ = gpu_run(R, CONFIGS[name], lcg40, 40, R.MATH_EXACT)
    assert np.array_equal(lens[0], wlens)
    assert got[0].size == want.size
    bad = np.flatnonzero(got[0] != want)
    assert bad.size == 0, "first mismatch at %d: gpu %d oracle %d" % (bad[0], got[0][bad[0]], want[bad[0]])
    assert hash16(got[0]) == KNOWN_HASH[name]


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_fast_within_one_lsb(R, lcg40, name, fast_math):
    want, wlens, _ = oracle_run(CONFIGS[name], lcg40)
    got, lens, _ = gpu_run(R, CONFIGS[name], lcg40, 40, fast_math)
    assert np.array_equal(lens[0], wlens)
    d = np.abs(got[0].astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1, "max |diff| %d at %d" % (d.max(), int(d.argmax()))


@pytest.mark.parametrize("name", ["stereo_300k", "mono_300k"])
def test_state_carry_across_launches(R, lcg40, name):
    """40 blocks in one launch == 5 launches of 8 blocks (state handed over in HBM)."""
    one, lens1, _ = gpu_run(R, CONFIGS[name], lcg40, 40, R.MATH_EXACT, launches=1)
    five, lens5, _ = gpu_run(R, CONFIGS[name], lcg40, 40, R.MATH_EXACT, launches=5)
    assert np.array_equal(lens1, lens5)
    assert np.array_equal(one[0], five[0])


def test_stage_taps_bit_identical(R, lcg40):
    """Decimated IQ, discriminator and resampler outputs, stage by stage."""
    import torch
    from oracle import OracleStream
    nb = 3
    cfg = R.wbfm_config(math=R.MATH_EXACT, **CONFIGS["stereo_300k"])
    b = R.BatchDemod(cfg, 1)
    M = BL // 16
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
    s = OracleStream(**CONFIGS["stereo_300k"])
    for k in range(nb):
        p, tr = s.block(lcg40[k * BL:(k + 1) * BL], trace=True)
        n = p.size
        assert int(lens[k]) == n
        assert np.array_equal(y[k * 2 * M:(k + 1) * 2 * M].cpu().numpy().view(np.uint32), tr["y"].view(np.uint32))
        assert np.array_equal(v[k * M:(k + 1) * M].cpu().numpy().view(np.uint32), tr["v"].view(np.uint32))
        assert np.array_equal(mpx[k * M:k * M + n].cpu().numpy(), tr["mpx"])
        assert np.array_equal(pcm[k * b.pcm_stride:k * b.pcm_stride + n].cpu().numpy(), p)


@pytest.mark.parametrize("name", ["stereo_300k", "mono_300k", "nfm_25k"])
@pytest.mark.parametrize("math", ["exact", "fast"])
def test_debug_build_and_plain_build_give_the_same_pcm(R, lcg40, name, math):
    """Every fused kernel exists twice: the build that serves fmd_debug_taps and the one without their checks
    (launches without taps).  Same arithmetic: PCM, lengths and the carried state must be bit-identical - noise
    input, so the cold paths run in both."""
    import torch
    nb = 4
    m = R.MATH_EXACT if math == "exact" else R.MATH_FAST
    dev = torch.device("cuda:0")
    iq = torch.from_numpy(lcg40[: nb * BL].copy()).to(dev)
    out = []
    for taps in (False, True):
        b = R.BatchDemod(R.wbfm_config(math=m, **CONFIGS[name]), 1)
        M = BL // 16
        pcm = torch.zeros(nb * b.pcm_stride, dtype=torch.int16, device=dev)
        lens = torch.zeros(nb, dtype=torch.int32, device=dev)
        if taps:
            v = torch.zeros(nb * M, dtype=torch.float32, device=dev)
            b.run_device(iq, nb, pcm, lens, debug={"v": v})
        else:
            b.run_device(iq, nb, pcm, lens)
        b.sync()
        st = b.get_state(0)
        out.append((pcm.cpu().numpy(), lens.cpu().numpy(), (st.acc, st.pre_r, st.pre_j, st.pp, st.deemph_l, st.deemph_r, list(st.br)[:128])))
        b.close()
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][0], out[1][0])
    assert out[0][2] == out[1][2]


def test_wait_stream_orders_the_batch_behind_torch(R, lcg40):
    """run_device(hip_stream=None) launches on the batch's own stream.  Here the IQ bytes are still on their way when it is called - an
    asynchronous copy queued on torch's stream behind a second of matrix products - and fmd_batch_wait_stream is all that orders the
    two: without it the kernel would read the zeros the buffer held before."""
    import torch
    nb = 3
    dev = torch.device("cuda:0")
    want, wl, _ = oracle_run(CONFIGS["stereo_300k"], lcg40[: nb * BL])
    b = R.BatchDemod(R.wbfm_config(math=R.MATH_EXACT, **CONFIGS["stereo_300k"]), 1)
    host = torch.from_numpy(lcg40[: nb * BL].copy()).pin_memory()
    iq = torch.zeros(nb * BL, dtype=torch.uint8, device=dev)
    pcm = torch.zeros(nb * b.pcm_stride, dtype=torch.int16, device=dev)
    lens = torch.zeros(nb, dtype=torch.int32, device=dev)
    big = torch.randn((8192, 8192), device=dev)
    torch.cuda.synchronize()
    for _ in range(30):
        big = (big @ big) * 1e-4                     # ~a second of queued work in front of the copy
    iq.copy_(host, non_blocking=True)
    b.wait_stream()                                   # torch's current stream
    b.run_device(iq, nb, pcm, lens)
    b.sync()
    got_l = lens.cpu().numpy()
    assert np.array_equal(got_l, wl)
    got = np.concatenate([pcm[k * b.pcm_stride:k * b.pcm_stride + int(got_l[k])].cpu().numpy() for k in range(nb)])
    assert np.array_equal(got, want)
    b.close()


def test_default_family_falls_back_when_taps_do_not_fit_the_float_accumulators(R, lcg40):
    """The default stereo family reads its int32 limb-pair sums as floats (accumulators started at the bits of 1.5 x 2^23), which needs every
    weight class inside +-2^22 for ANY samples: fmd_batch_create checks the taps' limbs (build_ci_scales).  The reference's design passes;
    a caller's filter whose every tap has three large limbs (90 x 0x7F7F7F / 2^23) does not - MATH_FAST must then resolve to the
    family with stage A only, and still agree with the exact kernels on the same taps (a gain-of-90 filter clips: +-1 LSB where it does not)."""
    import torch
    cfg_kw = CONFIGS["stereo_300k"]
    b = R.BatchDemod(R.wbfm_config(math=R.MATH_FAST, **cfg_kw), 1)
    assert b.math == R.MATH_FAST_MFMA_F
    b.close()
    # the second stage on the matrix pipe needs sixteen frames to be a whole number of samples (a multiple of four: up to 100 for stereo, 32 .. 128 for mono)
    # and rate_out >= 4 rate_out2 (stereo) / 2 rate_out2 (mono); everything else runs the stage-A family
    for kw, want in ((CONFIGS["stereo_192k"], R.MATH_FAST_MFMA_F), (dict(rate_in=220000, rate_out2=48000, mode=2), R.MATH_FAST_MFMA),
                     (dict(rate_in=171000, rate_out2=44100, mode=2), R.MATH_FAST_MFMA),
                     (dict(rate_in=300000, rate_out2=48000, mode=2, size=64), R.MATH_FAST_MFMA), (CONFIGS["mono_300k"], R.MATH_FAST_MFMA_F),
                     (CONFIGS["nfm_25k"], R.MATH_FAST_MFMA_F), (dict(rate_in=96000, rate_out2=32000, mode=1), R.MATH_FAST_MFMA_F),
                     (dict(rate_in=100000, rate_out2=48000, mode=1), R.MATH_FAST_MFMA),     # (16 x 100 / 48 is no integer)
                     (dict(rate_in=48000, rate_out2=32000, mode=1), R.MATH_FAST_MFMA), (dict(rate_in=240000, rate_out2=48000, mode=1, size=90), R.MATH_FAST_MFMA)):
        for m in (R.MATH_FAST, R.MATH_FAST_MFMA_F, R.MATH_FAST_MFMA_E, R.MATH_FAST_MFMA_D):   # (the retired names mean MATH_FAST)
            b = R.BatchDemod(R.wbfm_config(math=m, **kw), 1)
            assert b.math == want, (kw, m, b.math)
            b.close()
    b = R.BatchDemod(R.wbfm_config(math=R.MATH_FAST, block_len=16 * 1000, **cfg_kw), 1)      # ragged tiles: stage A only
    assert b.math == R.MATH_FAST_MFMA
    b.close()
    taps = R.design_taps(R.wbfm_config(math=R.MATH_FAST, **cfg_kw))
    for k in range(45):
        taps.fs[k] = 8355711.0 / 2 ** 23            # 0x7F7F7F / 2^23: limbs (127, 127, 127)
    out = {}
    dev = torch.device("cuda:0")
    nb = 2
    iq = torch.from_numpy(lcg40[: nb * BL].copy()).to(dev)
    for name, m in (("fast", R.MATH_FAST), ("exact", R.MATH_EXACT)):
        b = R.BatchDemod(R.wbfm_config(math=m, volume=0.001, **cfg_kw), 1, taps=taps)
        if name == "fast":
            assert b.math == R.MATH_FAST_MFMA, "the L-R filter's limbs do not fit: stage C must not run on the matrix pipe"
        pcm = torch.zeros(nb * b.pcm_stride, dtype=torch.int16, device=dev)
        lens = torch.zeros(nb, dtype=torch.int32, device=dev)
        b.run_device(iq, nb, pcm, lens)
        b.sync()
        out[name] = (pcm.cpu().numpy().astype(np.int32), lens.cpu().numpy())
        b.close()
    assert np.array_equal(out["fast"][1], out["exact"][1])
    assert int(np.abs(out["fast"][0] - out["exact"][0]).max()) <= 1


def test_carried_state_matches_oracle(R, lcg40):
    nb = 5
    _, _, s = oracle_run(CONFIGS["stereo_300k"], lcg40[: nb * BL])
    _, _, b = gpu_run(R, CONFIGS["stereo_300k"], lcg40[: nb * BL], nb, R.MATH_EXACT)
    a, g = s.get_state(), b.get_state(0)
    assert a.acc == g.acc
    for f in ("pre_r", "pre_j", "pp", "deemph_l", "deemph_r"):
        assert getattr(a, f) == getattr(g, f), f
    assert list(a.tb) == list(g.tb)
    for f in ("br", "bm", "bs"):
        assert list(getattr(a, f))[:90] == list(getattr(g, f))[:90], f


@pytest.mark.parametrize("name", ["stereo_300k", "stereo_192k"])
def test_fast_carried_rings_match_the_oracle(R, lcg40, name, fast_math):
    """What a +-1 LSB launch leaves in the br / bm / bs rings (the drop-in surface mirrors them into the caller's struct lp_real) against the
    reference's rings: br in the reference's own bits where the hand-over recomputes it, bm and bs to a few steps of the 2^-20 grid the
    matrix-pipe stages keep them on.  FMD_MATH_FAST_MFMA_F keeps NO bm ring while it runs (its L+R chain is one composite filter): the ring
    is made at the launch's end from the discriminator history, and met again at the next launch's start (lr_head_fix) - five one-block
    launches here, so the second to fifth start from a ring the launch before made."""
    nb = 5
    _, _, s = oracle_run(CONFIGS[name], lcg40[: nb * BL])
    _, _, b = gpu_run(R, CONFIGS[name], lcg40[: nb * BL], nb, fast_math, launches=nb)
    a, g = s.get_state(), b.get_state(0)
    assert a.acc == g.acc
    for f, tol in (("br", 2e-6), ("bm", 4e-6), ("bs", 1e-4)):   # (bs: noise has no pilot - the regenerated carrier is ill-conditioned, the redo threshold is what bounds it)
        d = np.abs(np.array(list(getattr(a, f))[:90], dtype=np.float64) - np.array(list(getattr(g, f))[:90], dtype=np.float64)).max()
        assert d <= tol, (f, d)


def test_many_streams_independent(R):
    """8 streams with different inputs in one launch, each equal to its own oracle."""
    from oracle import lcg_bytes
    ns, nb = 8, 4
    iqs = [lcg_bytes(nb * BL, 12345 + s)[0] for s in range(ns)]
    got, lens, _ = gpu_run(R, CONFIGS["stereo_300k"], np.concatenate(iqs), nb, R.MATH_EXACT, n_streams=ns)
    for s in range(ns):
        want, wl, _ = oracle_run(CONFIGS["stereo_300k"], iqs[s])
        assert np.array_equal(lens[s], wl)
        assert np.array_equal(got[s], want), "stream %d" % s


@pytest.mark.parametrize("block_len", [64, 80, 1024, 16 * 2048 + 16, 16 * 2048 + 48, 50000 * 16 // 16 * 16])
@pytest.mark.parametrize("name", ["stereo_300k", "mono_300k", "nfm_25k"])
def test_ragged_block_lengths(R, lcg40, name, block_len, fast_math):
    """Block lengths that are not tile multiples, down to the 64-byte minimum."""
    nb = 6
    iq = lcg40[: nb * block_len]
    want, wlens, _ = oracle_run(CONFIGS[name], iq, block_len)
    got, lens, _ = gpu_run(R, CONFIGS[name], iq, nb, R.MATH_EXACT, block_len=block_len)
    assert np.array_equal(lens[0], wlens)
    assert np.array_equal(got[0], want)
    fast, flens, _ = gpu_run(R, CONFIGS[name], iq, nb, fast_math, block_len=block_len)
    assert np.array_equal(flens[0], wlens)
    assert np.abs(fast[0].astype(np.int32) - want.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("name", ["stereo_300k", "mono_300k"])
def test_fast_state_carried_across_launches(R, lcg40, name, fast_math):
    """Fast kernels, 12 blocks as 1, 3 and 12 launches: the carried state (including the
    de-emphasis state made by the blocked recurrence) keeps every split within 1 LSB."""
    nb = 12
    want, _, _ = oracle_run(CONFIGS[name], lcg40[: nb * BL])
    for launches in (1, 3, 12):
        got, _, _ = gpu_run(R, CONFIGS[name], lcg40[: nb * BL], nb, fast_math, launches=launches)
        assert np.abs(got[0].astype(np.int32) - want.astype(np.int32)).max() <= 1, launches


@pytest.mark.parametrize("kw", [
    dict(rate_in=300000, rate_out2=48000, mode=0),                       # drop decimator
    dict(rate_in=300000, rate_out2=0, mode=1),                           # no resampler at all
    dict(rate_in=300000, rate_out2=48000, mode=2, deemph=False),         # de-emphasis off
    dict(rate_in=300000, rate_out2=48000, mode=2, offset_tuning=True),   # u8_f32 instead of the rotation
    dict(rate_in=300000, rate_out2=48000, mode=2, size=64),              # non-default filter length
    dict(rate_in=300000, rate_out2=48000, mode=1, size=90),
    dict(rate_in=300000, rate_out2=48000, mode=2, volume=4.0),           # clipping branch
    dict(rate_in=171000, rate_out=171000, rate_out2=44100, mode=2),      # awkward ratio, Q1 pattern differs
])
def test_variant_configs(R, lcg40, kw):
    nb = 6
    want, wlens, _ = oracle_run(kw, lcg40[: nb * BL])
    got, lens, _ = gpu_run(R, kw, lcg40[: nb * BL], nb, R.MATH_EXACT)
    assert np.array_equal(lens[0], wlens)
    assert np.array_equal(got[0], want)


@pytest.mark.parametrize("kw", [
    dict(rate_in=300000, rate_out2=96000, mode=1, tau=750e-6),            # lambda = 0.986: lambda^16 = 0.80
    dict(rate_in=300000, rate_out2=48000, mode=2, tau=300e-6),            # lambda = 0.933
    dict(rate_in=300000, rate_out2=96000, mode=2, tau=75e-6),             # lambda = 0.870, 164 frames per tile
])
def test_slow_deemphasis(R, lcg40, kw, fast_math):
    """De-emphasis that decays slowly: the blocked recurrence of the fast kernels (scan over
    16-frame groups) and the chunk warm-up lengths must hold for lambda close to 1."""
    nb = 8
    want, wlens, _ = oracle_run(kw, lcg40[: nb * BL])
    got, lens, _ = gpu_run(R, kw, lcg40[: nb * BL], nb, R.MATH_EXACT)
    assert np.array_equal(lens[0], wlens)
    assert np.array_equal(got[0], want)
    fast, flens, _ = gpu_run(R, kw, lcg40[: nb * BL], nb, fast_math)
    assert np.array_equal(flens[0], wlens)
    assert np.abs(fast[0].astype(np.int32) - want.astype(np.int32)).max() <= 1


def test_synthetic_fm_stereo(R, fast_math):
    """Integer-DDS stereo multiplex: exact bit-identical, fast within 1 LSB, and audible tones."""
    from oracle import dds_bytes
    nb = 8
    iq = dds_bytes(nb * BL)
    want, _, _ = oracle_run(CONFIGS["stereo_300k"], iq)
    got, _, _ = gpu_run(R, CONFIGS["stereo_300k"], iq, nb, R.MATH_EXACT)
    assert np.array_equal(got[0], want)
    fast, _, _ = gpu_run(R, CONFIGS["stereo_300k"], iq, nb, fast_math)
    assert np.abs(fast[0].astype(np.int32) - want.astype(np.int32)).max() <= 1
    left = want[0::2].astype(np.float64)
    assert left[4000:].std() > 200          # a demodulated tone, not silence


def test_drop_in_full_demod(R, lcg40):
    """The reference-shaped calls on a layout-compatible struct demod_state."""
    from oracle import OracleStream
    from rtl_fm_player_amd.capi import DemodState
    L = R.lib()
    d = DemodState()
    L.demod_init(C.byref(d))
    d.rate_in = d.rate_out = 300000
    d.rate_out2 = 48000
    d.deemph_lambda = L.fmd_deemph_lambda(48000, 50e-6)
    L.init_u8_f32_table()
    L.init_lp_f32()
    L.init_lp_real_f32(C.byref(d))
    s = OracleStream(**CONFIGS["stereo_300k"])
    for k in range(4):
        blk = lcg40[k * BL:(k + 1) * BL]
        C.memmove(d.buf, blk.ctypes.data, BL)
        d.buf_len = BL
        L.rotate_90_u8_f32(C.byref(d))
        L.full_demod(C.byref(d))
        want = s.block(blk)
        got = np.frombuffer(d.result, dtype=np.int16, count=d.result_len)
        assert d.result_len == want.size
        assert np.array_equal(got, want), "block %d" % k
    st = s.get_state()
    assert d.prev_lpr_index == st.acc and d.lpr.pos == st.pos
    assert d.lpr.pp == st.pp and d.deemph_l_f32 == st.deemph_l and d.deemph_r_f32 == st.deemph_r
    ring = [d.lpr.br[(d.lpr.pos + i) % 90] for i in range(90)]
    assert ring == list(st.br)[:90]
    L.deinit_lp_real_f32(C.byref(d))


def test_drop_in_notices_a_caller_who_edits_the_struct(R, lcg40):
    """The drop-in keeps the carried state on the device between calls (one synchronisation per block) and skips the upload
    while the struct still holds what the last call mirrored into it.  The reference's state IS the struct, though
    (src/rtl_fm_player.c:758-788 reads and writes its fields), so a caller may reset or restore it between blocks: the next
    call must see the struct's values.  Two blocks, then the struct's state is zeroed like a fresh stream's: block 3 must
    equal the oracle's first block of a new stream on that input - and a restore of a saved state must continue where it was."""
    import copy
    from oracle import OracleStream
    from rtl_fm_player_amd.capi import DemodState
    L = R.lib()
    d = DemodState()
    L.demod_init(C.byref(d))
    d.rate_in = d.rate_out = 300000
    d.rate_out2 = 48000
    d.deemph_lambda = L.fmd_deemph_lambda(48000, 50e-6)
    L.init_u8_f32_table(); L.init_lp_f32(); L.init_lp_real_f32(C.byref(d))

    def block(k):
        blk = lcg40[k * BL:(k + 1) * BL]
        C.memmove(d.buf, blk.ctypes.data, BL)
        d.buf_len = BL
        L.rotate_90_u8_f32(C.byref(d))
        L.full_demod(C.byref(d))
        return np.frombuffer(d.result, dtype=np.int16, count=d.result_len).copy()

    s = OracleStream(**CONFIGS["stereo_300k"])
    for k in range(2):
        assert np.array_equal(block(k), s.block(lcg40[k * BL:(k + 1) * BL]))
    # save what the struct holds after block 1 (the scalars and the three rings)
    saved = dict(tb=list(d.lowpass_tb), pre=(d.pre_r_f32, d.pre_j_f32), de=(d.deemph_l_f32, d.deemph_r_f32), acc=d.prev_lpr_index,
                 pp=d.lpr.pp, pos=d.lpr.pos, br=[d.lpr.br[i] for i in range(90)], bm=[d.lpr.bm[i] for i in range(90)],
                 bs=[d.lpr.bs[i] for i in range(90)])
    # a fresh stream's state, written by the caller
    for i in range(48): d.lowpass_tb[i] = 0.0
    d.pre_r_f32 = d.pre_j_f32 = d.deemph_l_f32 = d.deemph_r_f32 = 0.0
    d.prev_lpr_index = 0
    d.lpr.pp = 0.0
    d.lpr.pos = 0
    for i in range(90): d.lpr.br[i] = d.lpr.bm[i] = d.lpr.bs[i] = 0.0
    fresh = OracleStream(**CONFIGS["stereo_300k"])
    assert np.array_equal(block(2), fresh.block(lcg40[2 * BL:3 * BL])), "the zeroed struct was not uploaded"
    # restore the saved state: block 2 again, now as the continuation of blocks 0, 1
    for i in range(48): d.lowpass_tb[i] = saved["tb"][i]
    d.pre_r_f32, d.pre_j_f32 = saved["pre"]
    d.deemph_l_f32, d.deemph_r_f32 = saved["de"]
    d.prev_lpr_index = saved["acc"]
    d.lpr.pp = saved["pp"]
    d.lpr.pos = saved["pos"]
    for i in range(90):
        d.lpr.br[i] = saved["br"][i]; d.lpr.bm[i] = saved["bm"][i]; d.lpr.bs[i] = saved["bs"][i]
    assert np.array_equal(block(2), s.block(lcg40[2 * BL:3 * BL])), "the restored struct was not uploaded"
    assert np.array_equal(block(3), s.block(lcg40[3 * BL:4 * BL]))          # and on from there without an upload
    L.deinit_lp_real_f32(C.byref(d))


def test_drop_in_keeps_the_stream_across_a_config_change(R, lcg40):
    """The reference keeps demodulating the same stream when its caller changes `volume` or `buf_len` between blocks (both are
    plain fields of struct demod_state, read per call: src/rtl_fm_player.c:758-788, :711-735).  Here both live in fmd_config, so a
    change rebuilds the single-stream batch - whose device state starts from zero: the struct's state must then be uploaded even
    though the struct still holds exactly what the previous call mirrored into it (the upload-skip shadow)."""
    from oracle import OracleStream
    from rtl_fm_player_amd.capi import DemodState
    L = R.lib()
    d = DemodState()
    L.demod_init(C.byref(d))
    d.rate_in = d.rate_out = 300000
    d.rate_out2 = 48000
    d.deemph_lambda = L.fmd_deemph_lambda(48000, 50e-6)
    L.init_u8_f32_table(); L.init_lp_f32(); L.init_lp_real_f32(C.byref(d))

    def run(iq):
        C.memmove(d.buf, iq.ctypes.data, iq.size)
        d.buf_len = iq.size
        L.rotate_90_u8_f32(C.byref(d))
        L.full_demod(C.byref(d))
        return np.frombuffer(d.result, dtype=np.int16, count=d.result_len).copy()

    s = OracleStream(**CONFIGS["stereo_300k"])
    pos = 0
    for k in range(2):
        assert np.array_equal(run(lcg40[pos:pos + BL]), s.block(lcg40[pos:pos + BL]))
        pos += BL
    # volume changes: the oracle continues the SAME stream (state carried over) with the new volume
    d.volume = 1.0
    s2 = OracleStream(**dict(CONFIGS["stereo_300k"], volume=1.0))
    s2.set_state(s.get_state())
    assert np.array_equal(run(lcg40[pos:pos + BL]), s2.block(lcg40[pos:pos + BL])), "the stream was reset by the volume change"
    pos += BL
    # buf_len changes (half a block, then a quarter, then back): same stream
    for n in (BL // 2, BL // 4, BL):
        assert np.array_equal(run(lcg40[pos:pos + n]), s2.block(lcg40[pos:pos + n])), "the stream was reset by buf_len %d" % n
        pos += n
    assert np.array_equal(run(lcg40[pos:pos + BL]), s2.block(lcg40[pos:pos + BL]))      # and on without an upload
    L.deinit_lp_real_f32(C.byref(d))


@pytest.mark.parametrize("math", ["exact", "fast"])
def test_launches_captured_in_a_hip_graph(R, math):
    """One block per launch is the reference's real-time cadence (src/rtl_fm_player.c:863-889); a caller who wants that without the per-launch
    host cost captures the launches into a hipGraph on its own stream and replays it.  The library then launches plainly (no timing events
    on the dispatch packet: a captured stream) and its state ping-pong must come out where it went in: an EVEN number of
    launches per graph.  Two one-block launches per graph, four replays, eight streams, against the oracle."""
    import torch
    from oracle import OracleStream, lcg_bytes
    S, kw = 8, CONFIGS["stereo_300k"]
    dev = torch.device("cuda:0")
    iq_all = np.stack([lcg_bytes(8 * BL, 100 + s)[0] for s in range(S)]).reshape(S, 8, BL)
    b = R.BatchDemod(R.wbfm_config(block_len=BL, math=R.MATH_EXACT if math == "exact" else R.MATH_FAST, **kw), S)
    bufs = [(torch.zeros((S, 1, BL), dtype=torch.uint8, device=dev), torch.zeros((S, 1, b.pcm_stride), dtype=torch.int16, device=dev),
             torch.zeros((S, 1), dtype=torch.int32, device=dev)) for _ in range(2)]
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st):
        for iq, pcm, lens in bufs:
            b.run_device(iq, 1, pcm, lens, hip_stream=st.cuda_stream)
    # a captured launch carries no timing events: asking for its time is a call-sequence error, not a stale number
    with pytest.raises(R.FmdError):
        b.last_kernel_ms()
    b.reset()
    out = [[] for _ in range(S)]
    for k in range(0, 8, 2):
        for j, (iq, _, _) in enumerate(bufs):
            iq.copy_(torch.from_numpy(iq_all[:, k + j:k + j + 1].copy()))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        for s in range(S):
            for _, pcm, lens in bufs:
                out[s].append(pcm[s, 0, :int(lens[s, 0])].cpu().numpy().copy())
    for s in range(S):
        want, _ = OracleStream(**kw).run(iq_all[s].reshape(-1), BL)
        got = np.concatenate(out[s])
        assert got.size == want.size
        d = int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max())
        assert d <= (0 if math == "exact" else 1), (s, d)
    # ... and a capture that would need the batch's event hand-over between streams (its previous launch sits on another stream, not yet
    # synchronised) is refused before anything is recorded into it; after fmd_batch_sync() the same capture goes through
    iq, pcm, lens = bufs[0]
    b.run_device(iq, 1, pcm, lens)                     # on the batch's own stream
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=st):
        with pytest.raises(R.FmdError):
            b.run_device(iq, 1, pcm, lens, hip_stream=st.cuda_stream)
        lens.zero_()                                   # (something for the capture to hold)
    b.sync()
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3, stream=st):
        b.run_device(iq, 1, pcm, lens, hip_stream=st.cuda_stream)
        b.run_device(iq, 1, pcm, lens, hip_stream=st.cuda_stream)
    g3.replay()
    torch.cuda.synchronize()
    b.close()


def test_ingest_callback_and_pump(R):
    """rtlsdr_read_async-shaped ingest: odd-sized callback buffers -> pinned ring -> batch."""
    from oracle import OracleStream, lcg_bytes
    L = R.lib()
    ns, nb = 2, 3
    cfg = R.wbfm_config(math=R.MATH_EXACT, **CONFIGS["stereo_300k"])
    b = R.BatchDemod(cfg, ns)
    iqs = [lcg_bytes(nb * BL + 12345, 777 + s)[0] for s in range(ns)]       # a partial block is left over
    rings = []
    for s in range(ns):
        h = C.c_void_p()
        assert L.fmd_ingest_create(C.byref(h), b._h, s, 0) == 0
        rings.append(h)
    for s in range(ns):
        pos = 0
        while pos < iqs[s].size:                                            # 100000-byte "USB transfers"
            n = min(100000, iqs[s].size - pos)
            chunk = np.ascontiguousarray(iqs[s][pos:pos + n])
            L.fmd_ingest_callback(chunk.ctypes.data, n, rings[s])
            pos += n
        assert L.fmd_ingest_buffered(rings[s]) == iqs[s].size
        assert L.fmd_ingest_dropped(rings[s]) == 0
    pcm = np.zeros((ns, 8, b.pcm_stride), dtype=np.int16)
    lens = np.zeros((ns, 8), dtype=np.int32)
    got = L.fmd_batch_pump(b._h, 8, pcm.ctypes.data, lens.ctypes.data)
    assert got == nb                                                        # only whole blocks
    pcm = pcm.reshape(-1)[: ns * nb * b.pcm_stride].reshape(ns, nb, b.pcm_stride)
    lens = lens.reshape(-1)[: ns * nb].reshape(ns, nb)
    for s in range(ns):
        want, wl = OracleStream(**CONFIGS["stereo_300k"]).run(iqs[s][: nb * BL], BL)
        assert np.array_equal(lens[s], wl)
        assert np.array_equal(np.concatenate([pcm[s, k, :lens[s, k]] for k in range(nb)]), want)
        assert L.fmd_ingest_buffered(rings[s]) == 12345
        L.fmd_ingest_destroy(rings[s])


def test_pipelined_pump_two_jobs_in_flight(R):
    """fmd_batch_pump_begin/_end: two jobs queued back to back (the second is staged while the
    first runs) give the same PCM as one sequential run: state is carried job to job."""
    from oracle import OracleStream, lcg_bytes
    L = R.lib()
    ns, nb1, nb2 = 3, 2, 3
    cfg = R.wbfm_config(math=R.MATH_EXACT, **CONFIGS["stereo_300k"])
    b = R.BatchDemod(cfg, ns)
    iqs = [lcg_bytes((nb1 + nb2) * BL, 4242 + s)[0] for s in range(ns)]
    rings = []
    for s in range(ns):
        h = C.c_void_p()
        assert L.fmd_ingest_create(C.byref(h), b._h, s, 0) == 0
        rings.append(h)

    def feed(s, lo, hi):
        chunk = np.ascontiguousarray(iqs[s][lo:hi])
        L.fmd_ingest_callback(chunk.ctypes.data, chunk.size, rings[s])

    for s in range(ns):
        feed(s, 0, nb1 * BL)
    assert L.fmd_batch_pump_begin(b._h, 8) == nb1
    for s in range(ns):
        feed(s, nb1 * BL, (nb1 + nb2) * BL)
    assert L.fmd_batch_pump_begin(b._h, 8) == nb2           # second job queued behind the first
    assert L.fmd_batch_pump_begin(b._h, 8) < 0              # a third would need a slot: refused
    outs = []
    for nb in (nb1, nb2):
        pcm = np.zeros((ns, nb, b.pcm_stride), dtype=np.int16)
        lens = np.zeros((ns, nb), dtype=np.int32)
        assert L.fmd_batch_pump_end(b._h, pcm.ctypes.data, lens.ctypes.data) == nb
        outs.append((pcm, lens))
    pcm = np.zeros((ns, 1, b.pcm_stride), dtype=np.int16)
    lens = np.zeros((ns, 1), dtype=np.int32)
    assert L.fmd_batch_pump_end(b._h, pcm.ctypes.data, lens.ctypes.data) == 0     # nothing left in flight
    for s in range(ns):
        want, wl = OracleStream(**CONFIGS["stereo_300k"]).run(iqs[s], BL)
        got_l = np.concatenate([o[1][s] for o in outs])
        assert np.array_equal(got_l, wl)
        got = np.concatenate([o[0][s, k, :o[1][s, k]] for o in outs for k in range(o[1].shape[1])])
        assert np.array_equal(got, want)
        L.fmd_ingest_destroy(rings[s])


def test_ingest_overflow_keeps_newest(R):
    L = R.lib()
    cfg = R.wbfm_config(math=R.MATH_EXACT, block_len=4096, **CONFIGS["mono_300k"])
    b = R.BatchDemod(cfg, 1)
    h = C.c_void_p()
    assert L.fmd_ingest_create(C.byref(h), b._h, 0, 8192) == 0
    data = (np.arange(3 * 8192) % 251).astype(np.uint8)
    for k in range(6):
        chunk = np.ascontiguousarray(data[k * 4096:(k + 1) * 4096])
        L.fmd_ingest_callback(chunk.ctypes.data, 4096, h)
    assert L.fmd_ingest_buffered(h) == 8192 and L.fmd_ingest_dropped(h) == 4 * 4096
    L.fmd_ingest_destroy(h)


@pytest.mark.parametrize("wav", [False, True])
def test_replay_driver(R, tmp_path, wav):
    """fmd_replay = the reference's demod thread for recorded IQ: whole blocks only, PCM/WAV out."""
    import os
    import subprocess
    from oracle import OracleStream, lcg_bytes
    exe = os.path.join(os.path.dirname(R.library_path()), "fmd_replay")
    nb = 5
    files, iqs = [], []
    for s in range(2):
        iq = lcg_bytes(nb * BL + 999, 4242 + s)[0]
        p = tmp_path / ("in%d.u8" % s)
        iq.tofile(p)
        files.append(str(p)); iqs.append(iq)
    prefix = str(tmp_path / "out")
    cmd = [exe, "-s", "300000", "-e", "-n", "2", "-o", prefix] + (["-w"] if wav else []) + files
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for s in range(2):
        want, _ = OracleStream(**CONFIGS["stereo_300k"]).run(iqs[s][: nb * BL], BL)
        raw = open(prefix + "%d.%s" % (s, "wav" if wav else "pcm"), "rb").read()
        if wav:
            assert raw[:4] == b"RIFF" and int.from_bytes(raw[40:44], "little") == len(raw) - 44
            raw = raw[260:]
        assert np.array_equal(np.frombuffer(raw, dtype=np.int16), want)


def test_full_size_batch_properties(R):
    """BASELINE.json configs[2] at full size (256 streams x 16 blocks, device resident),
    through properties that do not need 256 oracle runs:
      * every stream fed the same IQ produces the same PCM (streams are independent and the
        time-chunk replay is deterministic), and that PCM equals the oracle's (bit-exact);
      * distinct inputs in odd streams do not disturb the even streams."""
    import torch
    from oracle import OracleStream, lcg_bytes
    S, B = 256, 16
    dev = torch.device("cuda:0")
    base = lcg_bytes(B * BL, 2024)[0]
    other = lcg_bytes(B * BL, 99)[0]
    iq = torch.empty((S, B * BL), dtype=torch.uint8, device=dev)
    iq[0::2] = torch.from_numpy(base).to(dev)
    iq[1::2] = torch.from_numpy(other).to(dev)
    want, wl = OracleStream(**CONFIGS["stereo_300k"]).run(base, BL)
    want_o, wl_o = OracleStream(**CONFIGS["stereo_300k"]).run(other, BL)
    for math, tol in [(R.MATH_EXACT, 0)] + [(m, 1) for m in R.FAST_MATHS]:
        b = R.BatchDemod(R.wbfm_config(math=math, **CONFIGS["stereo_300k"]), S)
        pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
        lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        b.run_device(iq, B, pcm, lens)
        b.sync()
        assert bool((lens[0::2] == torch.from_numpy(wl).to(dev)).all())
        assert bool((lens[1::2] == torch.from_numpy(wl_o).to(dev)).all())
        assert bool((pcm[0::2] == pcm[0]).all()) and bool((pcm[1::2] == pcm[1]).all())
        for s, (w, l) in ((0, (want, wl)), (1, (want_o, wl_o)), (254, (want, wl)), (255, (want_o, wl_o))):
            p = pcm[s].cpu().numpy()
            got = np.concatenate([p[k, :l[k]] for k in range(B)])
            assert np.abs(got.astype(np.int32) - w.astype(np.int32)).max() <= tol
        b.close()


def test_chunking_does_not_change_results(R, lcg40):
    """The same launch without time chunks (fmd_batch_set_time_split < 0), with a different worker target and with
    the default split gives identical PCM and identical carried state."""
    nb = 16
    outs, states = [], []
    for split in (-1, 3, 0):
        got, lens, b = gpu_run(R, CONFIGS["stereo_300k"], lcg40[: nb * BL], nb, R.MATH_EXACT, time_split=split)
        outs.append(got[0])
        st = b.get_state(0)
        states.append((st.acc, st.pre_r, st.pre_j, st.pp, st.deemph_l, st.deemph_r, list(st.br)[:90], list(st.bs)[:90]))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert states[0] == states[1] == states[2]


@pytest.mark.parametrize("name", ["stereo_300k", "mono_300k", "nfm_25k"])
def test_chunking_fast_math_stays_within_one_lsb(R, lcg40, name, fast_math):
    """Fast kernels, whole tiles (the own-words decimator hands partial sums from lane to lane and from tile to tile):
    one worker for the whole launch, many short time chunks and the default split all stay within 1 LSB of the oracle,
    and the block lengths are the oracle's."""
    nb = 16
    want, wlens, _ = oracle_run(CONFIGS[name], lcg40[: nb * BL])
    for split in (-1, 3, 24, 0):
        got, lens, b = gpu_run(R, CONFIGS[name], lcg40[: nb * BL], nb, fast_math, time_split=split)
        b.close()
        assert np.array_equal(lens[0], wlens), split
        diff = int(np.abs(got[0].astype(np.int32) - want.astype(np.int32)).max())
        assert diff <= 1, "%s time_split %s: fast PCM differs from the oracle by %d LSB" % (name, split, diff)


def test_misaligned_iq_pointer_is_rejected(R):
    import torch
    cfg = R.wbfm_config(math=R.MATH_FAST, **CONFIGS["stereo_300k"])
    b = R.BatchDemod(cfg, 1)
    dev = torch.device("cuda:0")
    buf = torch.zeros(BL + 64, dtype=torch.uint8, device=dev)
    pcm = torch.zeros(b.pcm_stride, dtype=torch.int16, device=dev)
    lens = torch.zeros(1, dtype=torch.int32, device=dev)
    with pytest.raises(R.FmdError, match="16-byte aligned"):
        b.run_device(buf.data_ptr() + 4, 1, pcm, lens)


@pytest.mark.parametrize("mode", [2, 1])
def test_fast_math_on_noise_streams_stays_within_one_lsb(R, mode, fast_math):
    """Noise is the worst case for a non-bit-exact discriminator: phase steps land arbitrarily
    close to the +-pi branch cut, where a last-bit difference flips the output by 2 pi.  The fast
    kernels redo such samples in exact arithmetic; seed 99 holds a known crossing (block 4)."""
    import torch
    from oracle import OracleStream, lcg_bytes
    S, B = 24, 8
    kw = dict(rate_in=300000, rate_out2=48000, mode=mode)
    dev = torch.device("cuda:0")
    seeds = [99] + list(range(5000, 5000 + S - 1))
    host = [lcg_bytes(B * BL, sd)[0] for sd in seeds]
    iq = torch.stack([torch.from_numpy(h) for h in host]).to(dev)
    b = R.BatchDemod(R.wbfm_config(math=fast_math, **kw), S)
    pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
    lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    b.run_device(iq, B, pcm, lens)
    b.sync()
    for i in range(S):
        want, wl = OracleStream(**kw).run(host[i], BL)
        p, l = pcm[i].cpu().numpy(), lens[i].cpu().numpy()
        assert np.array_equal(l, wl)
        got = np.concatenate([p[k, :l[k]] for k in range(B)])
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1, "seed %d: |diff| %d at %d" % (seeds[i], d.max(), int(d.argmax()))



def test_fast_math_nfm_noise_next_to_the_origin(R, fast_math):
    """NFM on uniform noise (bench.py's generator and seed): stream 0 holds a decimated sample of
    magnitude 1e-4, whose phase a 1e-7 rounding difference moves by 3 LSB of PCM; the fast kernels
    redo samples that close to the origin in exact arithmetic."""
    import torch
    from oracle import OracleStream
    S, B = 4, 16
    kw = dict(rate_in=25000, rate_out2=12500, mode=1)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(12345)
    iq = torch.randint(0, 256, (S, B, BL), dtype=torch.uint8, device=dev, generator=g)
    b = R.BatchDemod(R.wbfm_config(math=fast_math, **kw), S)
    pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
    lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    b.run_device(iq, B, pcm, lens)
    b.sync()
    for i in range(S):
        want, wl = OracleStream(**kw).run(iq[i].cpu().numpy().reshape(-1), BL)
        p, l = pcm[i].cpu().numpy(), lens[i].cpu().numpy()
        assert np.array_equal(l, wl)
        got = np.concatenate([p[k, :l[k]] for k in range(B)])
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1, "stream %d: |diff| %d at %d" % (i, d.max(), int(d.argmax()))


@pytest.mark.parametrize("mode", ["stereo", "nfm"])
def test_bench_fm_input(R, mode):
    """bench.py's synthetic FM broadcasts (stereo multiplex with pilot / narrow FM): exact kernels
    bit-identical, fast kernels within 1 LSB, and the stereo decode is doing something (L != R)."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle import OracleStream
    S, B = 3, 6
    nfm = mode == "nfm"
    kw = dict(rate_in=25000, rate_out2=12500, mode=1) if nfm else dict(rate_in=300000, rate_out2=48000, mode=2)
    dev = torch.device("cuda:0")
    iq = bench.synth_fm_iq(torch, dev, S, B * BL // 2, 200e3 if nfm else 2.4e6, not nfm, 777).view(S, B, BL)
    host = iq.cpu().numpy()
    for math, tol in [(R.MATH_EXACT, 0)] + [(m, 1) for m in R.FAST_MATHS]:
        b = R.BatchDemod(R.wbfm_config(math=math, **kw), S)
        pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
        lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        b.run_device(iq, B, pcm, lens)
        b.sync()
        for i in range(S):
            want, wl = OracleStream(**kw).run(host[i].reshape(-1), BL)
            p, l = pcm[i].cpu().numpy(), lens[i].cpu().numpy()
            assert np.array_equal(l, wl)
            got = np.concatenate([p[k, :l[k]] for k in range(B)])
            assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= tol
            if not nfm and math == R.MATH_EXACT:
                left, right = got[4000::2].astype(np.float64), got[4001::2].astype(np.float64)
                assert left.std() > 200 and right.std() > 200 and np.abs(left - right).std() > 100



@pytest.mark.parametrize("mode", [2, 1, "nfm"])
def test_full_size_every_stream_against_the_oracle(R, mode):
    """BASELINE.json configs[2] at full size with 256 DIFFERENT streams (LCG seeds 30000 + s), 16 blocks each:
    every stream of the launch is compared with the oracle (threads: the oracle call releases the GIL) -
    exact kernels bit for bit, fast kernels within 1 LSB.  537 M samples per launch, nothing duplicated."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from oracle import OracleStream, lcg_bytes
    S, B = 256, 16
    kw = dict(rate_in=25000, rate_out2=12500, mode=1) if mode == "nfm" else dict(rate_in=300000, rate_out2=48000, mode=mode)
    dev = torch.device("cuda:0")
    host = np.empty((S, B * BL), dtype=np.uint8)

    def fill(s):
        host[s] = lcg_bytes(B * BL, 30000 + s)[0]

    with ThreadPoolExecutor(16) as ex:
        list(ex.map(fill, range(S)))
    iq = torch.from_numpy(host).to(dev)

    def oracle(s):
        return OracleStream(**kw).run(host[s], BL)

    with ThreadPoolExecutor(16) as ex:
        want = list(ex.map(oracle, range(S)))
    for math, tol in [(R.MATH_EXACT, 0)] + [(m, 1) for m in R.FAST_MATHS]:
        b = R.BatchDemod(R.wbfm_config(math=math, **kw), S)
        pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
        lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        b.run_device(iq, B, pcm, lens)
        b.sync()
        p_all, l_all = pcm.cpu().numpy(), lens.cpu().numpy()
        worst = 0
        for s in range(S):
            w, wl = want[s]
            assert np.array_equal(l_all[s], wl), "stream %d lens" % s
            got = np.concatenate([p_all[s, k, :wl[k]] for k in range(B)])
            d = int(np.abs(got.astype(np.int32) - w.astype(np.int32)).max())
            worst = max(worst, d)
            assert d <= tol, "math %d stream %d: |diff| %d" % (math, s, d)
        b.close()


@pytest.mark.parametrize("kw", [
    dict(rate_in=300000, rate_out2=48000, mode=2),
    dict(rate_in=48000, rate_out2=16000, mode=2),          # the wide pilot filter (fmd_host.c: K grows with sum fp^2)
], ids=["300k", "48k"])
@pytest.mark.parametrize("via", ["run_device", "pump"])
def test_fast_hand_over_between_one_block_launches(R, fast_math, kw, via):
    """Real-time use: ONE reference block per launch, so every launch's first `size` samples reach back into what the
    launch before handed over.  Noise input (no pilot: the carrier redo fires every few tiles, and with 64 streams x 64
    launches many of those samples lie in that window), every stream within 1 LSB of the oracle - through
    fmd_batch_run_device and through the ingest rings + fmd_batch_pump.  The hand-over is exact: the last decimated
    sample and the last `size` discriminator outputs are carried in the reference's own bits
    (src/rtl_fm_player.c:430-436, :533-568 carry exact rings from block to block)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from oracle import OracleStream
    S, NL = 64, 64
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(777)
    iq = torch.randint(0, 256, (S, NL, BL), dtype=torch.uint8, device=dev, generator=g)
    torch.cuda.synchronize()
    h_iq = iq.cpu().numpy()
    b = R.BatchDemod(R.wbfm_config(math=fast_math, **kw), S)
    got = [[] for _ in range(S)]
    glens = np.zeros((S, NL), dtype=np.int32)
    if via == "run_device":
        pcm = torch.zeros((S, 1, b.pcm_stride), dtype=torch.int16, device=dev)
        lens = torch.zeros((S, 1), dtype=torch.int32, device=dev)
        for k in range(NL):
            blk = iq[:, k:k + 1].contiguous()
            torch.cuda.synchronize()
            b.run_device(blk, 1, pcm, lens)
            b.sync()
            p, l = pcm.cpu().numpy(), lens.cpu().numpy()
            glens[:, k] = l[:, 0]
            for s in range(S):
                got[s].append(p[s, 0, :l[s, 0]].copy())
    else:
        L = R.lib()
        rings = []
        for s in range(S):
            h = C.c_void_p()
            assert L.fmd_ingest_create(C.byref(h), b._h, s, 0) == 0
            rings.append(h)
        pcm = np.zeros((S, 1, b.pcm_stride), dtype=np.int16)
        lens = np.zeros((S, 1), dtype=np.int32)
        for k in range(NL):
            for s in range(S):
                blk = np.ascontiguousarray(h_iq[s, k])
                L.fmd_ingest_callback(blk.ctypes.data, BL, rings[s])
            assert L.fmd_batch_pump(b._h, 1, pcm.ctypes.data, lens.ctypes.data) == 1
            glens[:, k] = lens[:, 0]
            for s in range(S):
                got[s].append(pcm[s, 0, :lens[s, 0]].copy())
        for h in rings:
            L.fmd_ingest_destroy(h)

    def check(s):
        want, wl = OracleStream(**kw).run(h_iq[s].reshape(-1), BL)
        if not np.array_equal(glens[s], wl):
            return 1 << 20
        return int(np.abs(np.concatenate(got[s]).astype(np.int32) - want.astype(np.int32)).max())

    with ThreadPoolExecutor(16) as ex:
        diffs = list(ex.map(check, range(S)))
    b.close()
    assert max(diffs) <= 1, "stream %d: |diff| %d" % (diffs.index(max(diffs)), max(diffs))


@pytest.mark.parametrize("mode", [2, 1])
def test_same_input_same_output_soak(R, fast_math, mode):
    """Determinism under full occupancy: 256 streams fed the SAME IQ (noise) for 10 launches of 16 blocks - every
    stream's PCM must equal stream 0's bit for bit, every launch.  Workers of one SIMD share the vector ALU, the matrix
    pipe and LDS; nothing one of them does may reach another's results.  (Round 3 measured a split-bf16 MFMA form of
    stage C that failed exactly this - about one tile in a thousand, only with two or more workers per SIMD - while
    passing every single-stream parity test: tools/experiments/mpx_tile_bf16.inc.)"""
    import torch
    from oracle import lcg_bytes
    S, B = 256, 16
    dev = torch.device("cuda:0")
    one = torch.from_numpy(lcg_bytes(B * BL, 2024)[0]).to(dev).view(1, B * BL)
    iq = one.expand(S, B * BL).contiguous()
    b = R.BatchDemod(R.wbfm_config(math=fast_math, rate_in=300000, rate_out2=48000, mode=mode), S)
    pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
    lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    first = None
    for rep in range(10):
        b.reset()
        b.run_device(iq, B, pcm, lens)
        b.sync()
        p = pcm.view(S, -1)
        if first is None:
            first = p[0].clone()
        odd = int((p != first.unsqueeze(0)).any(dim=1).sum().item())
        assert odd == 0, "launch %d: %d of %d streams differ from stream 0 of the first launch" % (rep, odd, S)
    b.close()
