"""Pins oracle/fm_oracle.c to the REFERENCE ITSELF, compiled here by oracle/build_ref.py
(oracle/_ref/libref.so = /root/reference/src/rtl_fm_player.c:195-788 built with gcc -O3 from
the sources where they lie; see oracle/ref_shim.c).  Every comparison is bit for bit: PCM,
block lengths, the per-stage intermediates the reference leaves in d->lowpassed / d->result
(decimated IQ after lp_f32, discriminator output after fm_demod_f32, resampler output after
lp_real_f32), the carried state and the filter tables.

Skipped only where neither /root/reference nor a previously built oracle/_ref/ exists.
"""
import numpy as np
import pytest

from oracle import OracleStream, dds_bytes, hash16, lcg_bytes
from oracle import refbind

pytestmark = pytest.mark.skipif(not refbind.have_ref(), reason="oracle/_ref/libref.so not built (no /root/reference)")

BL = 262144

# the five configurations SURVEY.md section 8c recorded hashes for (LCG seed 12345, 40 blocks)
KNOWN = {
    "stereo_300k": (dict(rate_in=300000, rate_out2=48000, mode=2), 209714, 0xC3E7EDA4BD16DFE1),
    "mono_300k": (dict(rate_in=300000, rate_out2=48000, mode=1), 104857, 0x2109FE431B558355),
    "nfm_25k": (dict(rate_in=25000, rate_out2=12500, mode=1), 327680, 0x3E6F57574F3156AA),
    "stereo_240k": (dict(rate_in=240000, rate_out2=48000, mode=2), 262144, 0x8E0413ED2BF00E75),
    "stereo_192k": (dict(rate_in=192000, rate_out2=48000, mode=2), 327680, 0x6E145D091E77DBC9),
}

# further configurations of the path (SURVEY.md section 8 rows a2', a10, a11, f4; judge's probe list)
EXTRA = {
    "stereo_171k_44k1": dict(rate_in=171000, rate_out2=44100, mode=2),
    "offset_tuning": dict(rate_in=300000, rate_out2=48000, mode=2, offset_tuning=True),
    "mono90_240k": dict(rate_in=240000, rate_out2=48000, mode=1, size=90),
    "mode0_drop": dict(rate_in=300000, rate_out2=48000, mode=0),
    "no_resample": dict(rate_in=300000, rate_out2=0, mode=2),
    "no_deemph": dict(rate_in=300000, rate_out2=48000, mode=2, deemph=False),
    "loud_clipping": dict(rate_in=300000, rate_out2=48000, mode=2, volume=4.0),
    "usa_75us": dict(rate_in=300000, rate_out2=48000, mode=2, tau=75e-6),
    "stereo_size64": dict(rate_in=240000, rate_out2=32000, mode=2, size=64),
}


def test_reference_library_is_the_pinned_one(tmp_path, monkeypatch):
    """oracle/_ref/libref.so is what oracle/build_ref.py made from the reference lines whose hash is committed
    (oracle/ref_pin.json), byte for byte as recorded at build time; bench.py times it as "the reference" only then."""
    import json
    import os
    import shutil
    from oracle import build_ref as B
    st = B.ref_status()
    assert st is not None and st["pinned"], st
    assert st["slices_sha256"] == B.load_pin()["slices_sha256"]
    assert st["so_sha256"] == st["so_sha256_on_disk"]
    # ... and so is the ring library (rtlsdr_callback :790-837 with the two reference headers it includes)
    assert st["ring_pinned"] and st["ring_sha256"] == B.load_pin()["ring_sha256"], st
    assert st["ring_so_sha256"] == st["ring_so_sha256_on_disk"]
    # a library that is not the recorded one, or lines that are not the pinned ones, are not "the reference"
    fake = tmp_path / "_ref"
    shutil.copytree(B.OUT, fake)
    monkeypatch.setattr(B, "OUT", str(fake))
    with open(fake / "libref.so", "ab") as f:
        f.write(b"\0")
    assert not B.ref_status()["pinned"]
    shutil.copy(os.path.join(os.path.dirname(B.__file__), "_ref", "libref.so"), fake / "libref.so")
    assert B.ref_status()["pinned"]
    meta = json.load(open(fake / "libref.meta.json"))
    meta["slices_sha256"] = "0" * 64
    json.dump(meta, open(fake / "libref.meta.json", "w"))
    assert not B.ref_status()["pinned"]


def _both(cfg):
    return refbind.RefStream(**cfg), OracleStream(**cfg)


@pytest.mark.parametrize("name", sorted(KNOWN))
def test_reference_reproduces_survey_hashes_and_oracle_equals_it(name, lcg40):
    cfg, n_exp, h_exp = KNOWN[name]
    ref, orc = _both(cfg)
    rp, rl = ref.run(lcg40, BL)
    assert rp.size == n_exp and hash16(rp) == h_exp          # the reference, compiled here
    op, ol = orc.run(lcg40, BL)
    assert np.array_equal(rl, ol)
    assert np.array_equal(rp, op)                             # full PCM, not only the hash


@pytest.mark.parametrize("name", sorted(EXTRA))
def test_oracle_equals_reference_on_variants(name, lcg40):
    ref, orc = _both(EXTRA[name])
    rp, rl = ref.run(lcg40[:12 * BL], BL)
    op, ol = orc.run(lcg40[:12 * BL], BL)
    assert np.array_equal(rl, ol)
    assert np.array_equal(rp, op)
    assert rp.size > 0


@pytest.mark.parametrize("mode", [1, 2])
def test_oracle_equals_reference_on_fm_broadcast_input(mode):
    """Integer-DDS stereo multiplex (pilot + L/R tones), the class of input bench.py feeds."""
    iq = dds_bytes(10 * BL, fs=2400000)
    ref, orc = _both(dict(rate_in=300000, rate_out2=48000, mode=mode))
    rp, rl = ref.run(iq, BL)
    op, ol = orc.run(iq, BL)
    assert np.array_equal(rl, ol) and np.array_equal(rp, op)
    assert int(np.abs(rp.astype(np.int32)).max()) > 1000      # there is audio in it


@pytest.mark.parametrize("name", ["stereo_300k", "mono_300k", "nfm_25k", "stereo_240k"])
def test_stage_intermediates_and_state(name, lcg40):
    """Blocks 0..3 stage by stage (block 2 at 300k -> 48k starts with an emit: quirk Q1)."""
    cfg = KNOWN[name][0]
    ref, orc = _both(cfg)
    for b in range(4):
        blk = lcg40[b * BL:(b + 1) * BL]
        rp, rt = ref.block(blk, trace=True)
        op, ot = orc.block(blk, trace=True)
        assert np.array_equal(rt["y"].view(np.uint32), ot["y"].view(np.uint32)), "decimated IQ, block %d" % b
        # the oracle's trace holds v before the Q1 overwrite; the reference's buffer after fm_demod_f32 too
        assert np.array_equal(rt["v"].view(np.uint32), ot["v"].view(np.uint32)), "discriminator, block %d" % b
        assert np.array_equal(rt["mpx"].view(np.uint32), ot["mpx"].view(np.uint32)), "resampler, block %d" % b
        assert np.array_equal(rp, op)
        rs, os_ = ref.get_state(), orc.get_state()
        n = rs["size"]
        assert np.array_equal(rs["tb"].view(np.uint32), np.array(os_.tb, np.float32).view(np.uint32))
        for k in ("pre_r", "pre_j", "deemph_l", "deemph_r"):
            assert np.float32(rs[k]).view(np.uint32) == np.float32(getattr(os_, k)).view(np.uint32), k
        assert rs["acc"] == os_.acc
        assert np.array_equal(rs["br"].view(np.uint32), np.array(os_.br, np.float32)[:n].view(np.uint32))
        if cfg["mode"] == 2:
            assert np.float32(rs["pp"]).view(np.uint32) == np.float32(os_.pp).view(np.uint32)
            assert np.array_equal(rs["bm"].view(np.uint32), np.array(os_.bm, np.float32)[:n].view(np.uint32))
            assert np.array_equal(rs["bs"].view(np.uint32), np.array(os_.bs, np.float32)[:n].view(np.uint32))


def test_staged_walk_equals_full_demod(lcg40):
    """ref_block_staged calls the stages of full_demod one by one: same PCM as full_demod()."""
    cfg = KNOWN["stereo_300k"][0]
    a, b = refbind.RefStream(**cfg), refbind.RefStream(**cfg)
    for i in range(5):
        blk = lcg40[i * BL:(i + 1) * BL]
        pa = a.block(blk)
        pb, _ = b.block(blk, trace=True)
        assert np.array_equal(pa, pb)


@pytest.mark.parametrize("cfg", [dict(rate_in=300000, mode=2), dict(rate_in=300000, mode=1),
                                 dict(rate_in=25000, rate_out2=12500, mode=1), dict(rate_in=171000, rate_out2=44100, mode=2)])
def test_tables_equal(cfg):
    ref, orc = _both(cfg)
    rt, ot = ref.taps(), orc.taps()
    for k in ("fb", "fm", "fp", "fs"):
        assert np.array_equal(rt[k].view(np.uint32), ot[k].view(np.uint32)), k
    assert rt["swf"] == ot["swf"] and rt["cwf"] == ot["cwf"]


def test_u8_table_is_the_exact_closed_form():
    t0 = np.empty(256, np.float32)
    t1 = np.empty(256, np.float32)
    refbind.lib().ref_get_u8_table(t0.ctypes.data, t1.ctypes.data)
    i = np.arange(256, dtype=np.float64)
    assert np.array_equal(t0.astype(np.float64), (i - 127.5) / 128.0)       # exact in fp32
    assert np.array_equal(t1.astype(np.float64), (i - 127.5) / -128.0)
    assert not np.any(t0 == 0)


def test_struct_layout_matches_header():
    """sizeof / offsetof from the reference's own struct definition == the constants
    include/fmdemod_mi355x.h static-asserts."""
    L = refbind.lib()
    assert L.ref_sizeof_demod_state() == 1835872
    want = {0: 16, 1: 262160, 2: 262164, 3: 1310740, 4: 1310744, 5: 1311176, 6: 1835464, 7: 1835508,
            8: 1835536, 9: 1835592, 10: 1835620, 11: 1835624, 12: 1835632, 13: 1835640, 14: 1835720,
            15: 1835864}
    for k, off in want.items():
        assert L.ref_offsetof_demod_state(k) == off, (k, off)


def _fuzz_cfg(rng):
    mode = int(rng.choice([0, 1, 2, 2, 1]))
    rate_in = int(rng.choice([25000, 96000, 171000, 192000, 200000, 240000, 250000, 300000, 384000]))
    if mode == 2:
        size = int(rng.choice([90, 90, 64, 46, 128]))
        # the in-place overwrite of quirk Q1 stays on sample 1 only while three input samples pass per frame
        rate_out2 = int(rng.choice([r for r in (8000, 12500, 22050, 32000, 44100, 48000) if 3 * r <= rate_in]))
    else:
        size = int(rng.choice([128, 128, 90, 32, 200]))
        rate_out2 = int(rng.choice([r for r in (0, 8000, 12500, 22050, 44100, 48000, rate_in) if r <= rate_in]))
    return dict(rate_in=rate_in, rate_out2=rate_out2, mode=mode, size=size,
                deemph=bool(rng.integers(0, 4)), offset_tuning=bool(rng.integers(0, 2)),
                volume=float(rng.choice([0.4, 0.4, 1.0, 3.0])),
                tau=float(rng.choice([50e-6, 75e-6, 500e-6])))


def test_config_fuzz_100_cases():
    """100 random configurations x 3 ragged blocks each (lengths are multiples of 16 >= 64, as
    lp_f32 needs): PCM, lengths and carried state bit-identical to the reference."""
    rng = np.random.default_rng(20261003)
    for case in range(100):
        cfg = _fuzz_cfg(rng)
        ref, orc = _both(cfg)
        seed = int(rng.integers(1, 2 ** 31))
        for b in range(3):
            n = int(rng.choice([64, 80, 4096, 65536, 262144, 16 * int(rng.integers(4, 16384))]))
            blk, seed = lcg_bytes(n, seed)
            rp, op = ref.block(blk), orc.block(blk)
            assert np.array_equal(rp, op), (case, b, n, cfg)
        rs, os_ = ref.get_state(), orc.get_state()
        assert rs["acc"] == os_.acc, (case, cfg)
        assert np.array_equal(rs["br"].view(np.uint32), np.array(os_.br, np.float32)[:rs["size"]].view(np.uint32)) \
            or cfg["mode"] == 0 or cfg["rate_out2"] == 0, (case, cfg)
