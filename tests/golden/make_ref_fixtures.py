#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors.npz from the REFERENCE compiled here.

    python tests/golden/make_ref_fixtures.py        (needs /root/reference; see oracle/build_ref.py)

The vectors are outputs of oracle/_ref/libref.so - /root/reference/src/rtl_fm_player.c:195-788
built with gcc -O3 - on inputs every box regenerates bit for bit (the survey's LCG byte stream,
SURVEY.md section 8c, and the integer-DDS FM multiplex of oracle/fm_oracle.c).  They are data
(inputs are seeds, outputs are numbers): nothing of the reference's text is stored.

Per configuration (SURVEY.md section 8c fixture plan):
  lens        result_len of each of the 40 blocks
  hash        64-bit FNV-style hash over all PCM of the 40 blocks
  pcm_first / pcm_last    first / last 256 int16 of the run
  q1_pcm      the first 96 int16 (~ 40 frames and more) of block 2, where the resampler emits on the
              block's first sample at 300k -> 48k (quirk Q1); also kept for the other configs
  y_head / y_tail   decimated IQ of blocks 0 and 2: first / last 64 floats   (after lp_f32)
  v_head / v_tail   discriminator output of blocks 0 and 2: first / last 64 floats (after fm_demod_f32)
  mpx_head    resampler output of blocks 0 and 2: first 96 floats            (after lp_real_f32)
  fb, fm, fp, fs, swf, cwf, lam   filter tables and scalars
  state_*     carried state after the 40 blocks (rings oldest -> newest)
and for the DDS broadcast input (10 blocks): lens, hash, pcm_first, pcm_last.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import dds_bytes, deemph_lambda, hash16, lcg_bytes  # noqa: E402
from oracle import refbind  # noqa: E402

BL = 262144
CONFIGS = {
    "stereo_300k": dict(rate_in=300000, rate_out2=48000, mode=2),
    "mono_300k": dict(rate_in=300000, rate_out2=48000, mode=1),
    "nfm_25k": dict(rate_in=25000, rate_out2=12500, mode=1),
    "stereo_240k": dict(rate_in=240000, rate_out2=48000, mode=2),
    "stereo_192k": dict(rate_in=192000, rate_out2=48000, mode=2),
    "stereo_171k_44k1": dict(rate_in=171000, rate_out2=44100, mode=2),
    "offset_tuning": dict(rate_in=300000, rate_out2=48000, mode=2, offset_tuning=True),
    "mono90_240k": dict(rate_in=240000, rate_out2=48000, mode=1, size=90),
}
DDS_CONFIGS = {
    "dds_stereo_300k": dict(rate_in=300000, rate_out2=48000, mode=2),
    "dds_mono_300k": dict(rate_in=300000, rate_out2=48000, mode=1),
}


def main():
    if not refbind.have_ref():
        sys.exit("make_ref_fixtures: oracle/_ref/libref.so cannot be built here (no /root/reference)")
    lcg40, _ = lcg_bytes(40 * BL, 12345)
    out = {}
    for name, cfg in CONFIGS.items():
        pcm, lens = refbind.RefStream(**cfg).run(lcg40, BL)
        out[name + "/lens"] = lens.astype(np.int32)
        out[name + "/hash"] = np.array([hash16(pcm)], dtype=np.uint64)
        out[name + "/pcm_first"] = pcm[:256].copy()
        out[name + "/pcm_last"] = pcm[-256:].copy()
        s = refbind.RefStream(**cfg)
        for b in range(3):
            p, tr = s.block(lcg40[b * BL:(b + 1) * BL], trace=True)
            if b == 2:
                out[name + "/q1_pcm"] = p[:96].copy()
            if b in (0, 2):
                out[name + "/y_head%d" % b] = tr["y"][:64].copy()
                out[name + "/y_tail%d" % b] = tr["y"][-64:].copy()
                out[name + "/v_head%d" % b] = tr["v"][:64].copy()
                out[name + "/v_tail%d" % b] = tr["v"][-64:].copy()
                out[name + "/mpx_head%d" % b] = tr["mpx"][:96].copy()
        t = s.taps()
        for k in ("fb", "fm", "fp", "fs"):
            out[name + "/" + k] = t[k]
        lam = deemph_lambda(cfg["rate_out2"])
        out[name + "/scalars"] = np.array([t["swf"], t["cwf"], lam], dtype=np.float32)
        s40 = refbind.RefStream(**cfg)
        s40.run(lcg40, BL)
        st = s40.get_state()
        out[name + "/state_f"] = np.array([st["pre_r"], st["pre_j"], st["pp"], st["deemph_l"], st["deemph_r"]], np.float32)
        out[name + "/state_acc"] = np.array([st["acc"]], np.int32)
        out[name + "/state_tb"] = st["tb"]
        for k in ("br", "bm", "bs"):
            out[name + "/state_" + k] = st[k]
    dds = dds_bytes(10 * BL, fs=2400000)
    out["dds/iq_hash_first4k"] = np.array([hash16(dds[:4096].view(np.int16))], dtype=np.uint64)
    for name, cfg in DDS_CONFIGS.items():
        pcm, lens = refbind.RefStream(**cfg).run(dds, BL)
        out[name + "/lens"] = lens.astype(np.int32)
        out[name + "/hash"] = np.array([hash16(pcm)], dtype=np.uint64)
        out[name + "/pcm_first"] = pcm[:256].copy()
        out[name + "/pcm_last"] = pcm[-256:].copy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d arrays, %d bytes" % (path, len(out), os.path.getsize(path)))


if __name__ == "__main__":
    main()
