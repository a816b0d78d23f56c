#!/usr/bin/env python3
"""GPU box: the synthetic FM broadcast of bench.py (stereo multiplex with pilot, +-75 kHz) x volume, every +-1 LSB family against the oracle: the
signal the headline is measured on, at the volumes around the composite L+R filter's gate (FMD_MATH_FAST_MFMA_F runs up to volume ~8 at 300 kHz).
Prints per volume: the family each name resolved to, the worst |PCM difference| and how many values differ.   python tools/fm_volume_scan.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import rtl_fm_player_amd as R
import bench
from oracle import OracleStream

BL, NB, NS = 262144, 4, 16
dev = torch.device("cuda:0")
iq = bench.synth_fm_iq(torch, dev, NS, NB * BL // 2, 2400000.0, True, 4242).cpu().numpy().reshape(NS, NB, BL)
FAMS = {"valu": R.MATH_FAST_VALU, "mfma": R.MATH_FAST_MFMA, "mfma_f": R.MATH_FAST_MFMA_F}
worst_all = 0
for rate_in in (300000, 240000):
    for vol in (0.4, 1.0, 2.0, 2.5, 3.0, 3.5, 5.0, 8.0):
        kw = dict(rate_in=rate_in, rate_out2=48000, mode=2, volume=vol)
        want = [OracleStream(**kw).run(iq[s].reshape(-1), BL)[0] for s in range(NS)]
        row = {}
        for fname, code in FAMS.items():
            b = R.BatchDemod(R.wbfm_config(block_len=BL, math=code, **kw), NS)
            got, lens = b.run_host_concat(iq, NB)
            d = max(int(np.abs(got[s].astype(np.int32) - want[s].astype(np.int32)).max()) for s in range(NS))
            nz = sum(int((got[s] != want[s]).sum()) for s in range(NS))
            row[fname] = (b.math, d, nz)
            worst_all = max(worst_all, d)
            b.close()
        print(json.dumps({"rate_in": rate_in, "volume": vol, "values": int(sum(w.size for w in want)), "family_run_maxdiff_count": row}), flush=True)
print("worst", worst_all)
