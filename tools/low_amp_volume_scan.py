#!/usr/bin/env python3
"""GPU box: low-amplitude inputs x volume: the worst |PCM difference| of every fast family against the oracle.
Decimated samples just ABOVE the origin threshold of the fast discriminator are the worst case of its error budget:
their phase error is (decimator difference) / magnitude, and the PCM step shrinks with the volume.   python tools/low_amp_volume_scan.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rtl_fm_player_amd as R
from oracle import OracleStream

BL = 262144
NB, NS = 4, 8
MODES = {"stereo": dict(rate_in=300000, rate_out2=48000, mode=2), "mono": dict(rate_in=300000, rate_out2=48000, mode=1),
         "nfm": dict(rate_in=25000, rate_out2=12500, mode=1)}
FAMS = {"valu": R.MATH_FAST_VALU, "mfma": R.MATH_FAST_MFMA, "mfma_f": R.MATH_FAST_MFMA_F}
worst_all = 0
AMPS = [int(a) for a in os.environ.get("SCAN_AMPS", "1,2,3,4,6,10,20").split(",")]
for amp in AMPS:
    rng = np.random.default_rng(1000 + amp)
    iq = rng.integers(max(0, 128 - amp), min(256, 128 + amp), NS * NB * BL, dtype=np.uint8).reshape(NS, NB, BL)
    for vol in (0.4, 1.0, 3.0, 8.0):
        for mname, kw in MODES.items():
            kw2 = dict(kw, volume=vol)
            want = [OracleStream(**kw2).run(iq[s].reshape(-1), BL)[0] for s in range(NS)]
            row = {}
            for fname, code in FAMS.items():
                b = R.BatchDemod(R.wbfm_config(block_len=BL, math=code, **kw2), NS)
                got, lens = b.run_host_concat(iq, NB)
                d = max(int(np.abs(got[s].astype(np.int32) - want[s].astype(np.int32)).max()) for s in range(NS))
                nz = sum(int((got[s] != want[s]).sum()) for s in range(NS))
                row[fname] = (d, nz)
                worst_all = max(worst_all, d)
                b.close()
            print(json.dumps({"amp": amp, "volume": vol, "mode": mname, "max_diff_and_count": row}), flush=True)
print("worst", worst_all)
