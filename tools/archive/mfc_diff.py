#!/usr/bin/env python3
"""Diagnostic (GPU box): where does the int8 stage C (FMD_MATH_FAST_MFMA_C) leave the +-1 LSB band?  One stream, LCG noise,
resampler-output tap against the exact family's.   python tools/diag/mfc_diff.py <rate_in> [time_split] [blocks]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rtl_fm_player_amd as R
from oracle import OracleStream, lcg_bytes
rate_in = int(sys.argv[1]); split = int(sys.argv[2]) if len(sys.argv) > 2 else 0; B = int(sys.argv[3]) if len(sys.argv) > 3 else 40
BL = 262144; M = BL // 16
kw = dict(rate_in=rate_in, rate_out2=48000, mode=2)
dev = torch.device("cuda:0")
h_iq = lcg_bytes(B * BL, 12345)[0]
iq = torch.from_numpy(h_iq).to(dev)
res = {}
for name, math in (("exact", R.MATH_EXACT), ("mfma", R.MATH_FAST_MFMA), ("mfma_c", R.MATH_FAST_MFMA_C)):
    b = R.BatchDemod(R.wbfm_config(block_len=BL, math=math, **kw), 1, device=0)
    b.set_time_split(split)
    pcm = torch.zeros(B * b.pcm_stride, dtype=torch.int16, device=dev)
    lens = torch.zeros(B, dtype=torch.int32, device=dev)
    v = torch.zeros(B * M, dtype=torch.float32, device=dev)
    mpx = torch.zeros(B * M, dtype=torch.float32, device=dev)
    b.run_device(iq, B, pcm, lens, debug={"v": v, "mpx": mpx}); b.sync(); torch.cuda.synchronize()
    l = lens.cpu().numpy(); p = pcm.cpu().numpy().reshape(B, -1)
    res[name] = (np.concatenate([p[k, :l[k]] for k in range(B)]), v.cpu().numpy(), mpx.cpu().numpy().reshape(B, M), l, b.math)
    b.close()
want, wl = OracleStream(**kw).run(h_iq, BL)
for name in res:
    d = np.abs(res[name][0].astype(np.int32) - want.astype(np.int32))
    print(name, "(runs family %d)" % res[name][4], "max |pcm diff| vs oracle", d.max(), "count>1", int((d > 1).sum()), "first", np.flatnonzero(d > 1)[:8].tolist())
me = res["exact"][2]
for name in ("mfma", "mfma_c"):
    md = np.abs(res[name][2] - me)
    print(name, "resampler output: max |diff| vs exact %.3g" % md.max(), "values off by > 3e-5:", int((md > 3e-5).sum()))
    blk, idx = np.nonzero(md > 3e-5)
    for k in range(min(12, blk.size)):
        fr = idx[k] // 2
        n_in = fr * rate_in // 48000
        print("   block", blk[k], "value", idx[k], "frame", fr, "rate_in sample ~", n_in, "tile", n_in // 512, "in-tile", n_in % 512, "got %.6g want %.6g" % (res[name][2][blk[k], idx[k]], me[blk[k], idx[k]]))
vd = np.abs(res["mfma_c"][1] - res["mfma"][1]); print("v: mfma_c vs mfma max diff", vd.max())
