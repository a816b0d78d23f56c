#!/usr/bin/env python3
"""Diagnostic (GPU box): one stream of bench.py's generator through the three kernel families with stage taps; where does
the MFMA family leave the +-1 LSB band, and what do the decimator outputs look like there?
   python tools/diag/mfma_diff.py <mode> <rank> <stream>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import rtl_fm_player_amd as R
from oracle import OracleStream

mode, rank, stream = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
BL, S, B = 262144, 256, 16
kw = dict(rate_in=25000, rate_out2=12500, mode=1) if mode == "nfm" else \
    dict(rate_in=300000, rate_out2=48000, mode=2 if mode == "stereo" else 1)
dev = torch.device("cuda:0")
iq_all = bench.synth_fm_iq(torch, dev, S, B * BL // 2, 200e3 if mode == "nfm" else 2.4e6, mode != "nfm", 12345 + rank).view(S, B, BL)
torch.cuda.synchronize()
iq = iq_all[stream].contiguous()
h_iq = iq.cpu().numpy().reshape(-1)
M = BL // 16
res = {}
for name, math in (("exact", R.MATH_EXACT), ("valu", R.MATH_FAST_VALU), ("mfma", R.MATH_FAST_MFMA)):
    b = R.BatchDemod(R.wbfm_config(block_len=BL, math=math, **kw), 1, device=0)
    pcm = torch.zeros(B * b.pcm_stride, dtype=torch.int16, device=dev)
    lens = torch.zeros(B, dtype=torch.int32, device=dev)
    y = torch.zeros(B * 2 * M, dtype=torch.float32, device=dev)
    v = torch.zeros(B * M, dtype=torch.float32, device=dev)
    b.run_device(iq, B, pcm, lens, debug={"y": y, "v": v})
    b.sync(); torch.cuda.synchronize()
    l = lens.cpu().numpy()
    p = pcm.cpu().numpy().reshape(B, -1)
    res[name] = (np.concatenate([p[k, :l[k]] for k in range(B)]), y.cpu().numpy().reshape(-1, 2), v.cpu().numpy())
    b.close()
want, wl = OracleStream(**kw).run(h_iq, BL)
for name in ("exact", "valu", "mfma"):
    d = np.abs(res[name][0].astype(np.int32) - want.astype(np.int32))
    print(name, "max |pcm diff| vs oracle", d.max(), "at", int(d.argmax()), "count>1", int((d > 1).sum()))
ye, ve = res["exact"][1], res["exact"][2]
for name in ("valu", "mfma"):
    yd = np.abs(res[name][1] - ye).max(axis=1)
    vd = np.abs(res[name][2] - ve)
    print(name, "max |y - y_exact|", yd.max(), "at sample", int(yd.argmax()), "| max |v - v_exact|", vd.max(), "at", int(vd.argmax()))
    i = int(vd.argmax())
    for j in range(max(i - 3, 0), i + 3):
        print("   m", j, "tile", j // 512, "in-tile", j % 512, "y_exact", ye[j], "y", res[name][1][j], "|y|", float(np.hypot(*ye[j])), "v_exact", ve[j], "v", res[name][2][j])
