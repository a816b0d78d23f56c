#!/usr/bin/env python3
"""Diagnostic (GPU box, library built with -DFMD_DBG_Y_LATE): 256 streams, same input; the decimated samples as they are
AFTER stage B: where and how do they deviate?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rtl_fm_player_amd as R
from oracle import lcg_bytes
BL, S, B = 262144, 256, 16
M = BL // 16
dev = torch.device("cuda:0")
base = lcg_bytes(B * BL, 2024)[0]
iq = torch.empty((S, B * BL), dtype=torch.uint8, device=dev)
iq[:] = torch.from_numpy(base).to(dev)
b = R.BatchDemod(R.wbfm_config(math=5, rate_in=300000, rate_out2=48000, mode=2), S)
pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
y = torch.zeros((S, B * 2 * M), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
b.run_device(iq, B, pcm, lens, debug={"y": y}); b.sync()
a = y.cpu().numpy()
ref = np.median(a.astype(np.float64), axis=0).astype(np.float32)
d = a != ref
bad = np.nonzero(d.any(axis=1))[0]
print("streams with deviating late y:", len(bad))
for s in bad[:10]:
    idx = np.nonzero(d[s])[0]
    samp = idx // 2
    print("stream", s, "float idx", idx[:8], "sample r", (samp % 8)[:8], "lane", ((samp // 8) % 64)[:8], "comp", (idx % 2)[:8], "n", idx.size)
    print("   got ", a[s, idx][:8])
    print("   want", ref[idx][:8])
    # does the wrong value occur elsewhere in the reference (same tile)?
    t0 = (idx[0] // 2 // 512) * 512 * 2
    tile = ref[t0:t0 + 1024]
    for k in idx[:4]:
        w = np.nonzero(tile == a[s, k])[0]
        print("   value", a[s, k], "found in this tile's reference at float offsets", w[:6], "(own offset", k - t0, ")")
