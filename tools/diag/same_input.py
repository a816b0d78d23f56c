#!/usr/bin/env python3
"""Diagnostic (GPU box): 256 streams fed the SAME IQ through one kernel family; which streams differ from stream 0, where?
   python tools/diag/same_input.py <math code> [reps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rtl_fm_player_amd as R
from oracle import lcg_bytes
math = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
BL, S, B = 262144, 256, 16
dev = torch.device("cuda:0")
base = lcg_bytes(B * BL, 2024)[0]
iq = torch.empty((S, B * BL), dtype=torch.uint8, device=dev)
iq[:] = torch.from_numpy(base).to(dev)
b = R.BatchDemod(R.wbfm_config(math=math, rate_in=300000, rate_out2=48000, mode=2), S)
pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for rep in range(reps):
    b.reset(); pcm.zero_()
    b.run_device(iq, B, pcm, lens); b.sync()
    p = pcm.cpu().numpy().astype(np.int32)
    d = np.abs(p - p[0:1])
    bad = np.nonzero(d.reshape(S, -1).max(axis=1))[0]
    print("rep", rep, "family", b.math, "streams that differ from stream 0:", len(bad), bad[:20])
    T = 32 * B
    for s in bad[:6]:
        blk, pos = np.nonzero(d[s])
        fr = pos // 2
        samp = blk * 16384 + fr * 300000 // 48000
        print("   stream", s, "max", d[s].max(), "blocks", sorted(set(blk.tolist()))[:8], "tiles", sorted(set((samp // 512).tolist()))[:12],
              "chunk starts", [c * T // 12 for c in range(13)])
