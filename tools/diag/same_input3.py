#!/usr/bin/env python3
"""Diagnostic (GPU box): 256 streams, SAME input (FM broadcast or LCG noise), one family: streams deviating from the majority."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import rtl_fm_player_amd as R
from oracle import lcg_bytes
math = int(sys.argv[1]); data = sys.argv[2]
BL, S, B = 262144, 256, 16
dev = torch.device("cuda:0")
if data == "fm":
    one = bench.synth_fm_iq(torch, dev, 1, B * BL // 2, 2.4e6, True, 12345).view(1, B * BL)
else:
    one = torch.from_numpy(lcg_bytes(B * BL, 2024)[0]).to(dev).view(1, B * BL)
iq = one.expand(S, B * BL).contiguous()
b = R.BatchDemod(R.wbfm_config(math=math, rate_in=300000, rate_out2=48000, mode=2), S)
pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for rep in range(3):
    b.reset()
    b.run_device(iq, B, pcm, lens); b.sync()
    a = pcm.view(S, -1).cpu().numpy()
    ref = np.where(a[0] == a[1], a[0], a[2])
    bad = np.nonzero((a != ref).any(axis=1))[0]
    print(data, "family", b.math, "rep", rep, "streams deviating:", len(bad))
