#!/usr/bin/env python3
"""Soak (GPU box): 256 streams fed the SAME IQ; every launch, every stream's PCM must equal stream 0's (compared on the
device).  python tools/diag/determinism.py <math code> <launches> [mode 2|1]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rtl_fm_player_amd as R
from oracle import lcg_bytes
math, n = int(sys.argv[1]), int(sys.argv[2]); mode = int(sys.argv[3]) if len(sys.argv) > 3 else 2
BL, S, B = 262144, 256, 16
dev = torch.device("cuda:0")
one = torch.from_numpy(lcg_bytes(B * BL, 2024)[0]).to(dev).view(1, B * BL)
iq = one.expand(S, B * BL).contiguous()
b = R.BatchDemod(R.wbfm_config(math=math, rate_in=300000, rate_out2=48000, mode=mode), S)
pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
bad_launches = bad_streams = 0
first = None
for rep in range(n):
    b.reset()
    b.run_device(iq, B, pcm, lens); b.sync()
    p = pcm.view(S, -1)
    if first is None:
        first = p[0].clone()
    dev_streams = int((p != first.unsqueeze(0)).any(dim=1).sum().item())
    bad_launches += dev_streams > 0
    bad_streams += dev_streams
print("family", b.math, "mode", mode, "launches", n, "launches with a deviating stream:", bad_launches, "deviating stream-launches:", bad_streams, "of", n * S)
