#!/usr/bin/env python3
"""GPU box: kernel ms per launch on the constructed inputs of tests/test_gpu_adversarial.py next to the FM input (stereo, mono).
   python tools/adversarial_time.py [streams] [blocks]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
import rtl_fm_player_amd as R
from test_gpu_adversarial import make, CUT, PILOTLESS

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
BL = 262144
dev = torch.device("cuda:0")
for mode, kw in (("stereo", dict(rate_in=300000, rate_out2=48000, mode=2)), ("mono", dict(rate_in=300000, rate_out2=48000, mode=1))):
    inputs = {"fm": bench.synth_fm_iq(torch, dev, S, B * BL // 2, 2.4e6, True, 12345).view(S, B, BL)}
    for kind in CUT + PILOTLESS:
        one = torch.from_numpy(make(kind, B * BL)).to(dev).view(1, B, BL)
        inputs[kind] = one.expand(S, B, BL).contiguous()
    for fam, code in (("fast", R.MATH_FAST), ("exact", R.MATH_EXACT)):
        b = R.BatchDemod(R.wbfm_config(block_len=BL, math=code, **kw), S)
        pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
        lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        row = {}
        for name, iq in inputs.items():
            for _ in range(3):
                b.run_device(iq, B, pcm, lens)
            b.sync()
            ms = []
            for _ in range(5):
                b.run_device(iq, B, pcm, lens); b.sync()
                ms.append(b.last_kernel_ms())
            row[name] = round(sorted(ms)[2], 4)
        print(json.dumps({"mode": mode, "family": fam, "resolved": b.math, "vs_fm": {k: round(v / row["fm"], 2) for k, v in row.items()}, "kernel_ms": row}), flush=True)
        del b
