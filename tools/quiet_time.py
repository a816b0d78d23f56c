#!/usr/bin/env python3
"""GPU box: kernel ms per launch for quiet inputs next to the FM input, per kernel family and mode.
   python tools/quiet_time.py [streams] [blocks]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import rtl_fm_player_amd as R

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
BL = 262144
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(99)
MODES = {"stereo": dict(rate_in=300000, rate_out2=48000, mode=2), "mono": dict(rate_in=300000, rate_out2=48000, mode=1),
         "nfm": dict(rate_in=25000, rate_out2=12500, mode=1)}
FAMS = {"fast": R.MATH_FAST, "fast-valu": R.MATH_FAST_VALU, "exact": R.MATH_EXACT}
for mode, kw in MODES.items():
    inputs = {
        "fm": bench.synth_fm_iq(torch, dev, S, B * BL // 2, 200e3 if mode == "nfm" else 2.4e6, mode != "nfm", 12345).view(S, B, BL),
        "noise": torch.randint(0, 256, (S, B, BL), dtype=torch.uint8, device=dev, generator=g),
        "all127": torch.full((S, B, BL), 127, dtype=torch.uint8, device=dev),
        "127or128": torch.randint(127, 129, (S, B, BL), dtype=torch.uint8, device=dev, generator=g),
        "126to129": torch.randint(126, 130, (S, B, BL), dtype=torch.uint8, device=dev, generator=g),
    }
    for fam, code in FAMS.items():
        b = R.BatchDemod(R.wbfm_config(block_len=BL, math=code, **kw), S)
        pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
        lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
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
        print(json.dumps({"mode": mode, "family": fam, "resolved": b.math, "kernel_ms": row,
                          "vs_fm": {k: round(v / row["fm"], 2) for k, v in row.items()}}), flush=True)
        del b
