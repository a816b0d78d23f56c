#!/usr/bin/env python3
"""bench.py's parity gate for every rank of an 8-GPU run, on one GPU: same generator (SWEEP_DATA=fm|noise), seeds
12345 + rank, streams 0 and (7 rank + S/3) % S, all three modes, fast math.  Prints the worst
|diff| per (mode, rank); anything above 1 would abort that rank's bench."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_fm_player_amd as R
from oracle import OracleStream

BL = 262144
S, B = 256, 16
dev = torch.device("cuda:0")
MODES = {"stereo": dict(rate_in=300000, rate_out2=48000, mode=2), "mono": dict(rate_in=300000, rate_out2=48000, mode=1),
         "nfm": dict(rate_in=25000, rate_out2=12500, mode=1)}
ranks = range(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
worst_all = 0
for name, kw in MODES.items():
    b = R.BatchDemod(R.wbfm_config(block_len=BL, math=R.MATH_FAST, **kw), S, device=0)
    pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
    lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
    for rank in ranks:
        if os.environ.get("SWEEP_DATA", "fm") == "noise":
            g = torch.Generator(device=dev)
            g.manual_seed(12345 + rank)
            iq = torch.randint(0, 256, (S, B, BL), dtype=torch.uint8, device=dev, generator=g)
        else:                                          # bench.py's default input
            import bench
            iq = bench.synth_fm_iq(torch, dev, S, B * BL // 2, 200e3 if name == "nfm" else 2.4e6, name != "nfm",
                                   12345 + rank).view(S, B, BL)
        torch.cuda.synchronize()          # iq is made on torch's stream, the kernel runs on the batch's own
        b.reset()
        b.run_device(iq, B, pcm, lens)
        b.sync()
        worst = 0
        for s in sorted({0, (7 * rank + S // 3) % S}):
            want, wl = OracleStream(**kw).run(iq[s].cpu().numpy().reshape(-1), BL)
            l = lens[s].cpu().numpy()
            assert np.array_equal(l, wl)
            p = pcm[s].cpu().numpy()
            got = np.concatenate([p[k, :l[k]] for k in range(B)])
            worst = max(worst, int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max()))
        worst_all = max(worst_all, worst)
        print(name, "rank", rank, "worst |diff|", worst, flush=True)
        del iq
print("worst overall", worst_all)
sys.exit(0 if worst_all <= 1 else 1)
