#!/usr/bin/env python3
"""Per-stage cycle shares of the fused kernel (fmd_debug_taps.prof), for tuning.

    python tools/stage_profile.py [--streams 256] [--blocks 16] [--math fast] [--mode stereo]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = ["load", "decimate", "discriminate", "q1", "mpx", "flush_fast", "resample", "roll", "flush", "state"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=16)
    ap.add_argument("--math", default="fast")
    ap.add_argument("--mode", default="stereo", choices=["stereo", "mono", "nfm"])
    ap.add_argument("--data", default="fm")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--volume", type=float, default=0.4)
    a = ap.parse_args()
    import torch
    import rtl_fm_player_amd as R
    BL = 262144
    dev = torch.device("cuda:0")
    fam = {"fast": R.MATH_FAST, "exact": R.MATH_EXACT, "fast-valu": R.MATH_FAST_VALU, "fast-mfma": R.MATH_FAST_MFMA,
           "fast-mfma-f": R.MATH_FAST_MFMA_F}[a.math]
    cfg = R.wbfm_config(block_len=BL, math=fam, volume=a.volume,
                        **(dict(rate_in=25000, rate_out2=12500, mode=1) if a.mode == "nfm" else
                           dict(rate_in=300000, rate_out2=48000, mode=2 if a.mode == "stereo" else 1)))
    b = R.BatchDemod(cfg, a.streams, device=0)
    import bench
    if a.data == "noise":
        g = torch.Generator(device=dev); g.manual_seed(12345)
        iq = torch.randint(0, 256, (a.streams, a.blocks, BL), dtype=torch.uint8, device=dev, generator=g)
    else:
        iq = bench.synth_fm_iq(torch, dev, a.streams, a.blocks * BL // 2, 200e3 if a.mode == "nfm" else 2.4e6, a.mode != "nfm", 12345).view(a.streams, a.blocks, BL)
    pcm = torch.zeros((a.streams, a.blocks, b.pcm_stride), dtype=torch.int16, device=dev)
    lens = torch.zeros((a.streams, a.blocks), dtype=torch.int32, device=dev)
    prof = torch.zeros((a.streams * 64, 16), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for _ in range(60):                      # settle the clocks (tools/ramp_check.py)
        b.run_device(iq, a.blocks, pcm, lens)
    b.sync()
    ms = []
    for _ in range(a.reps):
        b.run_device(iq, a.blocks, pcm, lens)
        ms.append(b.last_kernel_ms())
    b.run_device(iq, a.blocks, pcm, lens, debug={"prof": prof})
    b.sync()
    ms_prof = b.last_kernel_ms()
    p = prof.cpu().numpy().astype(np.float64)
    p = p[p[:, 15] > 0]
    tot = p[:, 15].mean()
    samples = a.streams * a.blocks * BL // 2
    out = {"kernel_ms_min": round(min(ms), 4), "kernel_ms_median": round(float(np.median(ms)), 4), "kernel_ms_with_stamps": round(ms_prof, 4),
           "Gsamples_per_s": round(samples / (min(ms) * 1e-3) / 1e9, 1),
           "cycles_total_mean": tot, "clock_GHz_est": round(tot / (ms_prof * 1e-3) / 1e9, 3),
           "share": {n: round(float(p[:, i].mean() / tot), 4) for i, n in enumerate(NAMES)},
           "redo_prof(-DFMD_REDO_PROF builds: count, cycles step1, step2, step3, rest; per worker)": [round(float(p[:, 10 + i].mean()), 1) for i in range(5)],
           "workers": int(p.shape[0])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
