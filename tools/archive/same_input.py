#!/usr/bin/env python3
"""Diagnostic (GPU box): S streams fed the SAME IQ through one kernel family - every stream must then produce the same PCM.
Which streams deviate from the majority, and (with --taps) at which stage (decimated IQ y, discriminator v, resampler output mpx)?
One script for what used to be same_input{,2,3,4}.py (round 3/4: the packed-fp32 hazard beside MFMA neighbours).

   python tools/diag/same_input.py <math code> [--data lcg|fm] [--taps] [--reps N] [--streams S]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rtl_fm_player_amd as R
from oracle import lcg_bytes

ap = argparse.ArgumentParser()
ap.add_argument("math", type=int)
ap.add_argument("--data", choices=["lcg", "fm"], default="lcg")
ap.add_argument("--taps", action="store_true", help="compare the stage taps too (runs the debug build of the kernel)")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--streams", type=int, default=256)
a = ap.parse_args()
BL, S, B = 262144, a.streams, 16
M = BL // 16
dev = torch.device("cuda:0")
if a.data == "fm":
    import bench
    one = bench.synth_fm_iq(torch, dev, 1, B * BL // 2, 2.4e6, True, 12345).view(1, B * BL)
else:
    one = torch.from_numpy(lcg_bytes(B * BL, 2024)[0]).to(dev).view(1, B * BL)
iq = one.expand(S, B * BL).contiguous()
b = R.BatchDemod(R.wbfm_config(math=a.math, rate_in=300000, rate_out2=48000, mode=2), S)
pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
dbg = None
if a.taps:
    dbg = {"y": torch.zeros((S, B * 2 * M), dtype=torch.float32, device=dev),
           "v": torch.zeros((S, B * M), dtype=torch.float32, device=dev),
           "mpx": torch.zeros((S, B * M), dtype=torch.float32, device=dev)}
torch.cuda.synchronize()
T = 32 * B
for rep in range(a.reps):
    b.reset(); pcm.zero_()
    if dbg: b.run_device(iq, B, pcm, lens, debug=dbg)
    else: b.run_device(iq, B, pcm, lens)
    b.sync()
    p = pcm.cpu().numpy().reshape(S, -1)
    ref = np.where(p[0] == p[1], p[0], p[2]) if S >= 3 else p[0]       # majority of the first three streams
    d = p != ref
    bad = np.nonzero(d.any(axis=1))[0]
    print("rep", rep, "family", b.math, "data", a.data, "streams deviating from the majority:", len(bad), bad[:20].tolist())
    for s in bad[:6]:
        pos = np.nonzero(d[s])[0]
        blk, fr = pos // b.pcm_stride, (pos % b.pcm_stride) // 2
        samp = blk * M + fr * 300000 // 48000
        print("   stream", s, "max |diff|", int(np.abs(p[s].astype(np.int32) - ref.astype(np.int32)).max()), "blocks", sorted(set(blk.tolist()))[:8],
              "tiles", sorted(set((samp // 512).tolist()))[:12])
    if dbg:
        for name, t in dbg.items():
            x = t.cpu().numpy()
            rf = np.median(x.astype(np.float64), axis=0).astype(np.float32)
            dd = x != rf
            bb = np.nonzero(dd.any(axis=1))[0]
            print("   tap", name, "streams with a deviation:", len(bb), "deviating values:", int(dd.sum()))
            for s in bb[:3]:
                idx = np.nonzero(dd[s])[0]
                per = 2 if name == "y" else 1
                sm = idx // per
                print("      stream", s, "tile", int(sm[0] // 512), "lanes", ((sm // 8) % 64)[:8].tolist(), "r", (sm % 8)[:8].tolist(), "n", idx.size,
                      "got", x[s, idx][:4], "want", rf[idx][:4])
