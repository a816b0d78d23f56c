#!/usr/bin/env python3
"""Diagnostic (GPU box): WHAT deviates when a neighbour wave issues dense matrix instructions (tools/diag/coburst.hip)?
Stage taps (decimated samples y, discriminator output v, resampler output mpx) and PCM of one launch with the neighbour
against the same launch without it.   python tools/diag/coburst_taps.py <math code> <mode> <kind> [reps]"""
import ctypes, os, subprocess, sys, time
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import torch
import rtl_fm_player_amd as R
from oracle import lcg_bytes
math, mode, kind = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
so = os.path.join(here, "libcoburst.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "coburst.hip")], check=True)
co = ctypes.CDLL(so)
BL, S, B = 262144, int(os.environ.get("CB_S", "256")), int(os.environ.get("CB_B", "16"))
M = BL // 16
dev = torch.device("cuda:0")
one = torch.from_numpy(lcg_bytes(B * BL, 2024)[0]).to(dev).view(1, B * BL)
iq = one.expand(S, B * BL).contiguous()
b = R.BatchDemod(R.wbfm_config(math=math, rate_in=300000, rate_out2=48000, mode=mode), S)

def run():
    pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
    lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
    y = torch.zeros((S, B * 2 * M), dtype=torch.float32, device=dev)
    v = torch.zeros((S, B * M), dtype=torch.float32, device=dev)
    mpx = torch.zeros((S, B * M), dtype=torch.float32, device=dev)
    torch.cuda.current_stream().synchronize()   # the fills above run on torch's stream, the kernel on the batch's own (a device-wide sync would wait for the neighbour)
    b.reset()
    if os.environ.get("CB_NOTAPS"):
        b.run_device(iq, B, pcm, lens); b.sync()
    else:
        b.run_device(iq, B, pcm, lens, debug={"y": y, "v": v, "mpx": mpx}); b.sync()
    return {"y": y.cpu().numpy(), "v": v.cpu().numpy(), "mpx": mpx.cpu().numpy(), "pcm": pcm.cpu().numpy().reshape(S, -1)}

clean = run()
again = run()
print("family", b.math, "mode", mode, "| clean run repeated: differing",
      {k: int((clean[k].view(np.uint32 if clean[k].dtype == np.float32 else np.int16) != again[k].view(np.uint32 if again[k].dtype == np.float32 else np.int16)).sum()) for k in clean})
for rep in range(reps):
    assert co.coburst_start(kind, 256, 0) == 0
    time.sleep(0.05)
    got = run()
    assert co.coburst_stop() == 0
    print("rep", rep, "neighbour kind", kind)
    for k in ("y", "v", "mpx", "pcm"):
        a, r = got[k], clean[k]
        if a.dtype == np.float32:
            d = a.view(np.uint32) != r.view(np.uint32)
        else:
            d = a != r
        n = int(d.sum())
        print("  tap", k, "differing values", n, "of", d.size, "in", int(d.any(axis=1).sum()), "streams")
        if n and k == "pcm":
            s_, i_ = np.nonzero(d)
            dd = a[s_, i_].astype(np.int64) - r[s_, i_].astype(np.int64)
            blk = i_ // (a.shape[1] // B)
            print("    pcm errors: |d| max", int(np.abs(dd).max()), "hist |d|<=1:", int((np.abs(dd) <= 1).sum()), "<=16:", int((np.abs(dd) <= 16).sum()), "all:", dd.size,
                  "| per stream min/median/max", np.bincount(s_, minlength=S).min(), int(np.median(np.bincount(s_, minlength=S))), np.bincount(s_, minlength=S).max(),
                  "| by block", np.bincount(blk, minlength=B).tolist())
            for j in range(min(6, n)):
                print("    stream", s_[j], "pcm idx", i_[j], "block", blk[j], "got", a[s_[j], i_[j]], "want", r[s_[j], i_[j]])
            # runs of consecutive wrong values in the first bad stream
            ii = i_[s_ == s_[0]]
            print("    first bad stream", s_[0], "wrong idx range", ii.min(), ii.max(), "count", ii.size, "first 12 idx", ii[:12].tolist())
        if n == 0 or k in ("mpx", "pcm"):
            continue
        s, idx = np.nonzero(d)
        per = 2 if k == "y" else 1
        samp = idx // per
        lanes = (samp % 512) // 8
        hist = np.bincount(lanes, minlength=64)
        print("    by lane group: 0-15", int(hist[:16].sum()), "16-31", int(hist[16:32].sum()), "32-47", int(hist[32:48].sum()), "48-63", int(hist[48:].sum()),
              "| by output r of the lane", np.bincount(samp % 8, minlength=8).tolist())
        for j in range(min(6, n)):
            g, w = a[s[j], idx[j]], r[s[j], idx[j]]
            print("    stream", s[j], "sample", samp[j], "tile", samp[j] // 512, "lane", lanes[j], "r", samp[j] % 8, "got %r (%08x) want %r (%08x) ulps %d" % (
                float(g), g.view(np.uint32), float(w), w.view(np.uint32), int(g.view(np.int32)) - int(w.view(np.int32))))
        if k == "v":
            # were the inputs of the wrong discriminator samples right?
            yy = got["y"].view(np.uint32).reshape(S, -1, 2); ry = clean["y"].view(np.uint32).reshape(S, -1, 2)
            same_in = sum(int((yy[s[j], samp[j]] == ry[s[j], samp[j]]).all() and (samp[j] == 0 or (yy[s[j], samp[j] - 1] == ry[s[j], samp[j] - 1]).all())) for j in range(min(n, 2000)))
            print("    of the first", min(n, 2000), "wrong v:", same_in, "have bit-identical y[n], y[n-1]")
            dd = np.abs(a[s, idx].astype(np.float64) - r[s, idx].astype(np.float64))
            print("    |error| max %.3g median %.3g; |ulps| <= 2: %d" % (dd.max(), np.median(dd), int((np.abs(a[s, idx].view(np.int32).astype(np.int64) - r[s, idx].view(np.int32).astype(np.int64)) <= 2).sum())))
