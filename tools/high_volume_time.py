#!/usr/bin/env python3
"""GPU box: what the volume-scaled origin threshold of the +-1 LSB kernels costs, and where their contract ends.
fmdk_params.org_thr grows with coef x (largest tap behind the discriminator) / 7600, up to 200 x 1e-3: at high volume on weak input most tiles then run
stages A and B in the reference's arithmetic.  Prints per mode and volume: the family FMD_MATH_FAST resolves to, kernel ms on a weak signal (a carrier of
20 LSB amplitude in +-2 LSB of noise) next to the FM broadcast of bench.py, and max |PCM difference| to the oracle on 8 streams of the weak input.
   python tools/high_volume_time.py [streams] [blocks]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import rtl_fm_player_amd as R
from oracle import OracleStream

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
BL = 262144
dev = torch.device("cuda:0")
rng = np.random.default_rng(7)
n = B * BL // 2
ph = 2 * np.pi * (-0.25) * np.arange(n) + 0.3 * np.sin(2 * np.pi * 1e-4 * np.arange(n))      # carrier at -fs/4 (rotate_90 centres it), slow phase wobble
weak = np.empty((8, 2 * n), dtype=np.uint8)
for s in range(8):
    i = 127.5 + 20.0 * np.cos(ph + s) + rng.normal(0, 2.0, n)
    q = 127.5 + 20.0 * np.sin(ph + s) + rng.normal(0, 2.0, n)
    weak[s, 0::2] = np.clip(np.round(i), 0, 255)
    weak[s, 1::2] = np.clip(np.round(q), 0, 255)
weak_t = torch.from_numpy(weak).to(dev).view(8, B, BL).repeat(S // 8, 1, 1).contiguous()
for mode, kw, rate in (("stereo", dict(rate_in=300000, rate_out2=48000, mode=2), 2.4e6), ("mono", dict(rate_in=300000, rate_out2=48000, mode=1), 2.4e6),
                       ("nfm", dict(rate_in=25000, rate_out2=12500, mode=1), 200e3)):
    fm = bench.synth_fm_iq(torch, dev, S, n, rate, mode != "nfm", 12345).view(S, B, BL)
    for vol in (0.4, 8.0, 32.0, 100.0):
        b = R.BatchDemod(R.wbfm_config(block_len=BL, math=R.MATH_FAST, volume=vol, **kw), S)
        pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
        lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        row = {"mode": mode, "volume": vol, "family": b.math}
        for name, iq in (("fm", fm), ("weak", weak_t)):
            b.reset()
            for _ in range(3):
                b.run_device(iq, B, pcm, lens)
            b.sync()
            ms = []
            for _ in range(10):
                b.run_device(iq, B, pcm, lens)
                b.sync()
                ms.append(b.last_kernel_ms())
            row[name + "_kernel_ms"] = round(float(np.median(ms)), 4)
        b.reset()
        b.run_device(weak_t, B, pcm, lens)
        b.sync()
        hp, hl = pcm.cpu().numpy(), lens.cpu().numpy()
        worst = 0
        for s in range(8):
            want, wl = OracleStream(volume=vol, **kw).run(weak[s], BL)
            got = np.concatenate([hp[s, k, :wl[k]] for k in range(B)])
            worst = max(worst, int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max()))
        row["weak_max_abs_lsb"] = worst
        row["weak_slowdown"] = round(row["weak_kernel_ms"] / row["fm_kernel_ms"], 2)
        print(json.dumps(row), flush=True)
        b.close()
