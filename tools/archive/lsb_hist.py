#!/usr/bin/env python3
"""How far the +-1 LSB families sit from the exact kernels (= the reference, bit for bit): share of PCM values that differ by 0 / 1 / more,
per volume and input kind.   python tools/diag/lsb_hist.py [streams] [blocks]      (FMD_LIB_PATH / FMD_MFMA choose the build / family)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import torch
    import rtl_fm_player_amd as R
    import bench
    BL = 262144
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(777)
    inputs = {"fm": bench.synth_fm_iq(torch, dev, S, B * BL // 2, 2.4e6, True, 4242).view(S, B, BL),
              "noise": torch.randint(0, 256, (S, B, BL), dtype=torch.uint8, device=dev, generator=g)}
    for vol in (0.4, 1.0, 3.0, 8.0):
        for kind, iq in inputs.items():
            out = {}
            for name, math in (("exact", R.MATH_EXACT), ("fast", R.MATH_FAST)):
                b = R.BatchDemod(R.wbfm_config(rate_in=300000, rate_out2=48000, mode=2, volume=vol, math=math), S)
                pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
                lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
                b.run_device(iq, B, pcm, lens); b.sync()
                out[name] = (pcm.cpu().numpy().astype(np.int32), lens.cpu().numpy(), b.math)
                b.close()
            (pe, le, _), (pf, lf, fam) = out["exact"], out["fast"]
            assert np.array_equal(le, lf)
            n = int(le[0, 0])
            d = np.abs(pe[:, :, :n] - pf[:, :, :n])
            clipped = float((np.abs(pe[:, :, :n]) >= 32767).mean())
            print("family %d volume %.1f %-5s values %d | diff 0: %.5f  1: %.6f  >1: %d (max %d) | clipped %.3f" % (
                fam, vol, kind, d.size, float((d == 0).mean()), float((d == 1).mean()), int((d > 1).sum()), int(d.max()), clipped), flush=True)


if __name__ == "__main__":
    main()
