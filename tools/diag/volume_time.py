import sys, json, numpy as np, torch
sys.path.insert(0, '.')
import bench, rtl_fm_player_amd as R
S, B, BL = 256, 4, 262144
dev = torch.device("cuda:0")
kw = dict(rate_in=300000, rate_out2=48000, mode=2)
fm = bench.synth_fm_iq(torch, dev, S, B * BL // 2, 2.4e6, True, 12345).view(S, B, BL)
for fam, code in (("fast", R.MATH_FAST), ("mfma", R.MATH_FAST_MFMA), ("valu", R.MATH_FAST_VALU)):
    for vol in (0.4, 1.0, 2.0, 4.0, 5.0, 7.5, 8.0, 16.0):
        b = R.BatchDemod(R.wbfm_config(block_len=BL, math=code, volume=vol, **kw), S)
        pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev); lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for _ in range(3): b.run_device(fm, B, pcm, lens)
        b.sync(); ms = []
        for _ in range(8):
            b.run_device(fm, B, pcm, lens); b.sync(); ms.append(b.last_kernel_ms())
        print(fam, vol, "family", b.math, "kernel_ms", round(float(np.median(ms)), 4), flush=True)
        b.close()
