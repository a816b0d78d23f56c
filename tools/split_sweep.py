#!/usr/bin/env python3
"""Kernel time against the number of time chunks per stream (fmd_batch_set_time_split) for short launches.
   python tools/split_sweep.py [blocks per launch] [mode stereo|mono]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtl_fm_player_amd as R
import bench
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = sys.argv[2] if len(sys.argv) > 2 else "stereo"
BL, S = 262144, int(os.environ.get("SWEEP_STREAMS", "256"))
dev = torch.device("cuda:0")
cfg = R.wbfm_config(block_len=BL, math=R.MATH_FAST, rate_in=300000, rate_out2=48000, mode=2 if mode == "stereo" else 1)
iq = bench.synth_fm_iq(torch, dev, S, blocks * BL // 2, 2.4e6, True, 12345).view(S, blocks, BL)
for split in [int(x) for x in os.environ.get("SWEEP_SPLITS", "0,2,3,4,6,8,12,16,24").split(",")]:
    b = R.BatchDemod(cfg, S, device=0)
    b.set_time_split(split)
    pcm = torch.zeros((S, blocks, b.pcm_stride), dtype=torch.int16, device=dev)
    lens = torch.zeros((S, blocks), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for _ in range(100):
        b.run_device(iq, blocks, pcm, lens)
    b.sync()
    ms = []
    for _ in range(50):
        b.run_device(iq, blocks, pcm, lens)
        ms.append(b.last_kernel_ms())
    ms.sort()
    print("blocks", blocks, mode, "time split", split if split else "auto", "kernel ms median %.4f min %.4f" % (ms[len(ms) // 2], ms[0]))
    b.close()
