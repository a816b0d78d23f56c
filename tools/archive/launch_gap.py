"""What a launch costs beside its kernel: 400 back-to-back launches of 256 streams, wall clock incl. the final sync, with and without the
timing event pair.   python tools/diag/launch_gap.py   (GAP_BLOCKS=1,2,16)"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import rtl_fm_player_amd as R
import bench
BL = 262144
dev = torch.device("cuda:0")
S = 256
BS = [int(x) for x in os.environ.get('GAP_BLOCKS', '1,2,16').split(',')]
for mode in (2, 1):
    for B in BS:
        cfg = R.wbfm_config(rate_in=300000, rate_out2=48000, mode=mode, math=R.MATH_FAST)
        b = R.BatchDemod(cfg, S)
        iq = bench.synth_fm_iq(torch, dev, S, B * BL // 2, 2.4e6, mode == 2, 1234).view(S, B, BL)
        pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
        lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
        # REQUIRED: the batch launches on its own stream.  Without this the kernels write pcm / lens while torch's queued synth kernels still use
        # that memory as scratch (the caching allocator handed it out in torch's stream order) - a corrupted rocPRIM scan then faults
        torch.cuda.synchronize()
        for timing in (True, False):
            b.set_timing(timing)
            for _ in range(100): b.run_device(iq, B, pcm, lens)
            b.sync()
            N = 400
            t0 = time.perf_counter()
            for _ in range(N): b.run_device(iq, B, pcm, lens)
            t1 = time.perf_counter()
            b.sync()
            t2 = time.perf_counter()
            print("mode %d blocks %2d timing %-5s: %.4f ms per launch (host enqueue %.4f), %.4f ms per block" % (mode, B, timing, (t2 - t0) / N * 1e3, (t1 - t0) / N * 1e3, (t2 - t0) / N / B * 1e3), flush=True)
        b.close()
