#!/usr/bin/env python3
"""GPU box: one block per launch - plain launches against a replayed hipGraph of N captured launches (torch.cuda.CUDAGraph on the caller's stream).
   python tools/diag/graph_time.py [streams] [launches per graph]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import rtl_fm_player_amd as R
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
BL = 262144
dev = torch.device("cuda:0")
for mode, kw in (("stereo", dict(rate_in=300000, rate_out2=48000, mode=2)), ("mono", dict(rate_in=300000, rate_out2=48000, mode=1))):
    b = R.BatchDemod(R.wbfm_config(block_len=BL, math=R.MATH_FAST, **kw), S)
    b.set_timing(False)
    iq = bench.synth_fm_iq(torch, dev, S, N * BL // 2, 2.4e6, True, 12345).view(S, N, BL)
    blocks = [iq[:, k:k + 1].contiguous() for k in range(N)]
    pcm = [torch.zeros((S, 1, b.pcm_stride), dtype=torch.int16, device=dev) for _ in range(N)]
    lens = [torch.zeros((S, 1), dtype=torch.int32, device=dev) for _ in range(N)]
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    def plain(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(st)
        for _ in range(reps):
            for k in range(N):
                b.run_device(blocks[k], 1, pcm[k], lens[k], hip_stream=st.cuda_stream)
        e1.record(st); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * N), (time.perf_counter() - t0) * 1e3 / (reps * N)
    plain(5)
    p_gpu, p_wall = plain(20)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for k in range(N):
            b.run_device(blocks[k], 1, pcm[k], lens[k], hip_stream=st.cuda_stream)
    def graph(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(st):
            e0.record(st)
            for _ in range(reps):
                g.replay()
            e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * N), (time.perf_counter() - t0) * 1e3 / (reps * N)
    graph(5)
    g_gpu, g_wall = graph(20)
    print("mode %s, %d streams, one block per launch: plain %.4f ms per launch on the stream (%.4f wall), graph of %d launches %.4f ms per launch (%.4f wall)"
          % (mode, S, p_gpu, p_wall, N, g_gpu, g_wall))
