#!/usr/bin/env python3
"""Soak (GPU box): the determinism soak of tools/diag/determinism.py with a NEIGHBOUR on every SIMD that only issues matrix
instructions (tools/diag/coburst.hip, its own stream, started before the product's launches).  Every stream of every launch
must equal the PCM of a launch made without the neighbour.
  python tools/diag/coburst.py <math code> <launches> <mode 2|1> <kind 0 bf16 | 1 i8 | 2 f32 | 3 valu ... 10, see coburst.hip> [blocks] [prio]
  kind 20 / 21: the neighbour is a LIBRARY GEMM instead - torch.matmul of two 8192 x 8192 matrices in bf16 (20) or fp32 (21), queued
  back to back on a side stream (hipBLASLt / rocBLAS kernels) while the product's launches run."""
import ctypes, os, subprocess, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import torch
import rtl_fm_player_amd as R
from oracle import lcg_bytes
math, n, mode, kind = (int(x) for x in sys.argv[1:5])
blocks = int(sys.argv[5]) if len(sys.argv) > 5 else 256
prio = int(sys.argv[6]) if len(sys.argv) > 6 else 0
so = os.path.join(here, "libcoburst.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "coburst.hip")], check=True)
co = ctypes.CDLL(so)
BL, S, B = 262144, 256, 16
dev = torch.device("cuda:0")
one = torch.from_numpy(lcg_bytes(B * BL, 2024)[0]).to(dev).view(1, B * BL)
iq = one.expand(S, B * BL).contiguous()
b = R.BatchDemod(R.wbfm_config(math=math, rate_in=300000, rate_out2=48000, mode=mode), S)
pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
torch.cuda.synchronize()

def launches(k):
    bad_l = bad_s = 0
    t0 = time.time()
    for rep in range(k):
        b.reset()
        b.run_device(iq, B, pcm, lens); b.sync()
        p = pcm.view(S, -1)
        d = int((p != first.unsqueeze(0)).any(dim=1).sum().item())
        if os.environ.get("CB_CPU_COMPARE"):         # the same comparison on the host, from a copy made by the runtime's DMA
            h = p.cpu().numpy()
            dc = int((h != first_h[None, :]).any(axis=1).sum())
            if dc != d:
                global disagree
                disagree += 1
                if disagree <= 3:
                    n_gpu = int((p != first.unsqueeze(0)).sum().item())
                    print("  launch", rep, ": device-side comparison says", d, "streams deviate (", n_gpu, "values ), host-side comparison of the same buffer says", dc)
            d = dc
        bad_l += d > 0; bad_s += d
    return bad_l, bad_s, (time.time() - t0) / k * 1e3

b.reset(); b.run_device(iq, B, pcm, lens); b.sync()
first = pcm.view(S, -1)[0].clone()
first_h = first.cpu().numpy()
disagree = 0
alone = launches(20)
tot_l = tot_s = done = 0
ms = []
side = torch.cuda.Stream() if kind >= 20 else None
if kind >= 20:
    dt = torch.bfloat16 if kind == 20 else torch.float32
    ga = torch.randn((8192, 8192), device=dev, dtype=torch.float32).to(dt); gb = torch.randn((8192, 8192), device=dev, dtype=torch.float32).to(dt)
    gc = torch.empty((8192, 8192), device=dev, dtype=dt)
    torch.matmul(ga, gb, out=gc); torch.cuda.synchronize()
    t0 = time.time(); torch.matmul(ga, gb, out=gc); torch.cuda.synchronize(); gemm_ms = (time.time() - t0) * 1e3
    print("library GEMM 8192^3", dt, "%.2f ms each alone" % gemm_ms)
while done < n:
    k = min(200 if kind < 20 else 20, n - done)   # a neighbour lives a few seconds at most: restart it every 200 launches
    if kind >= 20:
        with torch.cuda.stream(side):
            for _ in range(max(4, int(k * 20.0 / max(gemm_ms, 0.05)) + 4)):   # enough GEMMs to outlast k product launches at ~20 ms each
                torch.matmul(ga, gb, out=gc)
    else:
        rc = co.coburst_start(kind, blocks, prio)
        assert rc == 0, rc
    time.sleep(0.05)
    bl, bs, t = launches(k)
    if kind >= 20:
        busy = not side.query()
        side.synchronize()
        if not busy: print("  (the GEMMs had finished before the product's launches did)")
    else:
        rc = co.coburst_stop()
        assert rc == 0, rc
    tot_l += bl; tot_s += bs; done += k; ms.append(t)
print("family", b.math, "mode", mode, "neighbour kind", kind, "blocks", blocks, "prio", prio, "| alone: deviating", alone[1], "ms/launch(host)", round(alone[2], 3),
      "| with neighbour: launches", n, "launches with a deviating stream", tot_l, "deviating stream-launches", tot_s, "of", n * S,
      "ms/launch(host)", [round(x, 3) for x in ms], "| launches where device-side and host-side comparison disagree:", disagree if os.environ.get("CB_CPU_COMPARE") else "n/a")
