#!/usr/bin/env python3
"""GPU box: what MOVING bytes costs.  A device-to-device copy and a read-only reduction of 1 GiB of random bytes, each looped for ~3 s with the package
power and shader clock sampled from hwmon beside it (bench.PowerWatch): GB/s, W, MHz, and nJ per byte moved above the 367 W the part draws with every
SIMD on s_nop (profiles/archive/r18_power_price.txt).  The fused kernel's own stream (stage A alone, tools/energy_ablate.sh mask 30) sits next to these.
   python tools/stream_power.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
N = 1 << 30
g = torch.Generator(device=dev); g.manual_seed(1)
src = torch.randint(0, 256, (N,), dtype=torch.uint8, device=dev, generator=g)
dst = torch.empty_like(src)
src32 = src.view(torch.int32)
props = torch.cuda.get_device_properties(dev)

def leg(name, fn, bytes_per_call, seconds=3.0):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    w = bench.PowerWatch(props); w.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    el = time.perf_counter() - t0
    p = w.stop() or {}
    gbs = bytes_per_call * n / el / 1e9
    pw = p.get("package_w_mean")
    print(json.dumps({"leg": name, "GB_per_s": round(gbs, 1), "package_w": pw, "sclk_mhz": p.get("sclk_mhz_mean"),
                      "nJ_per_byte_above_367W": round((pw - 367.0) / gbs, 4) if pw else None}), flush=True)

leg("d2d copy (read + write counted)", lambda: dst.copy_(src), 2 * N)
leg("read-only sum of int32", lambda: torch.sum(src32), N)
leg("read-only max of uint8", lambda: torch.max(src), N)
