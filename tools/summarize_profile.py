#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/<tag>_*.{csv,json}."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# optional: <raw dir> <destination dir> (tools/profile_round.sh summarises on the GPU box into gpurun_out/<tag>/,
# because the raw rocprofv3 output is too large to travel back)
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "gpurun_out", tag)
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def find(pat):
    return sorted(glob.glob(os.path.join(src, pat), recursive=True))


out = {"tag": tag}
for name in ("bench_unprofiled.json", "bench_traced.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        for line in open(p):
            line = line.strip()
            if line.startswith("{"):
                out[name[:-5]] = json.loads(line)

# kernel stats (rocprofv3 --kernel-trace --stats)
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as o:
        w = csv.DictWriter(o, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:                       # torch's own kernels have kilobyte-long names
            r = dict(r)
            r["Name"] = r["Name"][:160]
            w.writerow(r)
    out["kernel_stats"] = [r for r in rows if "fmd_" in r.get("Name", "")]
# per-dispatch durations of the fused kernel
for f in find("trace/**/*kernel_trace.csv"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f))
         if "fmd_" in r["Kernel_Name"]]
    if d:
        out["kernel_trace_ms"] = {"n": len(d), "mean": sum(d) / len(d), "min": min(d), "max": max(d)}
        steps = (out.get("bench_traced") or {}).get("steps")
        if steps and len(d) >= steps:          # the timed steps are the last ones; the launches before them settle the clocks
            t = d[-steps:]
            out["kernel_trace_ms_timed_steps"] = {"n": len(t), "mean": sum(t) / len(t), "min": min(t), "max": max(t)}
        r0 = next(r for r in csv.DictReader(open(f)) if "fmd_" in r["Kernel_Name"])
        out["kernel_resources"] = {k: r0.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                                           "Scratch_Size", "Workgroup_Size", "Grid_Size")}
        # rocprofv3's VGPR_Count is the ARCH-VGPR half of gfx950's unified register file as the trace reports it (128 for a kernel whose code object says
        # .vgpr_count 229 - 256 with no AGPRs in use): the code object's own figure is what the occupancy follows (label it)
        out["kernel_resources"]["note"] = "VGPR_Count as rocprofv3 reports it; the code object's .vgpr_count (tools/isa_report.sh) is the allocation: 229 stereo / 168 mono for FMD_MATH_FAST_MFMA_F"


def pmc(dirname):
    acc = collections.defaultdict(list)
    for f in find(dirname + "/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "fmd_" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for k, v in acc.items()}


for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_sq4"):
    c = pmc(d)
    if c:
        out[d] = c

b = out.get("bench_unprofiled", {})
if b and "pmc_fetch" in out and "pmc_write" in out:
    fetch_kib = out["pmc_fetch"]["FETCH_SIZE"]["mean_per_launch"]
    write_kib = out["pmc_write"]["WRITE_SIZE"]["mean_per_launch"]
    algo = b["roofline"]["algorithmic_bytes_per_launch"]
    out["hbm_traffic"] = {
        "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
        "fetch_bytes_raw": fetch_kib * 1024, "write_bytes": write_kib * 1024,
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE reads half the bytes of a 16 B/lane streaming read on gfx950
        "fetch_bytes_corrected_x2": 2 * fetch_kib * 1024,
        "traffic_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
        "algorithmic_bytes_per_launch": algo,
        "traffic_over_algorithmic": (2 * fetch_kib * 1024 + write_kib * 1024) / algo,
        "workload": b["config"],
    }
json.dump(out, open(os.path.join(dst, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps({k: out[k] for k in out if k in ("kernel_trace_ms", "kernel_resources", "hbm_traffic")}, indent=1))
