#!/usr/bin/env python3
"""The three-term energy model of DESIGN.md section 5 applied to a committed profile summary (tools/profile_round.sh + summarize_profile.py):

    energy per launch = BASE_W x time + NJ_PER_BYTE x HBM bytes + TOGGLE x sum(instructions x list price)
    time at the cap   = (bytes + instruction terms) / (CAP_W - BASE_W)        (the kernel is energy-bound: DESIGN.md section 5)

with the per-launch instruction counts the PMC passes measured (SQ_INSTS_VALU - of which SQ_INSTS_MFMA -, SQ_INSTS_LDS, SQ_INSTS_SALU) and the
list prices of tools/ubench/power_price.hip (profiles/archive/r18_power_price.txt, r21_power_price_lds.txt).  Prints predicted against measured kernel time.
    python tools/energy_model.py profiles/archive/r30q_stereo_summary.json [more summaries ...]
Constants: BASE_W 367 (every SIMD on s_nop), CAP_W = what the package was measured at under that workload (1 379 stereo, 1 400 mono / narrow FM),
NJ_PER_BYTE 0.124 (device-to-device copy, profiles/archive/r33_stream_power.txt), list prices in nJ per wave instruction: MFMA i8 3.8, vector 0.65 (the kernels'
mix of plain 0.5 - 0.63 and packed 1.56), LDS 2.1, scalar 0.05; TOGGLE 1.6 = real operands over the micro-benchmark's constants: ONE factor, chosen so that
the stereo kernel closes (rounds 4 / 5 stereo summaries: 0.98 - 1.02 of the measured time; mono / narrow FM are over-predicted by 10 - 12 %: DESIGN.md says why)."""
import json, sys

BASE_W, NJ_PER_BYTE, TOGGLE = 367.0, 0.124, 1.6
PRICE = {"mfma": 3.8, "valu": 0.65, "lds": 2.1, "salu": 0.05}

def counters(d):
    c = {}
    for grp in d.values():
        if isinstance(grp, dict):
            for k, v in grp.items():
                if isinstance(v, dict) and "mean_per_launch" in v:
                    c[k] = float(v["mean_per_launch"])
    return c

for path in sys.argv[1:]:
    d = json.load(open(path))
    c = counters(d)
    wl = d["hbm_traffic"]["workload"]["workload"]
    cap = 1379.0 if "stereo" in wl else 1400.0
    mfma = c.get("SQ_INSTS_MFMA", 0.0)
    valu = c["SQ_INSTS_VALU"] - mfma
    lds, salu = c["SQ_INSTS_LDS"], c.get("SQ_INSTS_SALU", 0.0)
    nj_instr = TOGGLE * (mfma * PRICE["mfma"] + valu * PRICE["valu"] + lds * PRICE["lds"] + salu * PRICE["salu"])     # nJ per launch
    nj_bytes = NJ_PER_BYTE * d["hbm_traffic"]["traffic_bytes_per_launch"]
    t_pred = (nj_instr + nj_bytes) * 1e-9 / (cap - BASE_W)                                                              # seconds
    t_meas = d["bench_unprofiled"]["roofline"]["kernel_ms"] * 1e-3
    e_meas = cap * t_meas
    print(json.dumps({"summary": path.split("/")[-1], "per_launch_M": {"mfma": round(mfma / 1e6, 2), "other_vector": round(valu / 1e6, 1), "lds": round(lds / 1e6, 2), "scalar": round(salu / 1e6, 1)},
                      "J": {"instructions": round(nj_instr * 1e-9, 4), "bytes": round(nj_bytes * 1e-9, 4), "base_at_measured_time": round(BASE_W * t_meas, 4),
                             "sum": round((nj_instr + nj_bytes) * 1e-9 + BASE_W * t_meas, 4), "measured_cap_x_time": round(e_meas, 4)},
                      "kernel_ms": {"predicted_at_the_cap": round(t_pred * 1e3, 4), "measured": round(t_meas * 1e3, 4), "ratio": round(t_pred / t_meas, 3)}}))
