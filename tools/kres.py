#!/usr/bin/env python3
"""Resources of every fmd_fused_kernel instantiation in a device-only assembly listing.
   tools/kres.py <fast|mfma|exact> [extra hipcc flags]   (compiles csrc/fmd_kernels_<kind>.hip to /tmp/k_<kind>.s)"""
import re, subprocess, sys, os
kind = sys.argv[1] if len(sys.argv) > 1 else "fast"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "rtl_fm_player_amd", "csrc")
out = "/tmp/k_%s.s" % kind
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
                "-I" + os.path.join(root, "include"), "-I" + src, "-fno-slp-vectorize", "-S", "--cuda-device-only",
                "-o", out, os.path.join(src, "fmd_kernels_%s.hip" % kind)] + sys.argv[2:], check=True,
               stderr=subprocess.DEVNULL)
s = open(out).read()
meta = s[s.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    inst = re.search(r"fmd_fused_kernelI(\S+?)EEv", name)
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or re.search(r"()", "")).group(1)
    body = s[s.index(name + ":"):]
    body = body[:body.index("s_endpgm")]
    cnt = lambda pat: len(re.findall(pat, body))
    print("%-22s vgpr %s agpr %s sgpr %s lds %s scratch %s spills %s | mfma %d valu~%d ds %d vmem %d" % (
        inst.group(1) if inst else name, g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("group_segment_fixed_size"),
        g("private_segment_fixed_size"), g("vgpr_spill_count"), cnt(r"\bv_mfma"), cnt(r"\n\s+v_(?!mfma)"),
        cnt(r"\n\s+ds_"), cnt(r"\n\s+(buffer|global)_")))
