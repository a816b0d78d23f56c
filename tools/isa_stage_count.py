#!/usr/bin/env python3
"""Instruction classes of a fused kernel's tile loop, segment by segment (the s_setprio markers the stages carry), from the device
assembly tools/kres.py leaves in /tmp/k_<kind>.s.   tools/isa_stage_count.py <kind> <instantiation substring, e.g. ILb0ELi2ELi45ELi3ELb0>
Only instructions in program order between the markers are counted: cold blocks the compiler moved behind the loop are listed as 'tail'."""
import re, sys, collections
kind, inst = sys.argv[1], sys.argv[2]
s = open("/tmp/k_%s.s" % kind).read()
m = re.search(r"^(_ZN\S*fmd_fused_kernel%s\S*):" % re.escape(inst), s, re.M)
body = s[m.end():]
body = body[:body.index("s_endpgm")]
lines = [l.strip() for l in body.splitlines()]
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait/nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("scratch_"): return "vmem"
    return "other"
segs, cur, name = [], collections.Counter(), "prologue"
ops = collections.Counter()
for l in lines:
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
        continue
    op = l.split()[0]
    if op == "s_setprio":
        segs.append((name, cur)); cur = collections.Counter(); name = "prio " + l.split()[1]
        continue
    cur[cls(op)] += 1
    if cls(op) == "valu": ops[(name, re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", op))] += 1
segs.append((name, cur))
tot = collections.Counter()
for n, c in segs:
    print("%-10s %s" % (n, dict(sorted(c.items()))), "sum", sum(c.values()))
if "--ops" in sys.argv:
    for n, _ in segs:
        top = sorted(((k[1], v) for k, v in ops.items() if k[0] == n), key=lambda kv: -kv[1])[:14]
        print(n, top)
