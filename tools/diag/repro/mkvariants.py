#!/usr/bin/env python3
"""Builds the code objects of the neighbour experiment into tools/diag/repro/build/ (hipcc cross-compiles; run here or on the box):
source-level variants of victim.hip (-DV_*), assembly-level variants of the base build (s_nop after every vector instruction,
a larger register allocation), the neighbour and the host program."""
import os, re, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
out = os.path.join(here, "build")
os.makedirs(out, exist_ok=True)
LLVM = "/opt/rocm/lib/llvm/bin"
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fno-slp-vectorize", "--cuda-device-only"]

def run(cmd):
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)

def to_s(name, defs, src="victim.hip"):
    s = os.path.join(out, name + ".s")
    run([HIPCC] + FLAGS + ["-D" + d for d in defs] + ["-S", "-o", s, os.path.join(here, src)])
    return s

def assemble(s_path):
    o = s_path[:-2] + ".o"
    run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s_path, "-o", o])
    run([LLVM + "/ld.lld", "-shared", o, "-o", s_path[:-2] + ".hsaco"])
    os.remove(o)

def body_edit(text, fn):
    """apply fn to the lines of the victim kernel's body only"""
    lines = text.split("\n")
    a = next(i for i, l in enumerate(lines) if l.startswith("victim:"))
    b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
    lines[a + 1:b] = fn(lines[a + 1:b])
    return "\n".join(lines)

VALU = re.compile(r"^\s+v_(?!mfma)")
def nop_after(pat, nop):
    def f(ls):
        o = []
        for l in ls:
            o.append(l)
            if pat.match(l): o.append("\t" + nop)
        return o
    return f
def nop_before(pat, nop):
    def f(ls):
        o = []
        for l in ls:
            if pat.match(l): o.append("\t" + nop)
            o.append(l)
        return o
    return f

SRC = {
    "base": [],
    "nodpp": ["V_NODPP"],
    "plainrot": ["V_PLAINROT"],
    "scalarfma": ["V_SCALARFMA"],
    "vtaps": ["V_VTAPS"],
    "noload": ["V_NOLOAD"],
    "loadtop": ["V_LOADTOP"],
    "prioflip": ["V_PRIOFLIP"],
    "w1": ["V_WAVES=1"],
    "plain_all": ["V_NODPP", "V_PLAINROT", "V_SCALARFMA", "V_VTAPS", "V_NOLOAD"],
    "noasm": ["V_NODPP", "V_PLAINROT"],
}
for f in range(19):
    SRC["form%d" % f] = ["V_ROTFORM=%d" % f]
    SRC["form%d_pad" % f] = ["V_ROTFORM=%d" % f, "V_JOINPAD"]
only = sys.argv[1:]
for name, defs in SRC.items():
    if only and name not in only: continue
    assemble(to_s(name, defs))
base = open(os.path.join(out, "base.s")).read()
ASM = {
    "nop7_all": body_edit(base, nop_after(VALU, "s_nop 7")),
    "nop0_all": body_edit(base, nop_after(VALU, "s_nop 0")),
    "nop1_all": body_edit(base, nop_after(VALU, "s_nop 1")),
    "nop3_pk": body_edit(base, nop_after(re.compile(r"^\s+v_pk_"), "s_nop 3")),
    "nop3_sdwa": body_edit(base, nop_after(re.compile(r"^\s+v_cvt_f32_i32_sdwa"), "s_nop 3")),
    "nop3_before_pkadd": body_edit(base, nop_before(re.compile(r"^\s+v_pk_add_f32"), "s_nop 3")),
    "vgpr152": re.sub(r"(victim\n(?:.*\n)*?\s+\.amdhsa_next_free_vgpr) \d+", r"\1 152", base, count=1),
}
def at_line(text, pat, before=None, after=None, others_before=None):
    """edit around the FIRST line of the victim body that matches pat (others_before: a nop before every OTHER v_pk_add_f32)"""
    def f(ls):
        i = next(k for k, l in enumerate(ls) if pat in l)
        o = []
        for k, l in enumerate(ls):
            if k == i and before: o.append("\t" + before)
            if k != i and others_before and re.match(r"^\s+v_pk_add_f32", l): o.append("\t" + others_before)
            o.append(l)
            if k == i and after: o.append("\t" + after)
        return o
    return body_edit(text, f)
FAIL = "v_pk_add_f32 v[44:45], v[38:39], v[28:29]"      # y2[2] of the base build: the value that goes wrong (profiles/archive/r04a)
assert FAIL in base
def swap_wait(text):
    def f(ls):
        i = next(k for k, l in enumerate(ls) if FAIL in l)
        j = next(k for k in range(i, len(ls)) if "s_waitcnt vmcnt(0)" in ls[k])
        w = ls.pop(j)
        ls.insert(i - 1, w)          # in front of the ;;#ASMSTART of the statement
        return ls
    return body_edit(text, f)
ASM.update({
    "b_nop3_before_this": at_line(base, FAIL, before="s_nop 3"),
    "b_nop0_before_this": at_line(base, FAIL, before="s_nop 0"),
    "b_nop1_before_this": at_line(base, FAIL, before="s_nop 1"),
    "b_vnop_before_this": at_line(base, FAIL, before="v_nop"),
    "b_nop3_after_this": at_line(base, FAIL, after="s_nop 3"),
    "b_nop0_after_this": at_line(base, FAIL, after="s_nop 0"),
    "b_nop3_before_others": at_line(base, FAIL, others_before="s_nop 3"),
    "b_wait_first": swap_wait(base),
})
for name, text in ASM.items():
    if only and name not in only: continue
    p = os.path.join(out, name + ".s")
    open(p, "w").write(text)
    assemble(p)
assemble(to_s("neighbour", [], "neighbour.hip"))
run([HIPCC, "-O2", "-o", os.path.join(out, "host"), os.path.join(here, "host.cpp")])
print(sorted(f for f in os.listdir(out) if f.endswith(".hsaco")))
