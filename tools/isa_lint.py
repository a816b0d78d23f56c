#!/usr/bin/env python3
"""ISA lint of the gfx950 kernels: no packed-fp32 instruction may pick DIFFERENT halves of its register sources for the low lane.

Finding of round 4 (tools/diag/repro, profiles/r04_pk_opsel_hazard.md): a VOP3P fp32 instruction whose op_sel differs between its
sources - e.g. `v_pk_add_f32 d, a, b op_sel:[0,1]`: low lane = a.lo + b.hi - computes its low-lane result for lanes 48-63 from a ZERO
instead of the high half while another wave of the SIMD issues 128-bit-operand MFMAs (v_mfma_f32_16x16x32_bf16 and friends).  Uniform
selections (op_sel all 0 or all 1 over the register sources), op_sel_hi in any combination, neg / neg_hi and scalar sources were clean
in every run.  The kernels therefore spell half swaps with plain 32-bit instructions, and this lint keeps the compiler (which forms such
instructions from plain 2-vector code by itself) and later edits from bringing them back.

  tools/isa_lint.py [fast|mfma|exact ...]     exit status 1 when an instruction of the forbidden form is found
"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rtl_fm_player_amd", "csrc")
PK = re.compile(r"^\s+(v_pk_(?:fma|mul|add)_f32|v_pk_mov_b32)\s+(.*)$")


def device_asm(kind, extra=()):
    out = "/tmp/fmd_lint_%s.s" % kind
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
                    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-fno-slp-vectorize", "-S", "--cuda-device-only",
                    "-o", out, os.path.join(CSRC, "fmd_kernels_%s.hip" % kind)] + list(extra), check=True, stderr=subprocess.DEVNULL)
    return out


def split_operands(s):
    """'v[0:1], v[2:3], s[4:5] op_sel:[0,1,0] neg_lo:[0,1,0]' -> (['v[0:1]', 'v[2:3]', 's[4:5]'], {'op_sel': [0,1,0], ...})"""
    s = s.split(";")[0].strip()
    mods = {m.group(1): [int(x) for x in m.group(2).split(",")] for m in re.finditer(r"(\w+):\[([0-9,]+)\]", s)}
    s = re.sub(r"\s*\w+:\[[0-9,]+\]", "", s)
    ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", s) if o.strip()]
    return ops, mods


def check_line(line):
    """(level, why) or None.  level "error": the signature that computed wrong results in tools/diag/repro (low lane = LOW half of
    one vector register pair with the HIGH half of a later, different pair: forms 0, 1, 6, 7, 11 of profiles/r04_pk_opsel_hazard.md);
    "warn": other mixed selections, which were clean in every run (forms 8, 13 - 15: a high half first; horizontal operations on one
    register pair: 17, 18) and are only counted."""
    m = PK.match(line)
    if not m:
        return None
    ops, mods = split_operands(m.group(2))
    srcs = ops[1:]
    sel = (mods.get("op_sel", []) + [0] * len(srcs))[:len(srcs)]
    regs = [(i, o) for i, o in enumerate(srcs) if o.startswith("v") or o.startswith("a")]
    if m.group(1) == "v_pk_mov_b32":           # a move: lane 0 reads source 0 only, lane 1 source 1 only - nothing is combined
        return None
    err = [(o1, o2) for (i, o1) in regs for (j, o2) in regs if i < j and o1 != o2 and sel[i] == 0 and sel[j] == 1]
    if err:
        return "error", "low lane = low half of %s with high half of %s (op_sel %s)" % (err[0][0], err[0][1], sel)
    if len({sel[i] for i, _ in regs}) > 1 and len({o for _, o in regs}) > 1:
        return "warn", "mixed op_sel %s" % sel
    return None


def lint_file(path):
    found, n_pk, kernel = [], 0, "?"
    for no, line in enumerate(open(path), 1):
        if PK.match(line):
            n_pk += 1
            r = check_line(line)
            if r:
                found.append((no, r[0], line.strip(), r[1]))
    return n_pk, found


def main(kinds):
    rc = 0
    for kind in kinds:
        path = kind if kind.endswith(".s") else device_asm(kind)
        n_pk, found = lint_file(path)
        n_err = sum(1 for f in found if f[1] == "error")
        print("%s: %d packed-fp32 instructions, %d of the forbidden form, %d other mixed selections" % (kind, n_pk, n_err, len(found) - n_err))
        seen = {}
        for no, level, text, why in found:
            seen.setdefault((level, re.sub(r"[vs]\[?\d+(:\d+)?\]?", "R", text)), []).append(no)
        for (level, form), lines in sorted(seen.items(), key=lambda kv: (kv[0][0], -len(kv[1]))):
            print("   %-5s %4d x %s   (lines %s%s)" % (level, len(lines), form, ", ".join(map(str, lines[:4])), " ..." if len(lines) > 4 else ""))
        rc |= n_err > 0
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or ["fast", "mfma", "exact"]))
