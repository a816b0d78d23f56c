#!/usr/bin/env python3
"""ISA lint of the gfx950 kernels: no packed-fp32 instruction may pick DIFFERENT halves of its register sources for the low lane.

Finding of round 4 (tools/diag/repro, profiles/archive/r04_pk_opsel_hazard.md): a VOP3P fp32 instruction whose op_sel differs between its
sources - e.g. `v_pk_add_f32 d, a, b op_sel:[0,1]`: low lane = a.lo + b.hi - computes its low-lane result for lanes 48-63 from a ZERO
instead of the high half while another wave of the SIMD issues 128-bit-operand MFMAs (v_mfma_f32_16x16x32_bf16 and friends).  Uniform
selections (op_sel all 0 or all 1 over the register sources), op_sel_hi in any combination, neg / neg_hi and scalar sources were clean
in every run.  The kernels therefore spell half swaps with plain 32-bit instructions, and this lint keeps the compiler (which forms such
instructions from plain 2-vector code by itself) and later edits from bringing them back.

  tools/isa_lint.py [fast|mfma|exact ...]     recompile the translation units with the shipped flags and lint the assembly
  tools/isa_lint.py --so <lib.so>             lint what a BUILT library really holds: the gfx950 code objects inside it are
                                              extracted and disassembled (llvm-objdump), so builds with EXTRA_HIPFLAGS, -DFMD_TUNING,
                                              other worker counts ... are checked too - csrc/Makefile runs this on every link
  exit status 1 when an instruction of the forbidden form is found (with --so also: any kernel with scratch)
"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rtl_fm_player_amd", "csrc")
PK = re.compile(r"^\s+(v_pk_(?:fma|mul|add)_f32|v_pk_mov_b32)\s+(.*)$")


def device_asm(kind, extra=()):
    out = "/tmp/fmd_lint_%s.s" % kind
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
                    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-fno-slp-vectorize", "-S", "--cuda-device-only",
                    "-o", out, os.path.join(CSRC, "fmd_kernels_%s.hip" % kind)] + list(extra), check=True, stderr=subprocess.DEVNULL)
    return out


ROCM = os.environ.get("ROCM") or os.environ.get("ROCM_PATH") or "/opt/rocm"     # (csrc/Makefile passes its own ROCM)
LLVM_BIN = os.path.join(ROCM, "lib", "llvm", "bin")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
PRIVATE_SEGMENT_MAX = 64       # bytes a kernel's private segment may reserve WITHOUT touching it (SGPR spill slots that went to VGPR lanes: 20 - 36 seen)
SGPR_SPILL_MAX = 200           # per kernel (118 - 160 in the stereo kernels at the register limit, 26 - 46 in the mono ones): beyond that something changed


class ToolUnavailable(RuntimeError):
    """llvm-objdump / llvm-readelf missing or without --offloading support: the lint could not LOOK - not the same as having found something."""



def so_disassembly(so_path):
    """Disassembly of every gfx950 code object bundled in a built shared library, and (kernel name, private segment bytes) of its
    kernels from the code objects' notes.  Works on a copy: llvm-objdump writes the extracted bundles next to its input."""
    import glob, shutil, tempfile
    tmp = tempfile.mkdtemp(prefix="fmd_lint_so_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(so_path, local)
        for tool in ("llvm-objdump", "llvm-readelf"):
            if not os.path.exists(os.path.join(LLVM_BIN, tool)):
                raise ToolUnavailable("%s not found under %s (set ROCM=...)" % (tool, LLVM_BIN))
        r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise ToolUnavailable("llvm-objdump --offloading failed: %s" % r.stderr.strip()[-200:])
        objs = sorted(glob.glob(local + ".*gfx950*"))
        if not objs:
            raise RuntimeError("no gfx950 code object in %s" % so_path)
        text, kernels = [], []
        for o in objs:
            text.append(subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", o], check=True, capture_output=True, text=True).stdout)
            notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", o], check=True, capture_output=True, text=True).stdout
            names = re.findall(r"\.name:\s+(\S+)", notes)
            scratch = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
            spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", notes)]
            sspills = [int(x) for x in re.findall(r"\.sgpr_spill_count:\s+(\d+)", notes)]
            kernels += list(zip(names, scratch, spills, sspills))
        return "\n".join(text), kernels
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


SCRATCH_OP = re.compile(r"^\s+(scratch_(?:load|store)\w*|buffer_(?:load|store)\w*\s.*\boffen\b.*\bs\[0:3\])")


def kernels_touching_scratch(text):
    """Kernel symbols whose code holds a scratch access.  (hipcc leaves a private segment of a few bytes on some kernels whose SGPR
    spills all went to VGPR lanes - a frame object nothing reads or writes; what must not happen is an ACCESS: a reload in the tile loop
    waits on vmcnt and drains the IQ words in flight.)"""
    hit, cur = [], None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
        elif cur and SCRATCH_OP.match(line.split("//")[0]) and cur not in hit:
            hit.append(cur)
    return hit


def lint_so(so_path):
    text, kernels = so_disassembly(so_path)
    touching = kernels_touching_scratch(text)
    kernels = [(n, sc, sp, n in touching, ss) for n, sc, sp, ss in kernels]
    n_pk, found = 0, []
    for no, line in enumerate(text.splitlines(), 1):
        line = line.split("//")[0]
        if PK.match(line):
            n_pk += 1
            r = check_line(line)
            if r:
                found.append((no, r[0], line.strip(), r[1]))
    return n_pk, found, kernels


def split_operands(s):
    """'v[0:1], v[2:3], s[4:5] op_sel:[0,1,0] neg_lo:[0,1,0]' -> (['v[0:1]', 'v[2:3]', 's[4:5]'], {'op_sel': [0,1,0], ...})"""
    s = s.split(";")[0].strip()
    mods = {m.group(1): [int(x) for x in m.group(2).split(",")] for m in re.finditer(r"(\w+):\[([0-9,]+)\]", s)}
    s = re.sub(r"\s*\w+:\[[0-9,]+\]", "", s)
    ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", s) if o.strip()]
    return ops, mods


def check_line(line):
    """(level, why) or None.  level "error": the signature that computed wrong results in tools/diag/repro (low lane = LOW half of
    one vector register pair with the HIGH half of a later, different pair: forms 0, 1, 6, 7, 11 of profiles/archive/r04_pk_opsel_hazard.md);
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
    if kinds and kinds[0] == "--so":
        kinds = ["so:" + k for k in kinds[1:]]
    for kind in kinds:
        if kind.startswith("so:"):
            try:
                n_pk, found, kernels = lint_so(kind[3:])
            except ToolUnavailable as e:
                # the lint could not look: say so and leave the decision to the caller (exit code 3; csrc/Makefile refuses unless ISA_LINT_OPTIONAL=1)
                print("%s: lint tool unavailable: %s" % (kind, e))
                return 3
            # a spilled VGPR, a scratch access, a private segment beyond the few bytes of untouched SGPR spill slots, or an SGPR spill count out of range
            bad = [k for k in kernels if k[2] or k[3] or k[1] > PRIVATE_SEGMENT_MAX or k[4] > SGPR_SPILL_MAX]
            if n_pk < 500 or not kernels:
                print("%s: %d packed-fp32 instructions in %d kernels: not the library this lint is for" % (kind, n_pk, len(kernels)))
                rc |= 1
            for name, sc, sp, touch, ss in bad:
                print("   error: kernel %s: %d spilled VGPRs, %d spilled SGPRs (limit %d), %s its %d bytes of scratch (limit untouched: %d)" %
                      (name, sp, ss, SGPR_SPILL_MAX, "accesses" if touch else "does not access", sc, PRIVATE_SEGMENT_MAX))
            for name, sc, sp, touch, ss in kernels:
                if sc and not sp and not touch and sc <= PRIVATE_SEGMENT_MAX:
                    print("   note: kernel %s has a private segment of %d bytes it never accesses (SGPR spill slots that went to VGPR lanes)" % (name, sc))
            rc |= bool(bad)
        else:
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
