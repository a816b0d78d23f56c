"""The kernels must not contain the packed-fp32 instruction form that computes wrong results beside double-rate MFMAs
(profiles/archive/r04_pk_opsel_hazard.md): tools/isa_lint.py over the device assembly of the three kernel translation units.
hipcc cross-compiles without a GPU, so this runs in the CPU suite."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402


@pytest.mark.parametrize("line,level", [
    ("\tv_pk_add_f32 v[44:45], v[38:39], v[28:29] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]", "error"),     # round 3's pk_add_rot90
    ("\tv_pk_fma_f32 v[66:67], v[100:101], v[98:99], v[66:67] op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_hi:[0,0,1]", "error"),
    ("\tv_pk_mul_f32 v[68:69], v[102:103], v[100:101] op_sel:[0,1]", "error"),
    ("\tv_pk_fma_f32 v[50:51], v[76:77], v[54:55], v[50:51] op_sel:[0,1,0]", "error"),
    ("\tv_pk_add_f32 v[44:45], v[28:29], v[38:39] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]", "warn"),      # swapped operand first
    ("\tv_pk_fma_f32 v[50:51], v[54:55], v[76:77], v[50:51] op_sel:[1,0,0]", "warn"),
    ("\tv_pk_add_f32 v[56:57], v[54:55], v[54:55] op_sel:[0,1] op_sel_hi:[1,0]", None),                     # both halves of ONE pair
    ("\tv_pk_fma_f32 v[38:39], v[34:35], s[4:5], v[38:39] op_sel:[0,1,0] neg_lo:[0,1,0] neg_hi:[0,1,0]", None),   # scalar pair
    ("\tv_pk_fma_f32 v[84:85], v[108:109], s[62:63], v[84:85] op_sel_hi:[1,0,1]", None),
    ("\tv_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,1] op_sel_hi:[0,1]", None),
    ("\tv_pk_mov_b32 v[2:3], v[4:5], v[6:7] op_sel:[1,0]", None),
    ("\tv_fma_f32 v1, v2, v3, v4", None),
])
def test_rule(line, level):
    r = isa_lint.check_line(line)
    assert (r[0] if r else None) == level, r


@pytest.mark.parametrize("kind", ["fast", "mfma", "exact"])
def test_kernels_hold_no_forbidden_packed_form(kind):
    n_pk, found = isa_lint.lint_file(isa_lint.device_asm(kind))
    assert n_pk > 500, "the listing holds no packed instructions: wrong file?"
    errors = [f for f in found if f[1] == "error"]
    assert not errors, "%d instructions of the forbidden form, first: line %d: %s (%s)" % (
        len(errors), errors[0][0], errors[0][2], errors[0][3])


@pytest.mark.parametrize("kind", ["fast", "mfma", "exact"])
def test_kernels_have_no_scratch_and_no_spills(kind):
    """The tile loop must not touch scratch (a scratch reload waits on vmcnt and drains the IQ words in flight, DESIGN.md
    section 3) - cold paths included: a change that pushes any instantiation over its register budget fails here."""
    import re
    text = open(isa_lint.device_asm(kind)).read()
    meta = text[text.index("amdhsa.kernels:"):]
    names = re.findall(r"\.name:\s+(\S+)", meta)
    scratch = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", meta)]
    spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", meta)]
    sspills = [int(x) for x in re.findall(r"\.sgpr_spill_count:\s+(\d+)", meta)]
    assert names and len(scratch) == len(names) == len(spills)
    bad = [(n, sc, sp) for n, sc, sp in zip(names, scratch, spills) if sp]
    assert not bad, "kernels with VGPR spills: %s" % bad
    assert len(sspills) == len(names)
    # a private segment may exist (SGPR spill slots hipcc then served from VGPR lanes: 20 - 36 bytes today) - but nothing may ACCESS it, it stays
    # a few bytes, and the SGPR spill counts stay in the range they have today (a spill served through another addressing form would
    # otherwise pass silently)
    body = text[:text.index("amdhsa.kernels:")]
    assert not re.findall(r"^\s+scratch_(?:load|store)", body, re.M), "scratch access in the device code"
    big = [(n, sc) for n, sc in zip(names, scratch) if sc > isa_lint.PRIVATE_SEGMENT_MAX]
    assert not big, "private segments beyond %d bytes: %s" % (isa_lint.PRIVATE_SEGMENT_MAX, big)
    many = [(n, ss) for n, ss in zip(names, sspills) if ss > isa_lint.SGPR_SPILL_MAX]
    assert not many, "SGPR spill counts beyond %d: %s" % (isa_lint.SGPR_SPILL_MAX, many)
    # the shipped default kernels (every stage on the matrix pipe: MX = 2, no debug taps) exist in this listing and obey the same bounds
    assert [n for n in names if re.search(r"fmd_fused_kernelILb0ELi[12]ELi(45|64)ELi2ELb0E", n)] or kind != "mfma"


def test_the_built_library_is_what_gets_linted():
    """the lint above recompiles the sources with fixed flags; what ships is the .so.  tools/isa_lint.py --so extracts the
    gfx950 code objects of the built library and lints their disassembly (csrc/Makefile runs it on every link): no forbidden packed
    form, no scratch, no spills - whatever flags the library was built with."""
    so = os.path.join(ROOT, "rtl_fm_player_amd", "libfmdemod_mi355x.so")
    if not os.path.exists(so):
        import rtl_fm_player_amd as R
        R.build_library()
    n_pk, found, kernels = isa_lint.lint_so(so)
    assert n_pk > 1500 and len(kernels) >= 16, (n_pk, len(kernels))
    errors = [f for f in found if f[1] == "error"]
    assert not errors, errors[:3]
    assert not [k for k in kernels if k[2] or k[3]], [k for k in kernels if k[2] or k[3]]     # spilled VGPRs / a scratch access
    assert not [k for k in kernels if k[1] > isa_lint.PRIVATE_SEGMENT_MAX or k[4] > isa_lint.SGPR_SPILL_MAX], kernels
    assert any("fmd_fused_kernel" in k[0] for k in kernels)


def test_a_missing_lint_tool_is_not_a_finding(tmp_path, monkeypatch):
    """"could not look" (no llvm-objdump --offloading under $ROCM) is reported as such - exit code 3, which csrc/Makefile turns into a
    refusal or, with ISA_LINT_OPTIONAL=1, a warning - not as a forbidden instruction."""
    monkeypatch.setattr(isa_lint, "LLVM_BIN", str(tmp_path))
    with pytest.raises(isa_lint.ToolUnavailable):
        isa_lint.so_disassembly(os.path.join(ROOT, "rtl_fm_player_amd", "libfmdemod_mi355x.so"))
    assert isa_lint.main(["--so", os.path.join(ROOT, "rtl_fm_player_amd", "libfmdemod_mi355x.so")]) == 3
