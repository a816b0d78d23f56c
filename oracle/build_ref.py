#!/usr/bin/env python3
"""Compile the REFERENCE's own hot path into oracle/_ref/ (test infrastructure only).

    python oracle/build_ref.py            # -> oracle/_ref/libref.so, oracle/_ref/libref_ring.so

Recipe.  The reference program (src/rtl_fm_player.c) is one translation unit that also
holds the USB / SDL / console code and includes <libusb.h> and <SDL2/SDL.h>, which this
image does not have; its build system (CMake + pkg-config for those libraries) is not run.
The hot path itself - src/rtl_fm_player.c:195-788, i.e. init_u8_f32_table ... full_demod -
uses nothing from them: it needs the type / constant / table lines of
include/rtl_fm_player.h listed in HEADER_RANGES below and six libc headers.  This script
reads exactly those line ranges from the sources WHERE THEY LIE under /root/reference,
joins them in memory with oracle/ref_shim.c (this repository's handle API, see its header)
and pipes the text to `gcc -O3 -x c -` (the reference's CMake Release flags: no -march,
no -ffast-math).  No reference text is written to disk, nothing of
it is committed, no stand-in header or library is made.  `#line` directives keep compiler
diagnostics pointing at the real files.

libref_ring.so is the same for the ingest boundary: rtlsdr_callback
(src/rtl_fm_player.c:790-837) with the ring globals (include/rtl_fm_player.h:64-74), the
consumer side being oracle/ref_ring_shim.c's restatement of demod_thread_fn :863-876.  It
includes the reference's own include/rtl-sdr.h (which is complete in the tree) for
rtlsdr_dev_t; librtlsdr's rtlsdr_cancel_async, only reached when the program is shutting
down, stays an undefined weak symbol and is never called.

Outputs go only to oracle/_ref/ (git-ignored, NOT gpurun-ignored: the .so files travel to
the GPU box like the product's own built library; /root/reference does not exist there and
nothing at run time reads it).  Skipped with a message when /root/reference is absent.
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("FMD_REFERENCE_DIR", "/root/reference")
OUT = os.path.join(HERE, "_ref")

SRC = "src/rtl_fm_player.c"
HDR = "include/rtl_fm_player.h"

# (file, first line, last line) - 1-based, inclusive
HEADER_RANGES = [
    (HDR, 30, 46),     # buffer-size macros, PI constants, DEEMPHASIS_*
    (HDR, 56, 56),     # _beverbose (read by init_lp_real_f32)
    (HDR, 95, 110),    # struct lp_real
    (HDR, 127, 175),   # struct demod_state
    (HDR, 205, 206),   # RMSShadowBuf (written by full_demod)
    (HDR, 274, 281),   # u8_f32_table, lp_filter_f32
]
HOT_PATH = [(SRC, 195, 788)]

RING_HEADERS = ("rtl-sdr.h", "rtl-sdr_export.h")   # included by the ring library as they lie in the reference tree
RING_RANGES = [
    (HDR, 30, 46),
    (HDR, 56, 57),     # _beverbose, _do_exit
    (HDR, 64, 74),     # _input_buffer ring + counters (and the output ring, unused here)
    (HDR, 95, 110),
    (HDR, 112, 125),   # struct dongle_state
    (HDR, 127, 175),
    (HDR, 210, 211),   # dongle, demod globals
    (SRC, 790, 837),   # rtlsdr_callback
]

LIBC = ["math.h", "string.h", "stdint.h", "stdlib.h", "stddef.h", "stdio.h", "pthread.h"]

# first words expected at the start of some ranges: a guard against a reference tree whose
# line numbers differ from the one this recipe was written for
ANCHORS = {
    (HDR, 30): "#define DEFAULT_SAMPLE_RATE",
    (HDR, 56): "static volatile int _beverbose",
    (HDR, 64): "/* 8 MB */",
    (HDR, 95): "struct lp_real",
    (HDR, 112): "struct dongle_state",
    (HDR, 127): "struct demod_state",
    (HDR, 205): "float RMSShadowBuf",
    (HDR, 210): "struct dongle_state dongle;",
    (HDR, 274): "static float u8_f32_table",
    (SRC, 195): "void init_u8_f32_table()",
    (SRC, 790): "static void rtlsdr_callback(",
}
ENDS = {
    (SRC, 788): "}",      # end of full_demod
    (SRC, 837): "}",      # end of rtlsdr_callback
    (HDR, 110): "};",
    (HDR, 125): "};",
    (HDR, 175): "};",
}


def _lines(path):
    with open(os.path.join(REF, path), "r", encoding="utf-8", errors="replace") as f:
        return f.read().split("\n")


def _slice(ranges):
    cache = {}
    out = []
    for path, a, b in ranges:
        ls = cache.setdefault(path, _lines(path))
        want = ANCHORS.get((path, a))
        if want is not None and not ls[a - 1].startswith(want):
            raise SystemExit("build_ref: %s:%d does not start with %r - different reference tree?" % (path, a, want))
        want = ENDS.get((path, b))
        if want is not None and ls[b - 1].strip() != want:
            raise SystemExit("build_ref: %s:%d is not %r - different reference tree?" % (path, b, want))
        out.append('#line %d "%s"' % (a, os.path.join(REF, path)))
        out.extend(ls[a - 1:b])
    return out


def _compile(text, out_so, extra=()):
    cmd = ["gcc", "-O3", "-fPIC", "-shared", "-fvisibility=hidden", "-Wl,-Bsymbolic", "-x", "c", "-",
           "-o", out_so, "-lm", "-lpthread"] + list(extra)
    r = subprocess.run(cmd, input=text.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        sys.stderr.write(r.stderr.decode())
        raise SystemExit("build_ref: gcc failed for %s" % out_so)


def _shim(name):
    path = os.path.join(HERE, name)
    with open(path) as f:
        return ['#line 1 "%s"' % path] + f.read().split("\n")


def load_pin():
    """oracle/ref_pin.json: sha256 of the reference lines this recipe slices (committed; numbers, not source)."""
    try:
        with open(os.path.join(HERE, "ref_pin.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def ref_status():
    """What oracle/_ref/libref.so is: {"slices_sha256", "so_sha256", "pinned"} where the library on disk is the one
    build_ref made from the pinned reference lines (its hash is re-checked here), else None."""
    so = ref_paths()[0]
    try:
        with open(os.path.join(OUT, "libref.meta.json")) as f:
            meta = json.load(f)
        with open(so, "rb") as f:
            actual = hashlib.sha256(f.read()).hexdigest()
    except (OSError, ValueError):
        return None
    pin = load_pin()
    ok = bool(meta.get("pinned")) and pin is not None and pin.get("slices_sha256") == meta.get("slices_sha256") \
        and actual == meta.get("so_sha256")
    try:
        with open(ref_paths()[1], "rb") as f:
            ring_actual = hashlib.sha256(f.read()).hexdigest()
    except OSError:
        ring_actual = None
    ring_ok = bool(meta.get("ring_pinned")) and pin is not None and pin.get("ring_sha256") == meta.get("ring_sha256") \
        and ring_actual is not None and ring_actual == meta.get("ring_so_sha256")
    return dict(meta, pinned=ok, so_sha256_on_disk=actual, ring_pinned=ring_ok, ring_so_sha256_on_disk=ring_actual)


def reference_present():
    return os.path.isfile(os.path.join(REF, SRC)) and os.path.isfile(os.path.join(REF, HDR))


def ref_paths():
    return os.path.join(OUT, "libref.so"), os.path.join(OUT, "libref_ring.so")


def build_ref(force=False, quiet=False):
    """Build oracle/_ref/*.so when the reference tree is present.  Returns True if the
    libraries exist afterwards (freshly built or left from an earlier build)."""
    so, ring = ref_paths()
    if not reference_present():
        if not quiet:
            print("build_ref: %s not present - keeping whatever oracle/_ref/ holds" % REF)
        return os.path.isfile(so)
    os.makedirs(OUT, exist_ok=True)
    shim_stamp = hashlib.sha256()
    for n in ("ref_shim.c", "ref_ring_shim.c", "build_ref.py"):
        with open(os.path.join(HERE, n), "rb") as f:
            shim_stamp.update(f.read())
    for p in (SRC, HDR):
        with open(os.path.join(REF, p), "rb") as f:
            shim_stamp.update(f.read())
    stamp_file = os.path.join(OUT, "stamp")
    stamp = shim_stamp.hexdigest()
    if not force and os.path.isfile(so) and os.path.isfile(ring) and os.path.isfile(stamp_file):
        with open(stamp_file) as f:
            same = f.read().strip() == stamp
        st = ref_status() if same else None           # sources unchanged AND both libraries on disk are the ones recorded then
        if st is not None and st.get("so_sha256_on_disk") == st.get("so_sha256") and st.get("ring_so_sha256_on_disk") == st.get("ring_so_sha256"):
            return True
    head = ["#include <%s>" % h for h in LIBC]
    hot = _slice(HEADER_RANGES) + _slice(HOT_PATH)
    # Pin: the text handed to gcc is untrusted input that becomes a library this process loads.  Its hash (the sliced
    # reference lines only, without the #line markers that carry the tree's path) must equal the committed one
    # (oracle/ref_pin.json); a tree that differs is built only on request (FMD_REFERENCE_UNPINNED=1) and marked so,
    # and bench.py then times the port instead.
    sliced = hashlib.sha256("\n".join(l for l in hot if not l.startswith("#line ")).encode()).hexdigest()
    pin = load_pin()
    pinned = pin is not None and pin.get("slices_sha256") == sliced
    if not pinned and not os.environ.get("FMD_REFERENCE_UNPINNED"):
        raise SystemExit("build_ref: the reference lines hash to %s, oracle/ref_pin.json says %s - not building "
                         "(FMD_REFERENCE_UNPINNED=1 builds an unpinned library)" % (sliced, pin and pin.get("slices_sha256")))
    text = "\n".join(head + hot + _shim("ref_shim.c")) + "\n"
    _compile(text, so)
    # the ring library (rtlsdr_callback, :790-837) is built from the same untrusted tree and loaded by the tests: its sliced lines
    # and the bytes of the two reference headers it includes are pinned the same way ("ring_sha256")
    ring_lines = _slice(RING_RANGES)
    ring_hash = hashlib.sha256("\n".join(l for l in ring_lines if not l.startswith("#line ")).encode())
    for h in RING_HEADERS:
        with open(os.path.join(REF, "include", h), "rb") as f:
            ring_hash.update(f.read())
    ring_sliced = ring_hash.hexdigest()
    ring_pinned = pin is not None and pin.get("ring_sha256") == ring_sliced
    if not ring_pinned and not os.environ.get("FMD_REFERENCE_UNPINNED"):
        raise SystemExit("build_ref: the reference's ring lines / headers hash to %s, oracle/ref_pin.json says %s - not building "
                         "(FMD_REFERENCE_UNPINNED=1 builds an unpinned library)" % (ring_sliced, pin and pin.get("ring_sha256")))
    ring_head = head + ['#include "%s"' % os.path.join(REF, "include", "rtl-sdr.h")]
    text = "\n".join(ring_head + ring_lines + _shim("ref_ring_shim.c")) + "\n"
    _compile(text, ring, extra=["-I" + os.path.join(REF, "include")])   # rtl-sdr.h includes <rtl-sdr_export.h>
    with open(so, "rb") as f:
        so_hash = hashlib.sha256(f.read()).hexdigest()
    with open(ring, "rb") as f:
        ring_so_hash = hashlib.sha256(f.read()).hexdigest()
    with open(os.path.join(OUT, "libref.meta.json"), "w") as f:
        json.dump({"slices_sha256": sliced, "so_sha256": so_hash, "pinned": pinned,
                   "ring_sha256": ring_sliced, "ring_so_sha256": ring_so_hash, "ring_pinned": ring_pinned}, f)
    with open(stamp_file, "w") as f:
        f.write(stamp + "\n")
    if not quiet:
        print("build_ref: built %s and %s from %s" % (os.path.relpath(so), os.path.relpath(ring), REF))
    return True


if __name__ == "__main__":
    ok = build_ref(force="--force" in sys.argv)
    sys.exit(0 if ok else 1)
