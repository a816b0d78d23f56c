#!/usr/bin/env python3
"""Randomised parity: random configurations (rates, filter sizes, modes, block lengths, block and
launch counts, stream counts) on LCG input; exact kernels must equal the oracle bit for bit, fast
kernels must stay within 1 LSB.  Usage: fuzz_parity.py [cases] [seed].  The case generator lives in
tests/fuzz_cases.py (tests/test_gpu_fuzz.py runs a bounded slice of it under -m gpu)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtl_fm_player_amd as R
from fuzz_cases import iter_cases, iter_cases_f, run_case

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
# FUZZ_F=1: the draw biased to the configurations FMD_MATH_FAST_MFMA_F runs (tests/fuzz_cases.py, iter_cases_f)
for c in (iter_cases_f(cases, seed) if os.environ.get("FUZZ_F") else iter_cases(cases, seed)):
    for math, s, d in run_case(R, c):
        bad += 1
        print("MISMATCH seed", seed, "case", c["case"], "math", math, "stream", s, "maxdiff", d, c["kw"],
              "block_len", c["block_len"], "nb", c["nb"], "launches", c["launches"], "ns", c["ns"], flush=True)
print("seed", seed, "cases", cases, "mismatches", bad)
sys.exit(1 if bad else 0)
