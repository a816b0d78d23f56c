#!/usr/bin/env python3
"""Randomised parity: random configurations (rates, filter sizes, modes, block lengths, block and
launch counts, stream counts) on LCG input; exact kernels must equal the oracle bit for bit, fast
kernels must stay within 1 LSB.  Usage: fuzz_parity.py [cases] [seed]."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtl_fm_player_amd as R
from oracle import OracleStream, lcg_bytes

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    mode = rng.choice([0, 1, 2, 2, 2])
    rate_in = rng.choice([300000, 240000, 192000, 171000, 96000, 25000, 48000])
    if mode == 2:
        rate_out2 = rng.choice([r for r in (48000, 44100, 32000, 24000, 8000) if 3 * r <= rate_in] or [rate_in // 4])
    else:
        rate_out2 = rng.choice([0, rate_in // 2, 48000 if rate_in >= 48000 else rate_in, rate_in // 3, 12500])
        rate_out2 = min(rate_out2, rate_in)
    size = rng.choice([90, 90, 128, 64, 32, 200]) if mode else 90
    if mode == 1 and rng.random() < 0.5:
        size = 128
    kw = dict(rate_in=rate_in, rate_out2=rate_out2, mode=mode, size=size, deemph=rng.random() < 0.8,
              offset_tuning=rng.random() < 0.2, volume=rng.choice([0.4, 1.0, 3.0]) if os.environ.get("FUZZ_VOLUMES") else 0.4, tau=rng.choice([50e-6, 75e-6, 300e-6]))
    if mode == 2 and rate_out2 == 0:
        continue
    block_len = 16 * rng.choice([4, 5, 33, 64, 100, 512, 513, 1000, 2048, 4097, 16384])
    launches = rng.choice([1, 1, 2, 3])
    nb = launches * rng.randint(1, 4)
    ns = rng.choice([1, 1, 2, 5])
    try:
        ref = [OracleStream(**kw).run(lcg_bytes(nb * block_len, 1000 + case * 7 + s)[0], block_len) for s in range(ns)]
    except ValueError:
        continue
    iq = np.concatenate([lcg_bytes(nb * block_len, 1000 + case * 7 + s)[0] for s in range(ns)]).reshape(ns, nb, block_len)
    for math, tol in ((R.MATH_EXACT, 0), (R.MATH_FAST, 1)):
        try:
            b = R.BatchDemod(R.wbfm_config(block_len=block_len, math=math, **kw), ns)
        except Exception as e:                     # configurations the library refuses (documented limits)
            print("case", case, "refused:", str(e)[:80])
            break
        per = nb // launches
        outs = [[] for _ in range(ns)]
        lens_all = []
        for l in range(launches):
            out, lens = b.run_host_concat(np.ascontiguousarray(iq[:, l * per:(l + 1) * per]), per)
            for s in range(ns):
                outs[s].append(out[s])
            lens_all.append(lens)
        lens_all = np.concatenate(lens_all, axis=1)
        for s in range(ns):
            got = np.concatenate(outs[s])
            want, wl = ref[s]
            ok = np.array_equal(lens_all[s], wl) and got.size == want.size and \
                (got.size == 0 or np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= tol)
            if not ok:
                bad += 1
                d = -1 if got.size != want.size else int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max())
                print("MISMATCH case", case, "math", math, "stream", s, "maxdiff", d, kw, "block_len", block_len, "nb", nb,
                      "launches", launches, flush=True)
print("cases", cases, "mismatches", bad)
sys.exit(1 if bad else 0)
