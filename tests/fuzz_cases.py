"""Randomised parity cases shared by tools/fuzz_parity.py and tests/test_gpu_fuzz.py.

iter_cases(cases, seed) reproduces, draw for draw, the sequence tools/fuzz_parity.py has used since
round 1 (so `fuzz_parity.py 400 {1,2,3,4}` names the same 1 600 configurations the round-1 verdict
counted mismatches on).  run_case() demodulates one case with the exact and the fast kernels through
the C ABI and returns the mismatches against the CPU oracle: exact must be bit-identical, fast within
1 LSB (BASELINE.json north_star)."""
import os
import random

import numpy as np


def iter_cases(cases, seed, volumes=None):
    if volumes is None:
        volumes = bool(os.environ.get("FUZZ_VOLUMES"))
    rng = random.Random(seed)
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
                  offset_tuning=rng.random() < 0.2, volume=rng.choice([0.4, 1.0, 3.0]) if volumes else 0.4,
                  tau=rng.choice([50e-6, 75e-6, 300e-6]))
        if mode == 2 and rate_out2 == 0:
            continue
        block_len = 16 * rng.choice([4, 5, 33, 64, 100, 512, 513, 1000, 2048, 4097, 16384])
        launches = rng.choice([1, 1, 2, 3])
        nb = launches * rng.randint(1, 4)
        ns = rng.choice([1, 1, 2, 5])
        yield dict(case=case, kw=kw, block_len=block_len, nb=nb, launches=launches, ns=ns, seed0=1000 + case * 7)


def iter_cases_f(cases, seed):
    """Round 6: configurations FMD_MATH_FAST_MFMA_F runs (the draw of iter_cases reaches them twice in 400 cases) - 90-tap stereo at 300 k / 240 k / 192 k -> 48 k
    and 128-tap mono at rates whose sixteen frames are a whole number of samples, whole tiles, volumes up to the error estimate's gate, everything else
    random (de-emphasis, tau, offset tuning, blocks, launches, streams)."""
    rng = random.Random(seed * 7919 + 17)
    stereo = [(300000, 48000, 7.5), (240000, 48000, 3.8), (192000, 48000, 3.8)]
    mono = [(300000, 48000, 8.8), (240000, 48000, 4.5), (192000, 48000, 4.5), (25000, 12500, 1.14), (96000, 32000, 1.0), (384000, 48000, 8.0), (96000, 48000, 2.0)]
    for case in range(cases):
        if rng.random() < 0.6:
            rate_in, rate_out2, vmax = rng.choice(stereo)
            mode, size = 2, 90
        else:
            rate_in, rate_out2, vmax = rng.choice(mono)
            mode, size = 1, 128
        volume = rng.choice([0.4, 0.4, 1.0, vmax, round(rng.uniform(0.1, vmax), 2)])
        kw = dict(rate_in=rate_in, rate_out2=rate_out2, mode=mode, size=size, deemph=rng.random() < 0.8, offset_tuning=rng.random() < 0.2,
                  volume=volume, tau=rng.choice([50e-6, 75e-6, 300e-6]))
        block_len = rng.choice([8192, 16384, 32768, 65536, 262144, 262144])
        launches = rng.choice([1, 1, 2, 3])
        nb = launches * rng.randint(1, 4 if block_len >= 65536 else 12)
        ns = rng.choice([1, 1, 2, 5])
        yield dict(case=case, kw=kw, block_len=block_len, nb=nb, launches=launches, ns=ns, seed0=5000 + case * 11)


def run_case(R, c, maths=None):
    """Returns a list of (math, stream, maxdiff) mismatches; [] when the case holds.  A configuration the
    library refuses (documented limits) or the oracle rejects counts as held."""
    from oracle import OracleStream, lcg_bytes
    kw, block_len, nb, launches, ns = c["kw"], c["block_len"], c["nb"], c["launches"], c["ns"]
    try:
        ref = [OracleStream(**kw).run(lcg_bytes(nb * block_len, c["seed0"] + s)[0], block_len) for s in range(ns)]
    except ValueError:
        return []
    iq = np.concatenate([lcg_bytes(nb * block_len, c["seed0"] + s)[0] for s in range(ns)]).reshape(ns, nb, block_len)
    bad = []
    for math, tol in [(R.MATH_EXACT, 0)] + [(m, 1) for m in R.FAST_MATHS]:
        if maths is not None and math not in maths:
            continue
        try:
            b = R.BatchDemod(R.wbfm_config(block_len=block_len, math=math, **kw), ns)
        except R.FmdError:
            return []
        per = nb // launches
        outs = [[] for _ in range(ns)]
        lens_all = []
        for l in range(launches):
            out, lens = b.run_host_concat(np.ascontiguousarray(iq[:, l * per:(l + 1) * per]), per)
            for s in range(ns):
                outs[s].append(out[s])
            lens_all.append(lens)
        b.close()
        lens_all = np.concatenate(lens_all, axis=1)
        for s in range(ns):
            got = np.concatenate(outs[s])
            want, wl = ref[s]
            ok = np.array_equal(lens_all[s], wl) and got.size == want.size and \
                (got.size == 0 or np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= tol)
            if not ok:
                d = -1 if got.size != want.size else int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max())
                bad.append((math, s, d))
    return bad
