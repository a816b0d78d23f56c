#!/usr/bin/env python3
"""Design study (CPU only, not part of the product): would an overlap-save FFT form of stage C hold the
+-1 LSB contract?

Stage C is three 90-tap FIRs on the discriminator output at rate_in: 45 pair sums and
135 multiply-adds per sample, half of the headline kernel's instructions.  In blocks of N = 2048 samples with 89 of
overlap the same three outputs cost one real FFT, three spectral products and three inverse FFTs, about 75 flop per
sample - but every output then carries the rounding of an fp32 FFT instead of a 90-term sum.

This script restates stages C..F (src/rtl_fm_player.c:533-735) in numpy float32 with the reference's operation order,
checks that restatement bit for bit against the oracle, then swaps the three direct FIRs for single-precision FFT
convolutions (numpy's pocketfft works in float32 for float32 input) and counts the PCM differences.

    python tools/fft_mpx_study.py [--blocks 8] [--rate-in 192000]

rate_in 192000 is used because block-start quirk Q1 never fires there (4 : 1), so the discriminator output of the
oracle's trace is what the filters see.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import OracleStream, dds_bytes, lcg_bytes  # noqa: E402

F32 = np.float32


def fir3_direct(v, taps):
    """bm, vp, vs in the reference's order: p = oldest + newest, acc += t[k] * p, k ascending."""
    n = v.size - 89
    out = [np.zeros(n, F32) for _ in range(3)]
    for k in range(45):
        p = v[k:k + n] + v[89 - k:89 - k + n]
        for o, t in zip(out, taps):
            o += (t[k] * p).astype(F32)          # float32 * float32 -> float32, then float32 add
    return out


def fir3_fft(v, taps, nfft):
    """The same three convolutions by overlap-save, everything in single precision."""
    n = v.size - 89
    step = nfft - 89
    full = []
    for t in taps:
        h = np.zeros(nfft, np.float64)
        ft = np.concatenate([t, t[::-1]]).astype(np.float64)      # F[j] = t[min(j, 89 - j)]
        h[:90] = ft
        full.append(np.fft.rfft(h).astype(np.complex64))
    out = [np.zeros(n, F32) for _ in range(3)]
    for s in range(0, n, step):
        seg = np.zeros(nfft, F32)
        m = min(nfft, v.size - s)
        seg[:m] = v[s:s + m]
        X = np.fft.rfft(seg)                                       # complex64
        assert X.dtype == np.complex64
        for o, H in zip(out, full):
            y = np.fft.irfft(X * H, nfft)
            assert y.dtype == F32
            cnt = min(step, n - s)
            o[s:s + cnt] = y[89:89 + cnt]
    return out


def back_end(bm, vp, vs, tp, cfg):
    """carrier, L-R, second-stage FIRs at the emits, de-emphasis, s16 - the reference's arithmetic."""
    swf, cwf, fm = F32(tp["swf"]), F32(tp["cwf"]), tp["fm"]
    vq = np.concatenate([[F32(0)], vp[:-1]])
    x = vp * swf
    y = vp * cwf - vq
    with np.errstate(divide="ignore", invalid="ignore"):
        z = y / x
        car = np.where(x == 0, F32(0), (z + z) / (F32(1) + z * z)).astype(F32)
    bs = vs * car
    n = bm.size
    slow, fast = cfg["rate_out2"], cfg["rate_in"]
    idx = np.arange(n, dtype=np.int64)
    emit = ((idx + 1) * slow) // fast > (idx * slow) // fast
    e = np.nonzero(emit)[0]
    bmh = np.concatenate([np.zeros(89, F32), bm])
    bsh = np.concatenate([np.zeros(89, F32), bs])
    om = np.zeros(e.size, F32)
    os_ = np.zeros(e.size, F32)
    for k in range(45):
        om += (fm[k] * (bmh[e + k] + bmh[e + 89 - k])).astype(F32)
        os_ += (fm[k] * (bsh[e + k] + bsh[e + 89 - k])).astype(F32)
    L, R = om + os_, om - os_
    lam = F32(cfg["lam"])
    pcm = np.empty(2 * e.size, np.int16)
    for c, x_ in enumerate((L, R)):
        yv = np.empty_like(x_)
        prev = F32(0)
        for i in range(x_.size):
            d = F32(prev - x_[i])
            prev = F32(x_[i] + F32(lam * d))
            yv[i] = prev
        t = yv * F32(cfg["coef"])
        pcm[c::2] = np.rint(np.clip(t, -32768.0, 32767.0)).astype(np.int16)
    return pcm, car, x, y, vs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=8)
    ap.add_argument("--rate-in", type=int, default=192000)
    ap.add_argument("--nfft", type=int, default=2048)
    a = ap.parse_args()
    BL = 262144
    kw = dict(rate_in=a.rate_in, rate_out2=48000, mode=2)
    for name, iq in (("FM broadcast (DDS)", dds_bytes(a.blocks * BL, fs=8 * a.rate_in)),
                     ("noise (LCG bytes)", lcg_bytes(a.blocks * BL, 12345)[0])):
        st = OracleStream(**kw)
        tp = st.taps()
        vs_, pcm_ref = [], []
        for b in range(a.blocks):
            p, tr = st.block(iq[b * BL:(b + 1) * BL], trace=True)
            vs_.append(tr["v"].copy())
            pcm_ref.append(p)
        pcm_ref = np.concatenate(pcm_ref)
        v = np.concatenate([np.zeros(89, F32)] + vs_)
        cfg = dict(rate_in=a.rate_in, rate_out2=48000, lam=st.cfg.deemph_lambda, coef=F32(st.cfg.volume) * F32(32768.0))
        taps = (tp["fm"], tp["fp"], tp["fs"])
        bm, vp, vs = fir3_direct(v, taps)
        pcm_d, car, x, y, _ = back_end(bm, vp, vs, tp, cfg)
        same = pcm_d.size == pcm_ref.size and np.array_equal(pcm_d, pcm_ref)
        print("%s: numpy restatement == oracle: %s (%d PCM values)" % (name, same, pcm_ref.size))
        bm2, vp2, vs2 = fir3_fft(v, taps, a.nfft)
        for nm, d_, f_ in (("bm", bm, bm2), ("vp", vp, vp2), ("vs", vs, vs2)):
            print("   %s: max |fft - direct| = %.3g (rms of the signal %.3g)" % (nm, float(np.abs(d_ - f_).max()), float(np.sqrt(np.mean(d_.astype(np.float64) ** 2)))))
        pcm_f, _, _, _, _ = back_end(bm2, vp2, vs2, tp, cfg)
        diff = np.abs(pcm_f.astype(np.int32) - pcm_ref.astype(np.int32))
        print("   FFT stage C, everything else exact: PCM max |diff| %d LSB, %d of %d values differ, %d by more than 1"
              % (int(diff.max()), int((diff > 0).sum()), diff.size, int((diff > 1).sum())))
        # with the shipped kernels' safety net: samples whose carrier is ill-conditioned are redone exactly
        r2 = x.astype(np.float64) ** 2 + y.astype(np.float64) ** 2
        gmax = float(np.abs(tp["fm"]).max())
        for scale in (1.0, 4.0, 16.0):
            K = scale * 12.0 * 1e-7 * float(cfg["coef"]) * gmax
            frag = r2 < (K * np.abs(vs.astype(np.float64))) ** 2
            vs3 = np.where(frag, vs, vs2)
            vp3 = vp2.copy()
            vp3[frag] = vp[frag]
            prev = np.roll(frag, -1)            # the carrier of n uses vp[n-1] as well
            vp3[prev] = vp[prev]
            pcm_g, _, _, _, _ = back_end(bm2, vp3, vs3, tp, cfg)
            dg = np.abs(pcm_g.astype(np.int32) - pcm_ref.astype(np.int32))
            print("   ... with the exact redo at K x %-4g (%d of %d samples redone): max |diff| %d LSB, %d values beyond 1"
                  % (scale, int(frag.sum()), frag.size, int(dg.max()), int((dg > 1).sum())))


if __name__ == "__main__":
    main()
