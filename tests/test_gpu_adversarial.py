"""Inputs CONSTRUCTED to sit on the ill-conditioned points of the chain, instead of waiting for noise to hit them.

The +-1 LSB families detect three kinds of sample and redo them in the reference's arithmetic:
  * decimated samples next to the origin (quiet input: tests/test_gpu_quiet.py);
  * discriminator samples next to the +-pi branch cut of `atan2_lagrange_f32` (src/rtl_fm_player.c:607-667);
  * stereo samples whose regenerated 38 kHz carrier `sin2atan2_f32` (:472-481) is the ratio of two small numbers - the pilot
    filter's output is at rounding level.
These generators put EVERY sample of a stream there:
  * `antiphase`: the baseband alternates sign from one decimated sample to the next, so every phase step is pi up to the last bits of
    the decimator's sums - each sample lands on the cut and the sign of a rounding error picks +pi or -pi;
  * `near_cut(eps)`: a tone whose phase advances by pi - eps per decimated sample, eps from 1e-2 down to 1e-6;
  * `carrier_only`: an unmodulated carrier (with a frequency offset): the discriminator output is a constant, the 19 kHz pilot filter's
    output is that constant times its DC gain - rounding level - at every sample: 100 % of the stereo samples take the carrier redo;
  * `mono_station`: a mono FM broadcast (no pilot) decoded in stereo mode - the realistic version of the same.
Exact kernels: bit-identical.  Fast families: within 1 LSB."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BL = 262144
STEREO = dict(rate_in=300000, rate_out2=48000, mode=2)
MONO = dict(rate_in=300000, rate_out2=48000, mode=1)
NFM = dict(rate_in=25000, rate_out2=12500, mode=1)


@pytest.fixture(scope="module")
def R():
    import rtl_fm_player_amd as R
    if R.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests need a real MI355X")
    return R


def to_iq(z, amp=100.0, dither=None):
    """complex baseband (after the fs/4 shift the path applies) -> the u8 IQ bytes a dongle would deliver: multiply by j^-n"""
    n = np.arange(z.size)
    raw = z * np.exp(-0.5j * np.pi * (n & 3))
    i = 127.5 + amp * raw.real
    q = 127.5 + amp * raw.imag
    if dither is not None:
        i = i + dither[0::2]
        q = q + dither[1::2]
    out = np.empty(2 * z.size, np.uint8)
    out[0::2] = np.clip(np.rint(i), 0, 255)
    out[1::2] = np.clip(np.rint(q), 0, 255)
    return out


def make(kind, n_bytes, seed=3):
    rng = np.random.default_rng(seed)
    n = n_bytes // 2
    k = np.arange(n)
    if kind == "antiphase":                       # + - + - ... per decimated sample (8 IQ samples)
        return to_iq(np.where((k // 8) & 1, -1.0, 1.0).astype(np.complex128))
    if kind == "antiphase_dithered":              # the same with +-1 LSB of ADC noise: the rounding that decides +pi / -pi differs per sample
        return to_iq(np.where((k // 8) & 1, -1.0, 1.0).astype(np.complex128), dither=rng.uniform(-0.8, 0.8, 2 * n))
    if kind.startswith("near_cut_"):              # phase step pi - eps per decimated sample
        eps = float(kind.split("_")[2])
        return to_iq(np.exp(1j * (np.pi - eps) * (k / 8.0)), dither=rng.uniform(-0.5, 0.5, 2 * n))
    if kind.startswith("carrier_"):               # unmodulated carrier, offset in Hz at 2.4 Msps
        f0 = float(kind.split("_")[1])
        return to_iq(np.exp(2j * np.pi * f0 * k / 2.4e6))
    if kind == "mono_station":                    # FM, 1 kHz + 3.3 kHz tones, +-60 kHz deviation, no pilot, a little ADC noise
        t = k / 2.4e6
        msg = 0.6 * np.sin(2 * np.pi * 1000 * t) + 0.4 * np.sin(2 * np.pi * 3300 * t)
        ph = 2 * np.pi * 60e3 * np.cumsum(msg) / 2.4e6
        return to_iq(np.exp(1j * ph), dither=rng.uniform(-1.0, 1.0, 2 * n))
    raise ValueError(kind)


def run(R, kw, iq, nb, math, ns=1):
    from oracle import OracleStream
    b = R.BatchDemod(R.wbfm_config(block_len=BL, math=math, **kw), ns)
    got, lens = b.run_host_concat(np.ascontiguousarray(iq.reshape(ns, nb, BL)), nb)
    out = []
    for s in range(ns):
        want, wl = OracleStream(**kw).run(iq.reshape(ns, -1)[s], BL)
        assert np.array_equal(lens[s], wl)
        out.append((got[s], want))
    return out


CUT = ["antiphase", "antiphase_dithered", "near_cut_1e-2", "near_cut_1e-4", "near_cut_1e-6"]
PILOTLESS = ["carrier_0", "carrier_10000", "carrier_-37500", "mono_station"]


@pytest.mark.parametrize("kind", CUT + PILOTLESS)
@pytest.mark.parametrize("cfg", [STEREO, MONO, NFM], ids=["stereo", "mono", "nfm"])
def test_constructed_inputs_exact_kernels_bit_identical(R, cfg, kind):
    nb = 3
    for got, want in run(R, cfg, make(kind, nb * BL), nb, R.MATH_EXACT):
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s: first mismatch at %d: gpu %d oracle %d" % (kind, bad[0], got[bad[0]], want[bad[0]])


@pytest.mark.parametrize("kind", CUT + PILOTLESS)
@pytest.mark.parametrize("cfg", [STEREO, MONO, NFM], ids=["stereo", "mono", "nfm"])
def test_constructed_inputs_fast_families_within_one_lsb(R, cfg, kind, fast_math):
    nb = 3
    for got, want in run(R, cfg, make(kind, nb * BL), nb, fast_math):
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1, "%s: max |diff| %d at %d (gpu %d oracle %d); %d values differ" % (
            kind, d.max(), int(d.argmax()), got[d.argmax()], want[d.argmax()], int((d > 0).sum()))


@pytest.mark.parametrize("kind", ["antiphase_dithered", "mono_station", "carrier_10000"])
def test_constructed_inputs_on_a_filled_device(R, kind, fast_math):
    """64 streams x 8 blocks, each stream its own dither: time chunks, replayed tiles and the hand-over between launches meet the cold
    paths at full rate."""
    ns, nb = 64, 8
    iq = np.concatenate([make(kind, nb * BL, seed=50 + s) for s in range(ns)])
    worst = 0
    for got, want in run(R, STEREO, iq, nb, fast_math, ns=ns):
        worst = max(worst, int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max()))
    assert worst <= 1, worst
