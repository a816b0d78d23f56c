"""Two batches of the +-1 LSB families on ONE device at the same time.

Every batch launches on its own HIP stream, so a process that demodulates two station groups (say a 300 k and a
240 k batch, or a stereo and a mono one) has waves of two different fused kernels resident on the same SIMDs.
Round 3 found the fast families computing wrong PCM beside a neighbour wave that issues 128-bit-operand MFMAs
(profiles/archive/r03m_mfma_neighbour.txt) - and the default family issues exactly such an instruction itself
(v_mfma_i32_16x16x64_i8, 36 per tile).  This is the product's own way into that situation: nothing is
synchronised between the two batches' launches, every launch of every stream is held against the oracle
(the reference's arithmetic, src/rtl_fm_player.c:758-788; tolerance +-1 LSB, BASELINE.json).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BL = 262144
S, B, NL = 64, 2, 200          # streams per batch, blocks per launch, launches (both grids fit the chip together)

STEREO_300 = dict(rate_in=300000, rate_out2=48000, mode=2)
STEREO_240 = dict(rate_in=240000, rate_out2=48000, mode=2)
MONO_300 = dict(rate_in=300000, rate_out2=48000, mode=1)


@pytest.fixture(scope="module")
def R():
    import rtl_fm_player_amd as R
    if R.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests need a real MI355X")
    return R


def _inputs(seed0):
    from oracle import lcg_bytes
    return np.stack([lcg_bytes(B * BL, seed0 + s)[0] for s in range(S)])      # [S, B * BL]: the same B blocks every launch


_ORACLE_CACHE = {}


def _oracle_all(kw, seed0, host):
    """PCM of NL launches of the same B blocks, state carried, per stream: list of (pcm, lens[NL * B])."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import OracleStream
    key = (tuple(sorted(kw.items())), seed0)
    if key not in _ORACLE_CACHE:
        def one(s):
            return OracleStream(**kw).run(np.tile(host[s], NL), BL)

        with ThreadPoolExecutor(16) as ex:
            _ORACLE_CACHE[key] = list(ex.map(one, range(S)))
    return _ORACLE_CACHE[key]


def _check(name, want, pcm, lens):
    p_all, l_all = pcm.cpu().numpy(), lens.cpu().numpy()       # [NL, S, B, stride], [NL, S, B]
    worst, where, bad_launches = 0, None, set()
    for s in range(S):
        w, wl = want[s]
        got_l = l_all[:, s, :].reshape(-1)
        assert np.array_equal(got_l, wl), "%s stream %d: block lengths differ" % (name, s)
        off = np.concatenate([[0], np.cumsum(wl)])
        for k in range(NL):
            for j in range(B):
                i = k * B + j
                d = int(np.abs(p_all[k, s, j, :wl[i]].astype(np.int32) - w[off[i]:off[i + 1]].astype(np.int32)).max())
                if d > 1:
                    bad_launches.add(k)
                if d > worst:
                    worst, where = d, (s, k, j)
    return worst, where, len(bad_launches)


@pytest.mark.parametrize("pair", ["stereo300+stereo240", "stereo300+mono300"])
@pytest.mark.parametrize("family", ["valu", "mfma", "default"])
def test_two_fast_batches_on_two_streams(R, pair, family):
    import torch
    math = {"valu": R.MATH_FAST_VALU, "mfma": R.MATH_FAST_MFMA, "default": R.MATH_FAST}[family]
    kws = (STEREO_300, STEREO_240 if pair.endswith("stereo240") else MONO_300)
    dev = torch.device("cuda:0")
    seeds = (41000, 42000)
    hosts = [_inputs(sd) for sd in seeds]
    wants = [_oracle_all(kw, sd, h) for kw, sd, h in zip(kws, seeds, hosts)]
    iqs = [torch.from_numpy(h).to(dev) for h in hosts]
    batches = [R.BatchDemod(R.wbfm_config(math=math, **kw), S) for kw in kws]
    for b in batches:
        b.set_timing(False)                                   # no event pair per launch: launches go out back to back
        b.set_time_split(6)                                   # each grid takes half of a CU's worker slots: both kernels resident together
    pcms = [torch.zeros((NL, S, B, b.pcm_stride), dtype=torch.int16, device=dev) for b in batches]
    lens = [torch.zeros((NL, S, B), dtype=torch.int32, device=dev) for b in batches]
    torch.cuda.synchronize()
    for k in range(NL):                                        # interleaved, never synchronised: each on its batch's own stream
        for i, b in enumerate(batches):
            b.run_device(iqs[i], B, pcms[i][k], lens[i][k])
    for b in batches:
        b.sync()
    for i, b in enumerate(batches):
        worst, where, nbad = _check(pair.split("+")[i], wants[i], pcms[i], lens[i])
        assert worst <= 1, "%s (family %d) beside the other batch: |diff| %d LSB at (stream, launch, block) %s, %d of %d launches off" % (
            pair.split("+")[i], b.math, worst, where, nbad, NL)
    for b in batches:
        b.close()
