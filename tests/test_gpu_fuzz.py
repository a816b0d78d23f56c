"""Randomised configurations under -m gpu (the gate the round-1 verdict asked for).

* a bounded fuzz: the first 100 cases of tools/fuzz_parity.py's seed 1 - exact kernels bit-identical,
  fast kernels within 1 LSB, on noise input (the hardest input for a non-bit-exact demodulator);
* the named cases on which the fast kernels of round 1 broke the +-1 LSB contract (stereo on noise:
  the 19 kHz pilot filter's envelope dips to ~1e-5 and the regenerated 38 kHz carrier was the ratio of
  two rounding errors); they are kept as regression tests of the exact redo of such samples;
* bench.py's own parity gate for every rank seed of an 8-GPU run (BASELINE.json configs[3]), on one GPU.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fuzz_cases import iter_cases, iter_cases_f, run_case  # noqa: E402

pytestmark = pytest.mark.gpu

# (seed, case) of tools/fuzz_parity.py 400 <seed> on which a fast kernel has differed from the oracle by
# more than 1 LSB at some point of the development
NAMED = [
    (1, 336),   # 48 k -> 8 k stereo, 128 taps, 65552-byte blocks: 2 LSB   (round-2 start, commit ff07075)
    (2, 170),   # 48 k -> 8 k stereo, 64 taps: 2 LSB
    (2, 361),   # 25 k -> 8 k stereo, 90 taps, no de-emphasis: 2 LSB
    (3, 380),   # 171 k -> 32 k stereo, 200 taps: 6 LSB
    (4, 96),    # 48 k -> 8 k stereo, 200 taps, offset tuning: 2 LSB
    (4, 144),   # 48 k -> 8 k stereo, 90 taps (the 45-pair kernel), 65552-byte blocks: 3 LSB
    (15, 95),   # 48 k -> 8 k stereo, 64 taps: 2 LSB with the rate-independent K (found by the seeds 5..24 soak)
    (20, 249),  # 48 k -> 8 k stereo, 32 taps, 5 streams x 12 blocks in 3 launches: 2 LSB, needed 4 x that K
]


@pytest.fixture(scope="module")
def R():
    import rtl_fm_player_amd as R
    if R.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests need a real MI355X")
    return R


def test_bounded_fuzz_100_cases(R):
    bad = []
    for c in iter_cases(100, 1, volumes=False):
        for m in run_case(R, c):
            bad.append((c["case"], m, c["kw"], c["block_len"]))
    assert not bad, bad


def test_bounded_fuzz_of_the_default_family_60_cases(R):
    """Round 6: sixty configurations FMD_MATH_FAST_MFMA_F runs (the general draw reaches them twice in 400): every family within 1 LSB, exact bit-identical,
    and the default resolves to _MFMA_F on every one of them."""
    bad, ran = [], 0
    for c in iter_cases_f(60, 1):
        fam = R.config_family(R.wbfm_config(block_len=c["block_len"], math=R.MATH_FAST, **{k: v for k, v in c["kw"].items() if k != "tau"}))
        ran += fam == R.MATH_FAST_MFMA_F
        for m in run_case(R, c):
            bad.append((c["case"], m, c["kw"], c["block_len"]))
    assert not bad, bad
    assert ran >= 55, ran


@pytest.mark.parametrize("seed,case", NAMED)
def test_named_fuzz_cases(R, seed, case):
    c = [c for c in iter_cases(case + 1, seed, volumes=False) if c["case"] == case]
    assert c, "case skipped by the generator"
    assert run_case(R, c[0]) == []


@pytest.mark.parametrize("mode", ["stereo", "mono", "nfm"])
def test_bench_parity_gate_for_all_eight_rank_seeds(R, mode):
    """What each rank of `bench.py --gpus 8` checks before timing (BASELINE.json configs[3]: 2048 streams,
    256 per GPU), for ranks 0..7 on this one GPU: bench.py's generator with seed 12345 + rank at full size
    (256 streams x 16 blocks), fast kernels, EVERY stream against the oracle: |diff| <= 1."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle import OracleStream
    BL, S, B = 262144, 256, 16
    kw = dict(rate_in=25000, rate_out2=12500, mode=1) if mode == "nfm" else \
        dict(rate_in=300000, rate_out2=48000, mode=2 if mode == "stereo" else 1)
    dev = torch.device("cuda:0")
    b = R.BatchDemod(R.wbfm_config(block_len=BL, math=R.MATH_FAST, **kw), S, device=0)
    pcm = torch.zeros((S, B, b.pcm_stride), dtype=torch.int16, device=dev)
    lens = torch.zeros((S, B), dtype=torch.int32, device=dev)
    for rank in range(8):
        iq = bench.synth_fm_iq(torch, dev, S, B * BL // 2, 200e3 if mode == "nfm" else 2.4e6, mode != "nfm",
                               12345 + rank).view(S, B, BL)
        torch.cuda.synchronize()          # iq is made on torch's stream, the kernel runs on the batch's own
        b.reset()
        b.run_device(iq, B, pcm, lens)
        b.sync()
        h_iq, h_pcm, h_lens = iq.cpu().numpy(), pcm.cpu().numpy(), lens.cpu().numpy()

        def check(s):
            want, wl = OracleStream(**kw).run(h_iq[s].reshape(-1), BL)
            if not np.array_equal(h_lens[s], wl):
                return 1 << 20
            got = np.concatenate([h_pcm[s, k, :wl[k]] for k in range(B)])
            return int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max())

        with ThreadPoolExecutor(16) as ex:
            diffs = list(ex.map(check, range(S)))
        assert max(diffs) <= 1, "mode %s rank %d stream %d: |diff| %d" % (mode, rank, diffs.index(max(diffs)), max(diffs))
        del iq


def test_bench_line_contract(R):
    """`python bench.py` (N = 1, a small batch so that it runs in seconds): ONE JSON line with the fields
    the driver reads, a live roofline (HIP events on the launch stream), the CPU baseline and the
    H2D-inclusive leg."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--preheat", "4", "--streams", "32", "--blocks", "4", "--cpu-seconds", "1.5",
                        "--e2e-streams", "8", "--e2e-jobs", "3"], capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "per_rank", "parity", "e2e_h2d"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["dtype"].startswith("f32")
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["achieved"] > 50
    assert abs(d["value"] - 32 * 4 * 131072 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.02
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 10
    assert set(cb["configs_single_thread"]) == {"mono_2.4Msps", "stereo_2.4Msps", "nfm_200ksps"}
    assert d["parity"]["max_abs_lsb"] <= 1 and d["e2e_h2d"]["value"] > 100 and d["e2e_h2d"]["pcie_gbs"] > 0.2
    assert len(d["per_rank"]) == 1 and d["per_rank"][0]["kernel_ns"] > 0
    # the extra legs (never `value`): every stream parity-checked on the worst-case and the quiet inputs, mono and NFM on the same device
    for leg in ("noise_input", "quiet_input"):
        assert d[leg]["parity"]["max_abs_lsb"] <= 1 and d[leg]["parity"]["streams_checked"] == 32 and d[leg]["kernel_ms"] > 0
    assert set(d["modes"]) == {"mono", "nfm"}
    for m in d["modes"].values():
        assert m["parity"]["max_abs_lsb"] <= 1 and 0 < m["frac"] < 1 and m["kernel_ms"] > 0
        q = m["quiet_input"]                      # the quiet-input leg of every mode (round 6)
        assert q["parity"]["max_abs_lsb"] <= 1 and q["parity"]["streams_checked"] == 32 and q["kernel_ms"] > 0 and q["slowdown_vs_timed_input"] > 0.5
    # the stereo default is the family with the composite L+R filter and the second stage at the emit instants only, and the sustained leg carries the package power / shader clock it ran at
    # where the device's hwmon files exist (they do on the MI355X boxes of the pool)
    assert d["config"]["kernel_family"] == "fast-mfma-f"
    pw = d["sustained"].get("power")
    import glob
    if glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        assert pw and 100 < pw["package_w_mean"] <= pw["package_w_max"] <= 1.05 * pw["cap_w"] and pw["sclk_mhz_mean"] > 300, pw


def test_rccl_counter_gather_runs_on_one_gpu(R):
    """The multi-GPU branch of bench.py executed where only one GPU exists: `--force-dist` initialises
    torch.distributed with backend nccl (= RCCL) at world size 1, on this rank's device, runs the barriers and the
    all_reduce(MAX) / all_gather of the counters through it, and still prints the one JSON line."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "4",
                        "--warmup", "1", "--preheat", "2", "--streams", "16", "--blocks", "2", "--no-cpu", "--no-e2e"],
                       capture_output=True, text=True, timeout=500, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["counters_gathered_over"] == "nccl" and d["n_gpus"] == 1
    assert len(d["per_rank"]) == 1 and d["per_rank"][0]["samples"] == 16 * 2 * 131072 * 4
    assert d["parity"]["max_abs_lsb"] <= 1


@pytest.mark.parametrize("name,kw,volume", [
    # the largest volumes at which the rms estimate of the fixed-point second stage still passes its gate (0.10 LSB: csrc/fmd_host.c, fixed_point_error)
    ("stereo_300k", dict(rate_in=300000, rate_out2=48000, mode=2), 7.5), ("stereo_240k", dict(rate_in=240000, rate_out2=48000, mode=2), 3.8),
    ("stereo_192k", dict(rate_in=192000, rate_out2=48000, mode=2), 3.8), ("mono_300k", dict(rate_in=300000, rate_out2=48000, mode=1), 8.8),
    ("nfm_25k", dict(rate_in=25000, rate_out2=12500, mode=1), 1.14),
])
def test_second_stage_at_the_gate_volumes(R, name, kw, volume):
    """the +-1 LSB contract of the matrix-pipe second stage is gated by an rms ESTIMATE; pin the configurations that sit right under the gate.
    32 streams x 8 blocks each of noise and of a low-amplitude carrier (where the second stage's own error is the largest share of a PCM step),
    FMD_MATH_FAST_MFMA_F against the oracle: 1 LSB at most."""
    import torch
    from oracle import OracleStream, lcg_bytes
    BL, NB, NS = 262144, 8, 32
    cfg = R.wbfm_config(block_len=BL, math=R.MATH_FAST, volume=volume, **kw)
    est = R.config_error_estimate(cfg)
    assert est["family"] == R.MATH_FAST_MFMA_F and 0.09 < max(f["rms_lsb"] for f in est["filters"]) <= 0.10, est
    rng = np.random.default_rng(1234)
    noise = np.stack([lcg_bytes(NB * BL, 900 + s)[0] for s in range(NS // 2)])
    # low-amplitude input: a few LSB of noise around the centre
    quiet = (127.5 + rng.normal(0.0, 2.0, (NS - NS // 2, NB * BL))).round().clip(0, 255).astype(np.uint8)
    iq = np.concatenate([noise, quiet]).reshape(NS, NB, BL)
    b = R.BatchDemod(cfg, NS)
    assert b.math == R.MATH_FAST_MFMA_F
    got, lens = b.run_host_concat(iq, NB)
    worst = 0
    for s in range(NS):
        want, wl = OracleStream(volume=volume, **kw).run(iq[s].reshape(-1), BL)
        assert np.array_equal(lens[s], wl)
        worst = max(worst, int(np.abs(got[s].astype(np.int32) - want.astype(np.int32)).max()))
    b.close()
    assert worst <= 1, (name, volume, worst)
