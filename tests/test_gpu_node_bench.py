"""The node-level C driver (csrc/fmd_node_bench.c) on the one device a box of this pool has: a demod thread bound to the device's NUMA node,
the resident and the H2D-inclusive legs, and the counters gathered with ncclAllGather from librccl directly (world of one communicator).
What it would print for eight devices is unmeasured."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_node_bench_on_one_device():
    import rtl_fm_player_amd as R
    if R.device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests need a real MI355X")
    exe = os.path.join(os.path.dirname(R.library_path()), "fmd_node_bench")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, "-d", "1", "-s", "64", "-B", "4", "-K", "6", "-W", "2", "-J", "3", "-T", "4"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["streams"] == 64 and d["scaling"] == "weak" and d["data_path_collectives"] == 0
    assert "ncclAllGather" in d["counters_gathered_by"]
    dev = d["per_device"][0]
    assert dev["streams"] == 64 and dev["first_stream"] == 0 and dev["math_run"] == R.MATH_FAST_MFMA_F
    samples = 6 * 64 * 4 * 131072
    assert abs(d["value"] - samples / (dev["ms_per_launch"] * 6e-3) / 1e6) / d["value"] < 0.02
    # 64 stereo streams x 4 blocks x ~5243 PCM values per block and launch
    assert 0.99 < dev["pcm_values"] / (6 * 64 * 4 * 5243.0) < 1.01 and 0.99 < dev["h2d_pcm_values"] / (3 * 64 * 4 * 5243.0) < 1.01
    assert d["h2d_value"] > 100 and d["h2d_pcie_gbs"] > 0.2 and d["value"] > d["h2d_value"]
