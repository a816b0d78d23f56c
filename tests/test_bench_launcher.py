"""`python bench.py --gpus N` must really run N ranks (BASELINE configs[3]).

The launcher is exercised here without a GPU: --dry-run makes every rank skip the device work
and use gloo for the counter gather, everything else (child processes, rank environment,
rendezvous on 127.0.0.1, rank 0's JSON line forwarded, exit codes) is the real code path.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=timeout)


def test_self_launch_two_ranks_gloo_dry_run():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert len(out["per_rank"]) == 2
    assert [p["checksum"] for p in out["per_rank"]] == [0, 256]          # each rank owns its own streams
    assert [p["kernel_ns"] for p in out["per_rank"]] == [1000, 1001]
    per = 256 * 16 * 131072 * 3
    assert [p["samples"] for p in out["per_rank"]] == [per, per]
    assert abs(out["value"] - 2 * per / 1.25 / 1e6) < 1e-3              # whole job / slowest rank


def test_gpus_beyond_device_count_fails_loudly():
    """On a box without (enough) GPUs the real run refuses up front, with the reason."""
    import torch
    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "--gpus %d but this node shows %d HIP device" % (have + 2, have) in r.stderr


def test_gpus_must_match_world_size():
    r = _run(["--gpus", "4", "--dry-run"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 4 but WORLD_SIZE is 2" in r.stderr


def test_failing_rank_fails_the_launch():
    """A child that dies makes the launcher exit non-zero (here: rank environment the ranks reject)."""
    r = _run(["--gpus", "0", "--dry-run"])
    assert r.returncode != 0
