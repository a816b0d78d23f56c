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
    """On a box without (enough) GPUs the real run refuses: up front from the driver's topology in sysfs where that is
    readable (the launcher itself never opens a HIP device), otherwise when its ranks find no device."""
    sys.path.insert(0, ROOT)
    import bench
    have = bench.kfd_gpu_count()
    n = (have or 0) + 2
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu", "--no-e2e"])
    assert r.returncode != 0
    if have is not None:
        assert "--gpus %d but this node shows %d HIP device" % (n, have) in r.stderr
    else:
        assert "rank exit codes" in r.stderr


def test_launcher_does_not_import_torch():
    """The parent of the ranks must not initialise HIP: no torch import on the launcher path."""
    code = ("import sys; sys.argv = ['bench.py', '--gpus', '2', '--dry-run', '--steps', '1']; import bench; "
            "bench.subprocess.Popen = lambda *a, **k: (_ for _ in ()).throw(SystemExit('torch' in sys.modules))")
    r = subprocess.run([sys.executable, "-c", code + "; bench.main()"], capture_output=True, text=True, cwd=ROOT, timeout=120,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, (r.stdout, r.stderr)          # SystemExit(False) -> 0: torch was not imported


def test_gpus_must_match_world_size():
    r = _run(["--gpus", "4", "--dry-run"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 4 but WORLD_SIZE is 2" in r.stderr


def test_failing_rank_fails_the_launch():
    """A rank that dies for real (rank 1 exits with 3 before the rendezvous) makes the launcher stop the other rank -
    which would otherwise wait in the rendezvous for its timeout - and exit non-zero, promptly, leaving no child."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run"], env={"FMD_BENCH_DRYRUN_FAIL_RANK": "1"},
             timeout=120)
    assert r.returncode != 0
    assert time.time() - t0 < 90
    assert "rank 1 failed" in r.stderr and "3" in r.stderr
    pids = [int(x) for l in r.stderr.splitlines() if l.startswith("bench.py: ranks ") for x in l.split()[2:]]
    assert len(pids) == 2
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except ProcessLookupError:
            alive = False
        assert not alive, "rank process %d left behind" % pid


def test_gpus_below_one_is_rejected():
    r = _run(["--gpus", "0", "--dry-run"])
    assert r.returncode != 0


def _ranks_by_hand(world, env_of_rank):
    """What torch.distributed.run (or a scheduler) does: one process per rank with RANK / LOCAL_RANK / WORLD_SIZE set."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    procs = []
    for k in range(world):
        e = dict(os.environ)
        e.update({"RANK": str(k), "LOCAL_RANK": str(k), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port})
        e.update(env_of_rank(k))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", str(world), "--steps", "1", "--warmup", "0", "--dry-run"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e))
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [0] * world, [o[1][-400:] for o in outs]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_one_visible_device_per_rank_binds_device_zero():
    """A launcher that narrows every rank to ONE visible device (HIP_VISIBLE_DEVICES=k) still sets LOCAL_RANK=k: ranks 1..N-1
    must bind the device they see (index 0), not exit because LOCAL_RANK >= device count."""
    out = _ranks_by_hand(2, lambda k: {"HIP_VISIBLE_DEVICES": str(k)})
    assert out["bound_devices"] == [0, 0] and out["n_gpus"] == 2
    assert [p["checksum"] for p in out["per_rank"]] == [0, 256]          # the stream shards do not depend on the binding


def test_all_devices_visible_binds_local_rank():
    out = _ranks_by_hand(2, lambda k: {"HIP_VISIBLE_DEVICES": "0,1"})
    assert out["bound_devices"] == [0, 1]


def test_pick_device_rules():
    sys.path.insert(0, ROOT)
    import bench
    import pytest
    assert bench.pick_device(3, 8, 8) == 3
    assert bench.pick_device(3, 8, 1) == 0
    assert bench.pick_device(5, 8, 4) == 1          # two ranks per visible device: still a device that exists
    with pytest.raises(SystemExit):
        bench.pick_device(1, 1, 1)                  # a single rank asking for a device that is not there
    with pytest.raises(SystemExit):
        bench.pick_device(0, 2, 0)
