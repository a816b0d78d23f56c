"""The N>1 path on CPU: world_size-2 gloo processes run the sharding + counter gather that
bench.py uses under RCCL (the data path itself has no collective)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rtl_fm_player_amd.shard import gather_counters, shard_streams


def test_shard_streams_partition():
    for total in (1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            parts = [shard_streams(total, world, r) for r in range(world)]
            assert parts[0][0] == 0
            assert sum(c for _, c in parts) == total
            for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                assert f1 == f0 + c0
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard_streams(2048, world, rank)
    # each rank "processes" its own streams; rank 1 is slower
    rep = gather_counters(dist, torch.device("cpu"), elapsed_s=1.0 + rank, samples=count * 1000,
                          kernel_ns=5 + rank, checksum=first)
    out[rank] = rep
    dist.barrier()
    dist.destroy_process_group()


def test_gather_counters_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] == out[1]                      # every rank sees the same whole-job view
    rep = out[0]
    assert rep["world"] == 2
    assert rep["elapsed_s"] == 2.0               # MAX over ranks
    assert rep["samples"] == 2048 * 1000         # whole job
    assert [r["checksum"] for r in rep["per_rank"]] == [0, 1024]
    assert [r["kernel_ns"] for r in rep["per_rank"]] == [5, 6]


def test_single_process_passthrough():
    rep = gather_counters(None, torch.device("cpu"), 0.5, 10, 3, 7)
    assert rep == {"world": 1, "elapsed_s": 0.5, "samples": 10,
                   "per_rank": [{"samples": 10, "kernel_ns": 3, "checksum": 7}]}


def _worker_forced(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out[rank] = gather_counters(dist, torch.device("cpu"), 0.5, 10, 3, 7, force=True)
    dist.barrier()
    dist.destroy_process_group()


def test_forced_collectives_with_one_rank():
    """bench.py --force-dist: with one rank the gather still goes through all_reduce / all_gather (gloo here, RCCL under
    -m gpu) and reports the backend."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_forced, args=(1, port, out), nprocs=1, join=True)
    assert out[0] == {"world": 1, "elapsed_s": 0.5, "samples": 10, "backend": "gloo",
                      "per_rank": [{"samples": 10, "kernel_ns": 3, "checksum": 7}]}


def test_node_bench_plan_matches_the_python_sharding():
    """The C node driver (csrc/fmd_node_bench.c -P: no device touched) places streams like shard.py where the split is even, contiguously
    and completely everywhere, and refuses a device without streams; BASELINE.json configs[3] = 2048 streams on 8 devices = 256 each."""
    import json
    import subprocess
    import rtl_fm_player_amd as R
    exe = os.path.join(os.path.dirname(R.library_path()), "fmd_node_bench")
    if not os.path.exists(exe):
        R.build_library()
    for total, world in ((2048, 8), (2000, 8), (256, 1), (7, 2), (2049, 8)):
        r = subprocess.run([exe, "-P", "-d", str(world), "-s", str(total)], capture_output=True, text=True, timeout=30)
        assert r.returncode == 0, r.stderr
        d = json.loads(r.stdout)
        sh = d["shards"]
        assert d["devices"] == world == len(sh) and d["streams"] == total and d["data_path_collectives"] == 0
        assert sh[0]["first_stream"] == 0 and sum(x["streams"] for x in sh) == total
        for a, b in zip(sh, sh[1:]):
            assert b["first_stream"] == a["first_stream"] + a["streams"]
        if total % world == 0:
            assert [(x["first_stream"], x["streams"]) for x in sh] == [shard_streams(total, world, r_) for r_ in range(world)]
    assert sh and subprocess.run([exe, "-P", "-d", "3", "-s", "2"], capture_output=True, text=True, timeout=30).returncode == 2
