"""Multi-GPU bookkeeping of the path: streams shard across ranks, nothing else.

Streams are independent (no cross-stream term in the reference's chain), so the data path has
no collective.  RCCL (torch.distributed "nccl") - or gloo in the CPU tests - is used only to
gather per-rank counters after the timed region.
"""


def shard_streams(total_streams, world, rank):
    """Contiguous block of streams owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(int(total_streams), int(world))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def gather_counters(dist, device, elapsed_s, samples, kernel_ns, checksum, force=False):
    """Whole-job view of a timed region.

    elapsed is reduced with MAX (the job is as slow as its slowest rank); samples, kernel time
    and the PCM checksum are all-gathered.  Returns a dict (identical on every rank).  `dist` is
    the torch.distributed module or None for a single process.  force: go through the collectives even with one
    rank (bench.py --force-dist: the RCCL branch executed on the one GPU a development box has).
    """
    import torch
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return {"world": 1, "elapsed_s": float(elapsed_s), "samples": int(samples),
                "per_rank": [{"samples": int(samples), "kernel_ns": int(kernel_ns), "checksum": int(checksum)}]}
    world = dist.get_world_size()
    el = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    mine = torch.tensor([int(samples), int(kernel_ns), int(checksum)], dtype=torch.int64, device=device)
    allc = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine)
    per_rank = [{"samples": int(c[0]), "kernel_ns": int(c[1]), "checksum": int(c[2])} for c in allc]
    return {"world": world, "elapsed_s": float(el.item()), "samples": sum(r["samples"] for r in per_rank),
            "per_rank": per_rank, "backend": dist.get_backend()}
