"""Sharding of independent images across ranks: contiguous shards, one process per
GPU, no collective inside an optimisation iteration; one broadcast of the run configuration at start and one
gather of per-image results at the end (RCCL when the backend is "nccl", gloo on CPU for tests)."""
from __future__ import annotations

from typing import Optional, Any, List, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of `n_items` for `rank`; the first n_items % world_size ranks get one extra item."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_config(cfg: Any, src: int = 0) -> Any:
    """Rank `src`'s run configuration (any picklable object) to every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return cfg
    box = [cfg if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def broadcast_tensor(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def gather_results(local: torch.Tensor, dst: int = 0) -> List[torch.Tensor]:
    """Per-image result rows of every rank collected on `dst` (ranks may hold different numbers of rows).
    Returns the list ordered by rank on `dst`, [] elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local]
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    cap = int(max(int(c.item()) for c in counts))
    padded = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    bufs = [torch.zeros_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst)
    if rank != dst:
        return []
    return [b[: int(c.item())] for b, c in zip(bufs, counts)]


def broadcast_state_dict(state_dict, device, src: int = 0):
    """Rank `src`'s network weights to every rank as ONE flat fp32 buffer (MaterialNet's 108 M parameters = 433 MB leave
    rank 0 once, over all of its xGMI links; the other ranks never touch the weights file).  `state_dict` is read on `src` only; returns
    the same mapping of name -> tensor (views of the received buffer, on `device`) on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return {k: v.to(device) for k, v in state_dict.items()}
    rank = dist.get_rank()
    layout = broadcast_config([(k, tuple(v.shape), str(v.dtype)) for k, v in state_dict.items()] if rank == src else None, src=src)
    # one flat buffer PER DTYPE, each in its own type: integer buffers (BatchNorm's num_batches_tracked, index tables) and float64 entries
    # arrive exactly -- a single fp32 buffer is exact for integers only up to 2^24 and rounds doubles
    out = {}
    for dtype in sorted({d for _, _, d in layout}):
        td = getattr(torch, dtype.split(".")[-1])
        group = [(k, shape) for k, shape, d in layout if d == dtype]
        total = sum(int(torch.Size(shape).numel()) for _, shape in group)
        wire = torch.uint8 if td == torch.bool else td                     # collectives do not take bool
        flat = torch.empty(total, dtype=wire, device=device)
        if rank == src:
            off = 0
            for k, shape in group:
                n = int(torch.Size(shape).numel())
                flat[off:off + n] = state_dict[k].detach().reshape(-1).to(device, wire)
                off += n
        dist.broadcast(flat, src=src)
        off = 0
        for k, shape in group:
            n = int(torch.Size(shape).numel())
            out[k] = flat[off:off + n].view(shape).to(td)
            off += n
    return {k: out[k] for k, _, _ in layout}


_PINNED: Optional[dict] = None          # what pin_rank_cores did in this process (it acts once)


def pin_rank_cores() -> Optional[dict]:
    """One process per GPU: the ranks of a node take DISJOINT sets of host cores (LOCAL_RANK-th slice of the cores this process may
    run on).  A rank's iterations are enqueued by one Python thread -- ~30 launches (pos_mlp) or 6 (none mode) per iteration -- and where that
    takes about as long as the GPU needs for them (bench.py `host_enqueue`: the 8-image none-mode shard), two ranks sharing a core is what an
    8-GPU run loses on, not the fabric.  Returns what was done (None: nothing -- a single local rank, no affinity API, MATPBR_NO_PIN set,
    LOCAL_WORLD_SIZE unknown, or a launcher that has bound this rank already).

    Acts ONCE per process (a second call returns the first call's record: slicing the slice again would leave a rank on 1 / local_world of
    its cores).  A process whose mask is already a strict subset of the node's cores was bound by its launcher (slurm --cpu-bind, numactl per
    rank): its mask is left alone.  torch's intra-op pool is sized to the slice (`torch.set_num_threads`: OMP_NUM_THREADS set here would come
    after torch and OpenMP were loaded)."""
    import os

    global _PINNED
    if _PINNED is not None:
        return _PINNED
    if os.environ.get("MATPBR_NO_PIN") or not hasattr(os, "sched_setaffinity") or "LOCAL_WORLD_SIZE" not in os.environ:
        return None          # LOCAL_WORLD_SIZE is required (torch.distributed.run sets it): WORLD_SIZE over-counts the ranks of a node on multi-node jobs
    local_world, local_rank = int(os.environ["LOCAL_WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    if local_world <= 1:
        return None
    cores = sorted(os.sched_getaffinity(0))
    node = os.cpu_count() or len(cores)
    if len(cores) < node and len(cores) * local_world <= node:
        # the launcher has given this rank a share of the node already: leave it (and size the thread pool to it)
        torch.set_num_threads(max(1, min(len(cores), torch.get_num_threads())))
        _PINNED = {"cores_of_this_rank": len(cores), "first": cores[0], "last": cores[-1], "local_world_size": local_world, "bound_by": "launcher"}
        return _PINNED
    per = max(1, len(cores) // local_world)
    mine = cores[local_rank * per:(local_rank + 1) * per] or cores
    os.sched_setaffinity(0, mine)
    torch.set_num_threads(max(1, min(len(mine), int(os.environ.get("OMP_NUM_THREADS", "4")))))
    _PINNED = {"cores_of_this_rank": len(mine), "first": mine[0], "last": mine[-1], "local_world_size": local_world, "bound_by": "pin_rank_cores"}
    return _PINNED
