"""Batches of independent images across the GPUs of one node (BASELINE config 3).

One process per GPU (`python -m torch.distributed.run --nproc-per-node N run_batch.py ...`): rank 0 broadcasts the run
configuration, every rank optimises its contiguous shard of the image list with no communication in between, and the
per-image result rows (loss, PSNR, iterations) are gathered on rank 0 at the end.  The reference processes images one
after another in a shell loop (run_inverse_pipeline.sh:16-28).
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from .dist import broadcast_config, gather_results, shard_range


def init_distributed(backend: str | None = None) -> tuple:
    """(rank, world, local_rank); initialises torch.distributed from the torchrun environment when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        from .dist import pin_rank_cores

        pin_rank_cores()                                               # the ranks of a node on disjoint core sets (dist.pin_rank_cores)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"     # "nccl" is RCCL on ROCm
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def run_batch(image_paths: Sequence[str], config: Dict, process_image: Callable[[int, str, Dict], Sequence[float]], device=None,
              process_shard: Optional[Callable[[List[int], List[str], Dict], List[Sequence[float]]]] = None) -> List[Dict]:
    """`process_image(image_id, path, config) -> [loss_mse, psnr, ...]` runs on the rank that owns the image; with
    `process_shard(ids, paths, config) -> one value list per image` the rank's whole contiguous shard is handed over at once
    (a batch in the kernels' batch dimension).  An exception while processing does not leave the other ranks hanging in the
    collectives: the affected images report NaN values and `"error"` carries the message.
    Returns on rank 0 one dict per image in list order ({"image_id", "path", "rank", "values", "error"}), [] elsewhere."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    cfg = broadcast_config({"paths": list(image_paths), "config": dict(config)} if rank == 0 else None)
    paths = cfg["paths"]
    lo, hi = shard_range(len(paths), world, rank)
    rows, errors = [], {}
    if process_shard is not None and hi > lo:
        ids = list(range(lo, hi))
        try:
            vals = process_shard(ids, [paths[i] for i in ids], cfg["config"])
            rows = [[float(i), float(rank), 0.0] + [float(v) for v in vv] for i, vv in zip(ids, vals)]
        except Exception as e:                                # noqa: BLE001 - reported, not swallowed
            errors = {i: f"{type(e).__name__}: {e}" for i in ids}
            rows = [[float(i), float(rank), 1.0] for i in ids]
    elif process_shard is None:
        for i in range(lo, hi):
            try:
                rows.append([float(i), float(rank), 0.0] + [float(v) for v in process_image(i, paths[i], cfg["config"])])
            except Exception as e:                            # noqa: BLE001
                errors[i] = f"{type(e).__name__}: {e}"
                rows.append([float(i), float(rank), 1.0])
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if (torch.cuda.is_available() and dist.is_initialized()
                                                                        and dist.get_backend() == "nccl") else torch.device("cpu")
    width = max([len(r) for r in rows], default=3)
    if dist.is_initialized():                     # ranks with an empty or failed shard still need the row width for the gather
        wt = torch.tensor([width], device=device)
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        width = int(wt.item())
    rows = [r + [float("nan")] * (width - len(r)) for r in rows]
    local = torch.tensor(rows, dtype=torch.float64, device=device).reshape(-1, width)
    parts = gather_results(local, dst=0)
    all_errors = [errors]
    if dist.is_initialized() and world > 1:
        all_errors = [None] * world
        dist.all_gather_object(all_errors, errors)
    if rank != 0:
        return []
    msgs = {k: v for e in all_errors for k, v in e.items()}
    out = []
    for part in parts:
        for row in part.cpu().tolist():
            i = int(row[0])
            out.append({"image_id": i, "path": paths[i], "rank": int(row[1]), "values": row[3:], "error": msgs.get(i) if row[2] else None})
    return out
