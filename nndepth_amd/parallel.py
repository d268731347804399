"""Batch-parallel inference over the GPUs of one node.

Every stereo pair is independent at inference, so the path shards by batch with NO collective
inside forward(); the only exchange is one all-gather of the final disparity maps
(`backend="nccl"` is RCCL over xGMI on ROCm; the CPU tests use gloo).  One process per GPU,
launched by `python -m torch.distributed.run`.
"""
import os
from typing import Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_distributed(backend: str = "nccl") -> Tuple[int, int, int]:
    rank, world, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous slice of `n_items` owned by `rank` (first n%world ranks get one extra)."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def gather_disparity(local: torch.Tensor) -> torch.Tensor:
    """All-gather equally-shaped per-rank disparity batches (b,1,H,W) -> (world*b,1,H,W), rank order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def gather_ragged(local: torch.Tensor, n_items: int) -> torch.Tensor:
    """All-gather of per-rank batches whose sizes follow shard_range(n_items, rank, world) (they differ by at most one):
    every rank pads its slice to the largest shard, ONE all_gather_into_tensor, then the padding rows are dropped.
    -> (n_items, ...) in item order, on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        assert local.shape[0] == n_items
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [len(shard_range(n_items, r, world)) for r in range(world)]
    assert local.shape[0] == counts[rank], f"rank {rank} holds {local.shape[0]} items, its shard has {counts[rank]}"
    cap = max(counts)
    if local.shape[0] < cap:
        pad = torch.zeros((cap - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    out = torch.empty((world * cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    if all(c == cap for c in counts):
        return out
    return torch.cat([out[r * cap:r * cap + counts[r]] for r in range(world)], 0)


def sharded_inference(n_pairs: int, load_pairs, forward_fn, micro_batch: int, rank: int = None, world: int = None,
                      gather: bool = True) -> torch.Tensor:
    """The batch-parallel job of BASELINE.json configs[3] / [4] (north_star: "full-image batches shard over the 8 GPUs of
    one node with RCCL all-gather of disparity only for the batch-parallel case"):

        pairs shard_range(n_pairs, rank, world)  ->  micro-batches of `micro_batch` pairs
        load_pairs(list of pair ids) -> (frame1, frame2) resident on this rank's GPU
        forward_fn(frame1, frame2)   -> final disparity (b, C, H, W) of the micro-batch      [no collective inside]
        one all-gather of the rank's disparities at the end -> (n_pairs, C, H, W) in pair order on every rank.

    With gather=False the rank-local result is returned (what a caller that only needs its own pairs would use)."""
    r, w, _ = env_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    # checked on every rank BEFORE any work or collective, so that all ranks raise together instead of the others
    # hanging in the all-gather
    if not 0 < world <= n_pairs or not 0 <= rank < world:
        raise ValueError(f"sharded_inference: rank {rank} / world {world} cannot shard {n_pairs} pairs (need 1 <= world <= n_pairs)")
    mine = list(shard_range(n_pairs, rank, world))
    outs = []
    for i in range(0, len(mine), micro_batch):
        ids = mine[i:i + micro_batch]
        f1, f2 = load_pairs(ids)
        outs.append(forward_fn(f1, f2))
    local = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
    return gather_ragged(local, n_pairs) if gather else local


def max_over_ranks(seconds: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
