"""Multi-GPU layout of the hot path: corpus replicated per GPU, query batch sharded contiguously across
ranks, one all-gather of the fused top-k (BASELINE.json north_star).

Queries are independent units, so there is no data-path collective inside the retrieval itself; the only
exchange is the final gather of [B_local x k] (id, score) blocks -- ~120 KB per rank at 1024 x 10, i.e. pure
latency on xGMI.  One process per GPU, `torch.distributed` backend "nccl" (= RCCL on ROCm); the same code
runs on CPU tensors with "gloo" (tests/test_dist_gloo.py).  The reference has no counterpart (single
process, one query at a time: src/main.py:48-52).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_queries: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`; shards differ by at most one query."""
    base, rem = divmod(n_queries, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def max_shard(n_queries: int, world: int) -> int:
    return (n_queries + world - 1) // world


def init_from_env(device_index: Optional[int] = None) -> Tuple[int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun sets them).
    Returns (rank, world).  Single-process runs skip the group entirely."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        use_gpu = torch.cuda.is_available()
        if use_gpu:
            if device_index is None:
                device_index = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(device_index)
        dist.init_process_group(backend="nccl" if use_gpu else "gloo", rank=rank, world_size=world)
    return rank, world


def allgather_topk(ids: torch.Tensor, scores: torch.Tensor, lens: torch.Tensor, n_queries: int,
                   group=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Gather every rank's [B_local x k] fused top-k into the global [n_queries x k] result, in query order.

    Shards are padded to max_shard rows so that all ranks contribute equal-sized blocks (one
    all_gather_into_tensor per array); the padding rows are dropped afterwards."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ids[:n_queries], scores[:n_queries], lens[:n_queries]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    m = max_shard(n_queries, world)
    lo, hi = shard_bounds(n_queries, rank, world)
    k = ids.shape[1]

    def pad(t, fill):
        if t.shape[0] == m:
            return t.contiguous()
        out = torch.full((m,) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=t.device)
        out[: hi - lo] = t[: hi - lo]
        return out

    p_ids, p_sc, p_ln = pad(ids, -1), pad(scores, 0), pad(lens, 0)
    # one collective instead of three (the exchange is pure latency): rows packed as [scores | ids | len] bytes,
    # widest type first so that every field stays aligned inside a row
    nb_sc, nb_id, nb_ln = k * p_sc.element_size(), k * p_ids.element_size(), p_ln.element_size()
    packed = torch.cat([p_sc.view(torch.uint8).reshape(m, nb_sc), p_ids.view(torch.uint8).reshape(m, nb_id),
                        p_ln.reshape(m, 1).view(torch.uint8).reshape(m, nb_ln)], dim=1).contiguous()
    g = torch.empty((world * m, packed.shape[1]), dtype=torch.uint8, device=ids.device)
    dist.all_gather_into_tensor(g, packed, group=group)
    g_sc = g[:, :nb_sc].contiguous().view(scores.dtype).reshape(world * m, k)
    g_ids = g[:, nb_sc:nb_sc + nb_id].contiguous().view(ids.dtype).reshape(world * m, k)
    g_ln = g[:, nb_sc + nb_id:].contiguous().view(lens.dtype).reshape(world * m)
    if n_queries == world * m:
        return g_ids, g_sc, g_ln
    keep = torch.cat([torch.arange(r * m, r * m + (shard_bounds(n_queries, r, world)[1] - shard_bounds(n_queries, r, world)[0]),
                                   device=ids.device) for r in range(world)])
    return g_ids[keep], g_sc[keep], g_ln[keep]
