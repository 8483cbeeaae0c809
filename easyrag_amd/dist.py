"""Multi-GPU layout of the hot path: corpus replicated per GPU, query batch sharded contiguously across
ranks, one all-gather of the fused top-k (BASELINE.json north_star).

Queries are independent units, so there is no data-path collective inside the retrieval itself; the only
exchange is the final gather of [B_local x k] (score, id, len) rows -- ~120 KB per rank at 1024 x 10, i.e. pure
latency on xGMI.  One process per GPU.  Three ways to run the gather, same result:

  "torch"   (default on GPUs) the library packs the rows (erh_pack_topk), `torch.distributed`
            all_gather_into_tensor (backend "nccl" = RCCL) moves ONE preallocated uint8 buffer, the library
            unpacks into preallocated global arrays (erh_unpack_topk): two tiny kernels + one collective, no
            per-step allocation, no torch.cat.
  "native"  erh_allgather_topk: pack + ncclAllGather + unpack inside the library on the caller's stream (its own
            RCCL communicator, bootstrapped by broadcasting erh_comm_unique_id through torch.distributed).
  "host"    CPU tensors / no engine: padded blocks through all_gather_into_tensor ("gloo" in the CPU tests).

The reference has no counterpart (single process, one query at a time: src/main.py:48-52).
"""
from __future__ import annotations

import os
import sys
from typing import Callable, Optional, Tuple

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


def init_from_env(device_index: Optional[int] = None, backend: Optional[str] = None) -> Tuple[int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun sets them).
    Returns (rank, world).  Single-process runs skip the group entirely.  `backend` (or ERH_DIST_BACKEND): "nccl" (= RCCL, the
    default on GPUs) or "gloo" -- the dry-run transport for several ranks on ONE GPU (RCCL does not share a device between
    ranks): QueryShards then stages the packed rows through the host, everything else is the same code."""
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
        backend = backend or os.environ.get("ERH_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def allgather_topk(ids: torch.Tensor, scores: torch.Tensor, lens: torch.Tensor, n_queries: int,
                   group=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """"host" gather: every rank's [B_local x k] fused top-k -> the global [n_queries x k] result, in query order.
    Shards are padded to max_shard rows so that all ranks contribute equal-sized blocks; the padding rows are
    dropped afterwards.  (GPU runs use QueryShards below, which packs in the library and allocates nothing.)"""
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


class QueryShards:
    """One rank's view of a sharded global query batch: its contiguous shard and the gather of the results.

        sh = QueryShards(n_queries, rank, world, engine=eng)          # engine=None -> "host" gather (CPU tests)
        lo, hi = sh.bounds
        ids, sc, ln = sh.step(lambda lo, hi: eng.hybrid_topk(q[lo:hi], ..., device_out=True))

    `step` runs the local retrieval on [lo, hi) and returns the GLOBAL result on every rank.  bench.py --gpus N and
    the tests drive exactly this code."""

    def __init__(self, n_queries: int, rank: int, world: int, engine=None, mode: Optional[str] = None, group=None):
        self.n, self.rank, self.world, self.engine, self.group = int(n_queries), int(rank), int(world), engine, group
        self.bounds = shard_bounds(self.n, self.rank, self.world)
        self.m = max_shard(self.n, self.world)
        if mode is None:
            mode = os.environ.get("ERH_GATHER", "torch") if engine is not None else "host"
        if mode not in ("torch", "native", "host"):
            raise ValueError("gather mode must be torch, native or host")
        if mode != "host" and engine is None:
            raise ValueError("the packed gathers need the engine (the library packs and unpacks)")
        self.mode = mode
        self._k = None
        self._send = self._recv = self._out = None
        self._send_host = self._recv_host = None
        self.fallback_reason = None
        self.transport = None                     # set by the first gather: "rccl", "gloo", "gloo-staged" (device rows through pinned host memory) or "local"
        if mode == "native" and world > 1:
            self._init_native()

    def _init_native(self):
        """Join the library's own RCCL communicator -- or, if any rank cannot (librccl not loadable, ncclCommInitRank
        failed or timed out), fall back to the "torch" gather on EVERY rank (the decision is an all-reduce, so no rank
        is left waiting inside a collective the others never enter)."""
        eng, rank, world, group = self.engine, self.rank, self.world, self.group
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

        def all_ok(ok: bool) -> bool:
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(int(t.item()))

        reason = None
        uid = None
        try:                                              # every rank probes its own librccl (cheap) before anyone blocks
            uid = eng.comm_unique_id()
        except Exception as exc:                          # noqa: BLE001 -- any failure means "not available here"
            reason = f"librccl not usable on rank {rank}: {exc}"
        if all_ok(reason is None):
            box = [uid if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            try:
                eng.comm_init(rank, world, box[0])
            except Exception as exc:                      # noqa: BLE001
                reason = f"erh_comm_init failed on rank {rank}: {exc}"
            if not all_ok(reason is None):
                reason = reason or "erh_comm_init failed on another rank"
                try:
                    eng.comm_destroy()                    # (a rank whose own init timed out: the library reaps the late communicator)
                except Exception:                         # noqa: BLE001
                    pass
        else:
            reason = reason or "librccl not usable on another rank"
        if reason is not None:
            self.mode, self.fallback_reason = "torch", reason
            print(f"[easyrag_amd.dist] rank {rank}: native gather unavailable, using torch.distributed ({reason})",
                  file=sys.stderr, flush=True)

    def _buffers(self, k: int, device):
        if self._k != k:
            row = self.engine.topk_row_bytes(k)
            self._send = torch.empty((self.m, row), dtype=torch.uint8, device=device)
            self._recv = torch.empty((self.m * self.world, row), dtype=torch.uint8, device=device)
            self._out = (torch.empty((self.n, k), dtype=torch.int32, device=device),
                         torch.empty((self.n, k), dtype=torch.float64, device=device),
                         torch.empty((self.n,), dtype=torch.int32, device=device))
            self._k = k
        return self._send, self._recv, self._out

    def gather(self, ids, sc, ln, check: bool = True):
        """All-gather this rank's [B_local x k] result.  `check` (default): erh_dense_check first.  Since round 5 a dense / fused call
        with device outputs enqueues only the COUNT of the queries its candidate budgets could not certify; EVERY answering round of
        the exhaustive path (and the re-run of the fusion over the corrected lists, and of a routed call's flagged groups) runs inside
        erh_dense_check, so rows must not leave the rank before it has run.  check=False is for callers that have called
        `engine.dense_check()` themselves since the last dense / fused call -- skipping it altogether hands out uncertified lists."""
        lo, hi = self.bounds
        if ids.shape[0] != hi - lo:
            raise ValueError(f"rank {self.rank} must contribute its shard of {hi - lo} queries, got {ids.shape[0]}")
        if self.mode == "host":
            return allgather_topk(ids, sc, ln, self.n, self.group)
        if check:
            self.engine.dense_check()                      # (returns at once when the last call had no dense route)
        k = int(ids.shape[1])
        send, recv, out = self._buffers(k, ids.device)
        if self.mode == "native":
            return self.engine.allgather_topk(ids, sc, ln, self.n, out=out)
        self.engine.pack_topk(ids, sc, ln, send)
        if self.world > 1:
            staged = send.is_cuda and dist.get_backend(self.group) == "gloo"
            if staged:
                # several ranks on ONE GPU (dry run of the multi-process path: RCCL does not share a device between ranks):
                # the packed rows cross the process boundary through pinned host memory; pack and unpack stay on the device
                if self._send_host is None or self._send_host.shape != send.shape:
                    self._send_host = torch.empty(send.shape, dtype=send.dtype, pin_memory=True)
                    self._recv_host = torch.empty(recv.shape, dtype=recv.dtype, pin_memory=True)
                self._send_host.copy_(send)                # (synchronises: the pack kernel has finished)
                dist.all_gather_into_tensor(self._recv_host, self._send_host, group=self.group)
                recv.copy_(self._recv_host)
                self.transport = "gloo-staged"
            else:
                dist.all_gather_into_tensor(recv, send, group=self.group)
                self.transport = "rccl" if send.is_cuda else "gloo"
        else:
            recv.copy_(send)
            self.transport = "local"
        self.engine.unpack_topk(recv, self.n, self.world, k, out)
        return out

    def step(self, run_local: Callable[[int, int], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]):
        lo, hi = self.bounds
        return self.gather(*run_local(lo, hi))
