"""N > 1 path on CPU: two gloo ranks shard a query batch, each fuses its shard, one all-gather rebuilds the
global result.  The per-rank "retrieval" here is the oracle (tests may use it); what is under test is the
sharding arithmetic and the gather in easyrag_amd.dist, which bench.py uses unchanged on RCCL."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from easyrag_amd import dist as erd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fused_oracle(n_q, k):
    """Deterministic per-query 'fused top-k' standing in for the GPU result."""
    rng = np.random.default_rng(123)
    ids = rng.integers(0, 10_000, size=(n_q, k)).astype(np.int32)
    sc = np.sort(rng.random((n_q, k)), axis=1)[:, ::-1].copy()
    ln = rng.integers(0, k + 1, size=n_q).astype(np.int32)
    return ids, sc, ln


def _worker(rank, world, port, n_q, k, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w = erd.init_from_env()
    assert (r, w) == (rank, world)
    ids, sc, ln = _fused_oracle(n_q, k)
    lo, hi = erd.shard_bounds(n_q, rank, world)
    g = erd.allgather_topk(torch.from_numpy(ids[lo:hi]), torch.from_numpy(sc[lo:hi]), torch.from_numpy(ln[lo:hi]), n_q)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), ids=g[0].numpy(), sc=g[1].numpy(), ln=g[2].numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


class _FakeEngine:
    """What bench.step() hands to QueryShards.step on a GPU rank, minus the GPU: a local retrieval over [lo, hi)."""

    def __init__(self, n_q, k):
        self.ids, self.sc, self.ln = _fused_oracle(n_q, k)

    def local(self, lo, hi):
        return torch.from_numpy(self.ids[lo:hi]), torch.from_numpy(self.sc[lo:hi]), torch.from_numpy(self.ln[lo:hi])


def _worker_shards(rank, world, port, n_q, k, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    erd.init_from_env()
    fake = _FakeEngine(n_q, k)
    sh = erd.QueryShards(n_q, rank, world, engine=None)              # "host" gather: same step() code path as bench.py
    assert sh.mode == "host" and sh.bounds == erd.shard_bounds(n_q, rank, world)
    for _ in range(2):                                               # a second step reuses the object
        g = sh.step(fake.local)
    np.savez(os.path.join(out_dir, f"s{rank}.npz"), ids=g[0].numpy(), sc=g[1].numpy(), ln=g[2].numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_query_shards_step(tmp_path):
    """The code path of bench.py --gpus N (QueryShards.step) on two gloo ranks with a fake engine."""
    for n_q in (32, 29):
        port = _free_port()
        mp.spawn(_worker_shards, args=(2, port, n_q, 10, str(tmp_path)), nprocs=2, join=True)
        ids, sc, ln = _fused_oracle(n_q, 10)
        for r in range(2):
            z = np.load(tmp_path / f"s{r}.npz")
            assert np.array_equal(z["ids"], ids) and np.array_equal(z["sc"], sc) and np.array_equal(z["ln"], ln)


def test_query_shards_rejects_wrong_shard():
    sh = erd.QueryShards(10, 0, 1, engine=None)
    import pytest
    with pytest.raises(ValueError):
        sh.gather(torch.zeros((3, 4), dtype=torch.int32), torch.zeros((3, 4), dtype=torch.float64),
                  torch.zeros((3,), dtype=torch.int32))


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 1024, 8191):
        for w in (1, 2, 3, 8):
            b = [erd.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
            assert max(h - l for l, h in b) <= erd.max_shard(n, w) or n == 0


def test_two_rank_allgather_rebuilds_global_result(tmp_path):
    for n_q in (16, 13):                      # even and ragged shards
        port = _free_port()
        mp.spawn(_worker, args=(2, port, n_q, 10, str(tmp_path)), nprocs=2, join=True)
        ids, sc, ln = _fused_oracle(n_q, 10)
        for r in range(2):
            z = np.load(tmp_path / f"r{r}.npz")
            assert np.array_equal(z["ids"], ids) and np.array_equal(z["sc"], sc) and np.array_equal(z["ln"], ln)
