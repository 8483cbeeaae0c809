"""bench.main()'s world > 1 control flow on two gloo ranks, no GPU: barrier, timed steps, per-step gather (pack -> ONE
all_gather_into_tensor -> unpack through QueryShards in "torch" mode), all_reduce(MAX) of the step time, rank 0's JSON line
with the `multi_gpu` block -- the code the driver's first 8-GPU run executes, with the engine replaced by a stand-in that
answers every query with a deterministic function of its tokens.  What is under test is bench.py and easyrag_amd.dist, not
retrieval.  (north_star: corpus replicated, query batch sharded, one all-gather of the fused top-k; the reference is single
process, /root/reference/src/main.py:48-52.)"""
import json
import os
import socket
import sys
import time

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ARGV = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--chunks", "4096", "--dim", "64", "--vocab", "512", "--batch", "9",
        "--pool", "2", "--cpu-queries", "0", "--sub", "0"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _answer(tokens, topk, n_docs):
    """The stand-in's 'fused top-k' of one query: a function of its token ids only."""
    base = int(np.sum(np.asarray(tokens, np.int64) * 31 + 7))
    ids = np.asarray([(base + 101 * j) % n_docs for j in range(topk)], np.int32)
    sc = np.asarray([1.0 / (j + 1 + (base % 13)) for j in range(topk)], np.float64)
    return ids, sc, np.int32(topk - (base % 3))


class StubEngine:
    """The RetrievalEngine surface bench.main() and QueryShards("torch") touch."""

    def __init__(self, local):
        self.local, self.calls, self.n_docs = local, [], 0

    def set_option(self, name, value): self.calls.append(("opt", name, value))
    def set_dense(self, x): self.n_docs = int(x.shape[0])
    def set_bm25(self, idx, payload_on_device=False, slot=None): self.n_docs = int(idx.n_docs)
    def set_doc_meta(self, n, cid, did): pass
    def set_profiling(self, on): self.calls.append(("prof", bool(on)))
    def reset_kernel_time(self): pass
    def kernel_time(self, cls): return {"ms": 0.0, "launches": 0, "bytes": 0.0, "flops": 0.0}
    def dense_check(self): self.calls.append(("dense_check",))
    def close(self): self.calls.append(("close",))

    def hybrid_topk(self, q16, qi, qt, k_dense, k_sparse, K, topk, device_out=True):
        B = len(qi) - 1
        assert q16.shape[0] == B and device_out
        rows = [_answer(qt[qi[b]:qi[b + 1]], topk, self.n_docs) for b in range(B)]
        return (torch.from_numpy(np.stack([r[0] for r in rows])), torch.from_numpy(np.stack([r[1] for r in rows])),
                torch.from_numpy(np.asarray([r[2] for r in rows], np.int32)))

    # the packed gather: rows of [k doubles | k int32 | int32 len], as the library's erh_pack_topk / erh_unpack_topk lay them out
    def topk_row_bytes(self, k): return 12 * k + 4

    def pack_topk(self, ids, sc, ln, send):
        b, k = ids.shape
        send.zero_()
        send[:b, :8 * k] = sc.contiguous().view(torch.uint8).reshape(b, 8 * k)
        send[:b, 8 * k:12 * k] = ids.contiguous().view(torch.uint8).reshape(b, 4 * k)
        send[:b, 12 * k:] = ln.contiguous().reshape(b, 1).view(torch.uint8).reshape(b, 4)

    def unpack_topk(self, recv, n, world, k, out):
        from easyrag_amd.dist import max_shard, shard_bounds
        m = max_shard(n, world)
        for r in range(world):
            lo, hi = shard_bounds(n, r, world)
            blk = recv[r * m:r * m + (hi - lo)]
            out[1][lo:hi] = blk[:, :8 * k].contiguous().view(torch.float64).reshape(hi - lo, k)
            out[0][lo:hi] = blk[:, 8 * k:12 * k].contiguous().view(torch.int32).reshape(hi - lo, k)
            out[2][lo:hi] = blk[:, 12 * k:].contiguous().view(torch.int32).reshape(hi - lo)


class _Event:
    def record(self): self.t = time.perf_counter()
    def elapsed_time(self, other): return (other.t - self.t) * 1e3


class StubPlatform:
    def __init__(self):
        from easyrag_amd import synth
        from easyrag_amd.index import build_bm25_index_from_postings
        self.synth, self.build_index, self.engines = synth, build_bm25_index_from_postings, []

    @staticmethod
    def queries_to_csr(queries):
        ptr = np.zeros(len(queries) + 1, np.int32)
        ptr[1:] = np.cumsum([len(q) for q in queries])
        return ptr, (np.concatenate([np.asarray(q, np.int32) for q in queries]) if len(queries) else np.zeros(0, np.int32))

    def set_device(self, local): return torch.device("cpu")
    def synchronize(self): pass
    def event(self): return _Event()

    def make_engine(self, local):
        self.engines.append(StubEngine(local))
        return self.engines[-1]


def _rank_main(rank, world, port, out_dir, extra=()):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    plat = StubPlatform()
    sys.stdout = open(os.path.join(out_dir, f"stdout{rank}.txt"), "w")
    res = bench.main(ARGV + list(extra), platform=plat)
    sys.stdout.flush()
    ids, sc, ln = res["out"]
    np.savez(os.path.join(out_dir, f"out{rank}.npz"), ids=ids.numpy(), sc=sc.numpy(), ln=ln.numpy())
    json.dump({"record": res["record"], "calls": [list(c) for c in plat.engines[0].calls]}, open(os.path.join(out_dir, f"res{rank}.json"), "w"))


def test_bench_main_two_gloo_ranks(tmp_path):
    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    # rank 0 printed exactly one JSON line; rank 1 printed nothing
    line0 = open(tmp_path / "stdout0.txt").read().strip().splitlines()
    assert len(line0) == 1 and open(tmp_path / "stdout1.txt").read().strip() == ""
    rec = json.loads(line0[0])
    assert rec == json.load(open(tmp_path / "res0.json"))["record"]
    assert json.load(open(tmp_path / "res1.json"))["record"] is None
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["queries_per_gpu"] == 9 and rec["config"]["global_batch"] == 18
    assert abs(rec["value"] - 18 * 3 / (rec["ms_per_step"] * 3e-3)) < 1e-6 * rec["value"]       # whole-job aggregate over both ranks
    mg = rec["multi_gpu"]
    assert mg["rccl_ranks"] == 2 and mg["backend"] == "gloo" and mg["gather_mode"] == "torch" and mg["gather_fallback_reason"] is None
    assert mg["allgather_ms_per_step"] >= 0.0
    # the token corpus was generated once (rank 0) and read by rank 1 from /dev/shm; the files are gone afterwards
    assert mg["corpus_shared"] is True and len(mg["setup_s_per_rank"]) == 2
    import glob
    assert not glob.glob("/dev/shm/erh_bench_n4096_v512_*")
    assert rec["cpu_baseline"] is None                                                          # rank 0 at N = 1 only
    # every rank holds the GLOBAL result of the last step, equal to the unsharded answer
    from easyrag_amd import synth
    n, vocab, n_global, pool = 4096, 512, 18, 2
    indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=torch.device("cpu"))
    p_last = (1 + 3 - 1) % pool
    queries = synth.token_queries(flat, lens, vocab, n_global, seed=2000 + p_last)
    want = [_answer(q, 10, n) for q in queries]
    for r in range(2):
        z = np.load(tmp_path / f"out{r}.npz")
        assert z["ids"].shape == (n_global, 10)
        assert np.array_equal(z["ids"], np.stack([w[0] for w in want]))
        assert np.array_equal(z["sc"], np.stack([w[1] for w in want]))
        assert np.array_equal(z["ln"], np.asarray([w[2] for w in want], np.int32))
        calls = json.load(open(tmp_path / f"res{r}.json"))["calls"]
        # per step: the gather checks the dense route's flags before rows leave the rank (4 steps) + the two checks around the timed region
        assert calls.count(["dense_check"]) == 4 + 2 and calls[-1] == ["close"]


def test_bench_native_gather_falls_back_on_every_rank(tmp_path):
    """--gather native on ranks whose engine cannot join an RCCL communicator (here: the stand-in has no erh_comm_* at all; on a
    box: librccl missing, ncclCommInitRank failing or timing out): the decision is an all-reduce, EVERY rank falls back to the
    torch.distributed gather, the record says so and carries the reason, and the results are the unsharded answer."""
    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path), ("--gather", "native")), nprocs=2, join=True)
    rec = json.loads(open(tmp_path / "stdout0.txt").read().strip().splitlines()[0])
    mg = rec["multi_gpu"]
    assert mg["gather_mode"] == "torch" and mg["gather_fallback_reason"] and "rank" in mg["gather_fallback_reason"]
    from easyrag_amd import synth
    n, vocab, n_global, pool = 4096, 512, 18, 2
    indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=torch.device("cpu"))
    queries = synth.token_queries(flat, lens, vocab, n_global, seed=2000 + (1 + 3 - 1) % pool)
    want = [_answer(q, 10, n) for q in queries]
    for r in range(2):
        z = np.load(tmp_path / f"out{r}.npz")
        assert np.array_equal(z["ids"], np.stack([w[0] for w in want])) and np.array_equal(z["sc"], np.stack([w[1] for w in want]))
