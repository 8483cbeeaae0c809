"""GPU side of the multi-GPU layout on ONE device: the world's shards are run one after the other through the same
sharding arithmetic and the library's pack / unpack kernels, and the gathered result must equal the unsharded call
("with < 8 devices run shards sequentially").  RCCL itself needs >= 2 GPUs; what a single GPU
can check is everything around the collective, and erh_allgather_topk at world size 1 (pack -> copy -> unpack)."""
import numpy as np
import pytest

from easyrag_amd import dist as erd
from easyrag_amd import synth
from easyrag_amd.engine import queries_to_csr
from easyrag_amd.index import BM25S, build_bm25_index

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workload(engine):
    import torch
    n, d, vocab, B = 30000, 256, 3000, 1003                        # 1003 queries: ragged over 8 ranks (126 / 125)
    x = synth.dense_corpus(n, d, seed=11)
    q = synth.dense_queries(x, B, seed=12).astype(np.float16)
    flat, lens = synth.token_corpus(n, vocab, seed=13, mean_len=24)
    docs = [list(map(int, t)) for t in synth.split_docs(flat, lens)]
    idx = build_bm25_index(docs, BM25S)
    queries = [idx.tokens_to_ids(list(map(int, t))) for t in synth.token_queries(flat, lens, vocab, B, seed=14)]
    engine.set_dense(x)
    engine.set_bm25(idx)
    engine.set_doc_meta(n, None, None)
    return torch.from_numpy(q).cuda(), queries


def _local(engine, q, queries, lo, hi, topk):
    qi, qt = queries_to_csr(queries[lo:hi])
    return engine.hybrid_topk(q[lo:hi], qi, qt, k_dense=288, k_sparse=192, K=60, topk=topk, device_out=True)


@pytest.mark.parametrize("world", [8, 3, 1])
def test_sequential_shards_equal_unsharded(engine, workload, world):
    import torch
    q, queries = workload
    B, topk = q.shape[0], 10
    full = _local(engine, q, queries, 0, B, topk)
    m = erd.max_shard(B, world)
    row = engine.topk_row_bytes(topk)
    recv = torch.zeros((world * m, row), dtype=torch.uint8, device=q.device)
    for r in range(world):                                            # what rank r would contribute to the all-gather
        lo, hi = erd.shard_bounds(B, r, world)
        ids, sc, ln = _local(engine, q, queries, lo, hi, topk)
        engine.pack_topk(ids, sc, ln, recv[r * m:(r + 1) * m])
    out = (torch.empty((B, topk), dtype=torch.int32, device=q.device),
           torch.empty((B, topk), dtype=torch.float64, device=q.device),
           torch.empty((B,), dtype=torch.int32, device=q.device))
    engine.unpack_topk(recv, B, world, topk, out)
    torch.cuda.synchronize()
    for a, b in zip(out, full):
        assert torch.equal(a, b)


def test_query_shards_world1_torch_and_native(engine, workload):
    """QueryShards at world size 1 in both GPU gather modes: "torch" (pack, copy, unpack) and "native"
    (erh_allgather_topk without a communicator)."""
    import torch
    q, queries = workload
    B = 200
    want = _local(engine, q, queries, 0, B, 10)
    for mode in ("torch", "native"):
        sh = erd.QueryShards(B, 0, 1, engine=engine, mode=mode)
        got = sh.step(lambda lo, hi: _local(engine, q, queries, lo, hi, 10))
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    # a one-rank RCCL communicator: init, gather through ncclAllGather's single-rank path, destroy
    uid = engine.comm_unique_id()
    assert len(uid) == 128
    engine.comm_init(0, 1, uid)
    try:
        got = engine.allgather_topk(*want, B)
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    finally:
        engine.comm_destroy()


def test_gather_completes_the_exhaustive_rounds_first(engine):
    """with device outputs only the first 16 uncertified queries of a batch are answered by the exhaustive path
    on the stream; the remaining rounds run inside erh_dense_check.  QueryShards.gather must therefore check before it
    packs: 40 queries whose speculative threshold fails (the sampling premise is broken on purpose, as in
    test_dense_speculation_failure_is_caught) go through a device-output call and the gather, and must equal the
    host-output call (which checks itself)."""
    import torch
    from oracle import to_f16_unit
    rng = np.random.default_rng(5)
    n, d, b, k = 40000, 256, 40, 60
    topic = rng.standard_normal(d)
    x32 = rng.standard_normal((n, d))
    x32[:40] = topic + 0.2 * rng.standard_normal((40, d))
    x = to_f16_unit(x32)
    q16 = to_f16_unit(topic + 0.2 * rng.standard_normal((b, d)))
    engine.set_option("dense_n0", 512)
    engine.set_option("dense_gemv", 0)
    engine.set_option("dense_shuffle", 0)
    try:
        engine.set_dense(x)
        want = engine.dense_topk(q16, k)                                  # host outputs: complete on return
        assert engine.dense_diag()["exhaustive"] == b
        qd = torch.from_numpy(q16).cuda()
        sh = erd.QueryShards(b, 0, 1, engine=engine, mode="torch")
        got = sh.step(lambda lo, hi: engine.dense_topk(qd[lo:hi], k, device_out=True))
        torch.cuda.synchronize()
        assert np.array_equal(got[0].cpu().numpy(), want[0])
        assert np.array_equal(got[1].cpu().numpy().view(np.uint64), want[1].view(np.uint64))
        assert np.array_equal(got[2].cpu().numpy(), want[2])
    finally:
        engine.set_option("dense_shuffle", 1)
        engine.set_option("dense_n0", 32768)
        engine.set_option("dense_gemv", 1)


def test_two_ranks_rccl_gather_modes():
    """Two RCCL ranks (skipped on a one-GPU box): tests/_rccl_rank_main.py under torch.distributed.run -- both gather modes
    ("torch": pack -> all_gather_into_tensor -> unpack; "native": erh_comm_init -> ncclAllGather inside the library) against
    the unsharded result on every rank.  north_star's split; the reference is single-process (src/main.py:48-52)."""
    import os
    import socket
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL does not share a device between ranks)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_rank_main.py")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), script], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL-GATHER-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_real_bench_two_processes_share_the_one_gpu(tmp_path):
    """VERDICT r4 (next 5): make the driver's first 8-GPU run not the first multi-process run.  The REAL bench.py with the REAL
    engine as two ranks under torch.distributed.run on the ONE GPU of this box (`--share-device --backend gloo`: RCCL does not
    share a device between ranks, so the packed rows cross through pinned host memory -- only the transport differs from the
    RCCL path; pack / unpack kernels, sharding, erh_dense_check before rows leave a rank, barriers, max-over-ranks timing and
    the rank-0 JSON line are the production code).  Exercises: two processes building / loading the library behind the file
    lock, two engines resident on one device, each rank generating the corpus and the 384-row tiled copy inside its warm-up.
    The global result of the last step must equal the single-process run over the whole batch."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "2", "--chunks", "65536", "--dim", "256", "--vocab", "8192", "--pool", "2",
              "--cpu-queries", "0", "--sub", "0"]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "512",
                          "--share-device", "--backend", "gloo", "--dump-out", str(tmp_path / "two")] + common,
                         env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    lines = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, two.stdout[-2000:]                                # rank 0 prints ONE JSON line, rank 1 none
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert rec["config"]["queries_per_gpu"] == 512 and rec["config"]["global_batch"] == 1024
    assert abs(rec["value"] - 1024 * 3 / (rec["ms_per_step"] * 3e-3)) < 1e-6 * rec["value"]
    mg = rec["multi_gpu"]
    assert mg["rccl_ranks"] == 2 and mg["backend"] == "gloo" and mg["transport"] == "gloo-staged" and mg["shared_device"] is True
    assert len(mg["setup_s_per_rank"]) == 2 and all(s > 0 for s in mg["setup_s_per_rank"])
    assert rec["roofline"] is not None and rec["cpu_baseline"] is None
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--batch", "1024",
                          "--dump-out", str(tmp_path / "one")] + common, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    want = np.load(tmp_path / "one.rank0.npz")
    for r in range(2):                                                        # every rank holds the GLOBAL result
        got = np.load(tmp_path / f"two.rank{r}.npz")
        assert got["ids"].shape == (1024, 10)
        assert np.array_equal(got["ids"], want["ids"]) and np.array_equal(got["len"], want["len"])
        assert np.array_equal(got["scores"].view(np.uint64), want["scores"].view(np.uint64))


def test_matrix_copies_are_allocated_on_first_use():
    """Eight ranks of one node each hold a replica of the corpus; the two optional copies of the chunk matrix (the 384-row tiled copy of
    batches padded to >= 512 queries, the per-dir block copies of filtered batches) must not exist before the first call that uses
    them: erh_set_dense leaves one matrix on the device, a 256-query call adds only work space, the first 1024-query call adds the
    tiled copy, the first filtered call the blocks.  (A handle of its own: the session's engine keeps the buffers of earlier tests.)"""
    import torch
    from easyrag_amd import synth
    from easyrag_amd.engine import RetrievalEngine
    dev = torch.device("cuda", 0)
    n, d = 400_000, 512
    mat = n * d * 2
    x = synth.dense_corpus_torch(n, d, seed=5, device=dev)
    q = synth.dense_queries_torch(x, 1024, seed=6)
    torch.cuda.synchronize()

    def used():
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        return total - free

    eng = RetrievalEngine(0)
    try:
        u0 = used()
        eng.set_dense(x)
        eng.set_doc_meta(n, None, (np.arange(n) * 4 // n).astype(np.int16))
        u1 = used()
        assert 0.95 * mat < u1 - u0 < 1.3 * mat, (u1 - u0, mat)                     # one copy (+ padding rows, metadata)
        eng.dense_topk(q[:256].contiguous(), 100)
        u2 = used()
        assert u2 - u1 < 0.6 * mat, (u2 - u1, mat)                                  # work space (seed scores, candidate lists), no matrix copy
        eng.dense_topk(q, 288)
        u3 = used()
        assert u3 - u2 > 0.9 * mat                                                  # the 384-row tiled copy, on first use
        eng.dense_topk(q, 288)
        assert used() - u3 < 0.05 * mat                                             # ... once
        filt = (np.arange(1024) % 4).astype(np.int16)
        eng.reset_stats()
        eng.dense_topk(q, 288, filter_dir=filt)
        u4 = used()
        assert eng.stat("dense_block_groups") == 4 and u4 - u3 > 0.9 * mat          # the dir blocks, on the first filtered call
    finally:
        eng.close()
    del x
