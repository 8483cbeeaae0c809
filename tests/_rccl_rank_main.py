"""One rank of tests/test_gpu_dist.py::test_two_ranks_rccl_gather_modes (run under torch.distributed.run, one process per
GPU): a small corpus replicated on every rank, the global query batch sharded with QueryShards, the fused top-k gathered in
BOTH gather modes -- "torch" (library pack -> all_gather_into_tensor on RCCL -> library unpack) and "native"
(erh_comm_init + ncclAllGather inside the library) -- and compared on every rank with the unsharded call."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from easyrag_amd import dist as erd, synth                      # noqa: E402
from easyrag_amd.engine import RetrievalEngine, queries_to_csr    # noqa: E402
from easyrag_amd.index import BM25S, build_bm25_index             # noqa: E402


def main():
    rank, world = erd.init_from_env()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    eng = RetrievalEngine(local)
    n, d, vocab, B, topk = 20000, 256, 2000, 301, 10                # 301 queries: ragged shards
    x = synth.dense_corpus(n, d, seed=21)
    q = torch.from_numpy(synth.dense_queries(x, B, seed=22).astype(np.float16)).cuda()
    flat, lens = synth.token_corpus(n, vocab, seed=23, mean_len=24)
    idx = build_bm25_index([list(map(int, t)) for t in synth.split_docs(flat, lens)], BM25S)
    queries = [idx.tokens_to_ids(list(map(int, t))) for t in synth.token_queries(flat, lens, vocab, B, seed=24)]
    eng.set_dense(x)
    eng.set_bm25(idx)
    eng.set_doc_meta(n, None, None)

    def local_fn(lo, hi):
        qi, qt = queries_to_csr(queries[lo:hi])
        return eng.hybrid_topk(q[lo:hi], qi, qt, k_dense=288, k_sparse=192, K=60, topk=topk, device_out=True)

    want = [t.clone() for t in local_fn(0, B)]
    modes = {}
    for mode in ("torch", "native"):
        sh = erd.QueryShards(B, rank, world, engine=eng, mode=mode)
        for _ in range(2):                                          # the second step reuses the buffers
            got = sh.step(local_fn)
        torch.cuda.synchronize()
        ok = all(torch.equal(a, b) for a, b in zip(got, want))
        modes[mode] = (sh.mode, sh.fallback_reason, ok)
        if not ok:
            raise SystemExit(f"rank {rank}: gather mode {mode} (ran as {sh.mode}) differs from the unsharded result")
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        print("RCCL-GATHER-OK " + repr(modes), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
