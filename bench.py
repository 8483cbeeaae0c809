#!/usr/bin/env python
"""bench.py -- queries/sec of the dual-route coarse ranker (dense + BM25 + RRF top-10) at 1M x 1024 chunks.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it runs one rank per GPU (RCCL):
either launched under torch.distributed.run by the caller, or -- when WORLD_SIZE is not set -- bench.py starts its
own N ranks the same way.  A "step" is one pass of the hot path over one batch of synthetic queries: BM25 top-192 over ~50M CSR postings + dense cosine top-288 over the fp16 chunk matrix +
RRF(K=60) -> top-10 (BASELINE.json configs[3]; the depths are the reference's yaml defaults f_topk_2 / f_topk_1).
Per-GPU work is fixed (1024 queries per rank, corpus replicated): the global batch of N x 1024 queries is sharded
contiguously over the ranks (easyrag_amd.dist.QueryShards) and the fused top-k rows are all-gathered (configs[4] at
N = 8): weak scaling.  Consecutive steps use different query batches (a small rotating pool), so no step finds its
predecessor's postings or query tiles warm in L2 / MALL.  Inputs are resident in HBM before the timed region; rank 0 prints ONE
JSON line with the whole-job aggregate, the roofline of the dominant kernel (dense MFMA scan, timed with HIP
events on the launch stream inside the library) and the CPU baseline (oracle port, rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TF = 2500.0    # dense fp16/bf16 MFMA peak (no sparsity)
RIDGE = MFMA_F16_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)   # FLOP per byte


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="hybrid", choices=["hybrid", "dense", "bm25"])
    ap.add_argument("--chunks", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--vocab", type=int, default=262_144)
    ap.add_argument("--batch", type=int, default=0, help="queries per GPU per step (default: hybrid 1024 = configs[3]; dense / bm25 256 = configs[1] / configs[2])")
    ap.add_argument("--pool", type=int, default=4, help="distinct query batches rotated through the steps")
    ap.add_argument("--gather", default=None, choices=["torch", "native"], help="multi-GPU gather: torch.distributed around the library's pack/unpack kernels (default) or erh_allgather_topk (RCCL inside the library)")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"], help="process-group backend for --gpus > 1 (default nccl = RCCL; gloo with "
                                                                            "--share-device is the one-GPU dry run of the multi-process path)")
    ap.add_argument("--share-device", action="store_true", help="all ranks use GPU 0 (dry run of the N-process path on a one-GPU box: needs "
                                                                "--backend gloo; the numbers it prints are not a scaling measurement)")
    ap.add_argument("--dump-out", default=None, help="every rank saves the last step's (global) result as <path>.rank<r>.npz (tests)")
    ap.add_argument("--variant", default="bm25s", choices=["bm25s", "okapi"])
    ap.add_argument("--dirs", type=int, default=0, help="> 0: every query carries a dir filter on both routes, as the reference's real questions do: the corpus is "
                    "D contiguous blocks of documents (its loader walks the directories one after the other), query b asks for dir b %% D")
    ap.add_argument("--corpus", default="gaussian", choices=["gaussian", "clustered"], help="chunk embeddings: isotropic Gaussian directions (SURVEY 8d) or "
                    "the anisotropic topic-sorted corpus of synth.clustered_corpus_torch (random-pair cosine ~0.4, intra-topic ~0.7, 2 %% exact duplicates)")
    ap.add_argument("--qlen", default="fixed", choices=["fixed", "ref"], help="token queries of 10 tokens (SURVEY 8d) or with the length distribution of the "
                    "reference's 103 questions (4 ... 45 tokens, mean 10.8)")
    ap.add_argument("--cpu-queries", type=int, default=96, help="queries in the bounded CPU-baseline sample, ~15 s of host work (0 = skip)")
    ap.add_argument("--option", action="append", default=[], help="library option name=value (e.g. dense_n1=131072)")
    ap.add_argument("--sub", type=int, default=1, help="1 (default): after the headline run, time the other BASELINE.json configs and the "
                                                       "single-query latencies on the same corpus (sub_benchmarks in the JSON line; "
                                                       "hybrid workload, one GPU, default sizes only)")
    return ap.parse_args(argv)


class GpuPlatform:
    """Everything main() needs that touches the GPU or the library: device selection, synchronisation, timing events, the
    engine, the synthetic data and the index builder.  The CPU tests pass a stand-in with the same surface
    (tests/test_bench_multi_gpu.py: two gloo ranks through main()'s world > 1 control flow); the product path is this class."""

    def __init__(self):
        import torch
        from easyrag_amd import synth
        from easyrag_amd.engine import RetrievalEngine, queries_to_csr
        from easyrag_amd.index import build_bm25_index_from_postings
        self.torch, self.synth = torch, synth
        self._engine_cls = RetrievalEngine
        self.queries_to_csr = queries_to_csr
        self.build_index = build_bm25_index_from_postings

    def set_device(self, local: int):
        self.torch.cuda.set_device(local)
        return self.torch.device("cuda", local)

    def synchronize(self):
        self.torch.cuda.synchronize()

    def event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def make_engine(self, local: int):
        return self._engine_cls(local)


def cpu_baseline(x_dev, q16_dev, idx, payload, queries, k_dense, k_sparse, topk, n_sample, workload):
    """Oracle (reference restatement) timed on this box's host cores on a bounded sample of the same batch."""
    import torch
    from oracle import BM25SLucene, bm25_filter, qdrant_cosine_search, reciprocal_rank_fusion
    from oracle.retrievers import Item
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    n_sample = min(n_sample, len(queries) if queries else int(q16_dev.shape[0]))
    x32 = None
    if workload in ("hybrid", "dense"):
        x32 = x_dev.cpu().numpy().astype(np.float32)            # what a Qdrant local collection would hold (fp32)
        q32 = q16_dev[:n_sample].float().cpu().numpy()
    ora = None
    if workload in ("hybrid", "bm25"):
        ora = BM25SLucene()
        ora.data, ora.indices, ora.indptr, ora.num_docs = payload, idx.doc_ids, idx.indptr, idx.n_docs
    top_ids = []
    t_sparse = t_dense = 0.0
    t0 = time.perf_counter()
    for b in range(n_sample):
        sp = de = None
        if ora is not None:
            ts = time.perf_counter()
            if idx.variant == 1:
                scores = ora.get_scores_from_ids(queries[b])
            else:                                               # Okapi payloads: same scatter-add in float64
                scores = np.zeros(idx.n_docs)
                for t in queries[b]:
                    s, e = idx.indptr[t], idx.indptr[t + 1]
                    np.add.at(scores, idx.doc_ids[s:e], payload[s:e])
            sp = bm25_filter(scores, k_sparse, tie="literal")
            t_sparse += time.perf_counter() - ts
        if x32 is not None:
            ts = time.perf_counter()
            did, dsc = qdrant_cosine_search(x32, q32[b], k_dense, prenormalized=True)
            de = list(zip(did.tolist(), dsc.tolist()))
            t_dense += time.perf_counter() - ts
        if workload == "hybrid":
            fused = reciprocal_rank_fusion([[Item(i, i, s) for i, s in sp], [Item(i, i, s) for i, s in de]], K=60, topk=topk)
            top_ids.append([it.idx for it in fused])
        elif workload == "dense":
            top_ids.append([i for i, _ in de][:topk])
        else:
            top_ids.append([i for i, _ in sp][:topk])
    dt = time.perf_counter() - t0
    what = {"hybrid": "BM25(add.at over CSR, argsort, walk) + dense(np.dot fp32 1Mx1024, argsort, walk) + RRF",
            "dense": "dense(np.dot fp32, argsort, walk)", "bm25": "BM25(add.at over CSR, argsort, walk)"}[workload]
    out = {"value": n_sample / dt, "unit": "queries/s", "cores": int(threads), "kind": "port",
           "sample": f"{n_sample} queries of the same batch, one at a time as the reference does ({int(threads)}-thread BLAS "
                     f"inside np.dot, everything else one Python thread); {what}; {dt:.1f} s of CPU work"}
    if x32 is not None:
        # what a batching CPU implementation would do with the dense route (the reference does not): one fp32 GEMM for
        # the whole sample, top-k per row by partition + sort; the sparse route and the fusion as timed above
        tb = time.perf_counter()
        S = q32 @ x32.T
        part = np.argpartition(-S, min(k_dense, S.shape[1] - 1), axis=1)[:, :k_dense]
        rows = np.arange(S.shape[0])[:, None]
        order = np.argsort(-S[rows, part], axis=1, kind="stable")
        _ = part[rows, order]
        t_batched = time.perf_counter() - tb
        out["batched_dense"] = {"value": n_sample / (dt - t_dense + t_batched), "unit": "queries/s",
                                "what": f"dense route as ONE fp32 GEMM [{n_sample} x {x32.shape[0]}] + partition/sort per row "
                                        f"({t_batched:.1f} s instead of {t_dense:.1f} s); sparse route and fusion unchanged"}
    if ora is not None:
        out["okapi_literal"] = okapi_literal_loop(idx, queries)
    return out, top_ids


def okapi_literal_loop(idx, queries, n_docs=50_000, n_queries=8):
    """rank_bm25.BM25Okapi.get_scores as the library runs it -- `[(doc.get(q) or 0) for doc in doc_freqs]` per query
    token over one dict per document (the reference's default bm25_type 0) -- timed on the first `n_docs`
    documents of the corpus (building 1M Python dicts would take minutes; the loop is linear in the number of documents,
    and the figure for 1M documents is given as an explicit extrapolation, not as a measurement)."""
    from oracle import BM25Okapi
    n_docs = min(n_docs, idx.n_docs)
    t0 = time.perf_counter()
    docs = [[] for _ in range(n_docs)]
    for t in range(idx.n_vocab):                               # postings -> token lists (tf repeats) of the first n_docs documents
        s, e = int(idx.indptr[t]), int(idx.indptr[t + 1])
        if s == e:
            continue
        d = idx.doc_ids[s:e]
        cut = int(np.searchsorted(d, n_docs))
        for di, f in zip(d[:cut].tolist(), idx.tf[s:s + cut].tolist()):
            docs[di].extend([t] * f)
    ora = BM25Okapi(docs, k1=1.5, b=0.75, epsilon=0.25)
    t_build = time.perf_counter() - t0
    nq = min(n_queries, len(queries))
    t0 = time.perf_counter()
    for b in range(nq):
        ora.get_scores([int(t) for t in queries[b]])
    dt = time.perf_counter() - t0
    per_q = dt / max(nq, 1)
    return {"n_queries": nq, "n_docs": n_docs, "seconds": dt, "s_per_query": per_q, "index_build_s": t_build,
            "extrapolated_s_per_query_at_corpus_size": per_q * idx.n_docs / n_docs,
            "what": "rank_bm25 dict loop (one dict lookup per document and query token), get_scores only; "
                    "linear extrapolation in the number of documents, stated not measured"}


def pmc_traffic(args, kernel_class, algorithmic_bytes, launches_per_step=1.0, key=None):
    """HBM bytes per launch of the dominant kernel class from the committed PMC pass (scripts/gpu_traffic.sh ->
    profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE in its own run, x2 gfx950 correction).  bench.py cannot
    sample counters itself, so the figure is only attached when this run's shape is the profiled one (default
    sizes, no option overrides); otherwise traffic stays null."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    default_shape = (args.chunks == 1_000_000 and args.dim == 1024 and args.vocab == 262_144 and args.batch == 0
                     and not args.option and args.variant == "bm25s" and args.gpus == 1 and args.corpus == "gaussian"
                     and args.qlen == "fixed" and (args.dirs == 0 or key is not None))
    try:
        table = json.load(open(path))
        rec = table[key or args.workload]
    except (OSError, KeyError, ValueError):
        return {"traffic": None}
    from easyrag_amd import _build
    if table.get("_kernel_digest", table.get("_lib_digest")) != _build._kernel_digest():
        return {"traffic": None, "traffic_note": "profiles/pmc_traffic.json was collected on different kernel sources (stale)"}
    if not default_shape or rec.get("kernel_class") != kernel_class:
        return {"traffic": None}
    # bytes per STEP of the kernels of the class / this run's timed launches of the class per step (a timed launch may hold two
    # kernels: the BM25 scan and its -- normally empty -- redo launch)
    t = float(rec.get("hbm_bytes_per_step", rec["hbm_bytes_per_launch"] * rec.get("launches_per_step", 1.0))) / max(launches_per_step, 1e-9)
    return {"traffic": t, "traffic_unit": "bytes/launch", "traffic_over_algorithmic": t / algorithmic_bytes,
            "traffic_source": "profiles/pmc_traffic.json (" + rec["command"] + "; " + rec["correction"] + ")"}


def pmc_counters(args, key, main_class, class_ms_per_step, timed_classes=None):
    """MFMA-busy, L2 hit rate, LDS-fill (TD) path and effective shader clock of the dominant kernel from the committed PMC passes
    (scripts/gpu_pmc.sh -> scripts/pmc_summary.py -> profiles/pmc_counters.json), attached under the same guard as the FETCH_SIZE
    traffic: default sizes only, and only when the kernel sources of this run are the profiled ones (digest).  north_star:
    "rocprof HBM GB/s and MFMA-busy counters reported against gfx950 peak"."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_counters.json")
    default_shape = (args.chunks == 1_000_000 and args.dim == 1024 and args.vocab == 262_144 and args.batch == 0
                     and not args.option and args.variant == "bm25s" and args.gpus == 1 and args.corpus == "gaussian" and args.qlen == "fixed")
    try:
        table = json.load(open(path))
        entry = table[key]
    except (OSError, KeyError, ValueError):
        return None
    from easyrag_amd import _build
    if table.get("_kernel_digest") != _build._kernel_digest():
        return {"note": "profiles/pmc_counters.json was collected on different kernel sources (stale): not attached"}
    if not default_shape or main_class not in entry.get("classes", {}):
        return None
    d = entry["classes"][main_class]["derived"]
    out = {"kernel_class": main_class, "mfma_busy": d.get("mfma_busy"), "mfma_busy_what": "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs): share of the kernel's cycles the matrix pipe is busy (peak = 1.0)",
           "l2_hit": d.get("l2_hit"), "td_busy": d.get("td_busy"), "td_busy_what": "TD_TD_BUSY_sum / 256 CUs / cycles: the data-return path that carries the L1 -> LDS fills",
           "lds_bank_conflict_share": d.get("lds_bank_conflict_share"), "waves_waiting": d.get("waves_waiting"),
           "source": "profiles/pmc_counters.json <- " + entry.get("source", "?") + " (separate rocprofv3 --pmc passes, --kernel-trace only)"}
    # cycles of the kernels that the HIP-event class time covers (the sample pass has a timing class of its own)
    cyc = sum(c["derived"].get("gui_cycles_per_xcd", 0.0) for name, c in entry["classes"].items()
              if timed_classes is None or name in timed_classes)
    if cyc and class_ms_per_step:
        out["effective_clock_ghz"] = cyc / (class_ms_per_step * 1e-3) / 1e9
        out["effective_clock_what"] = ("GRBM_GUI_ACTIVE / 8 of the class's launches per step (PMC pass) / this run's HIP-event time of the class per step; "
                                       "the fp16 MFMA peak of 2.5 PF assumes 2.4 GHz")
    return out


def roofline_of(kd, dom):
    """Roofline block of one kernel class from the library's HIP-event time and the algorithmic work it booked."""
    if not kd["launches"]:
        return None
    sec = kd["ms"] * 1e-3
    ai = kd["flops"] / kd["bytes"] if kd["bytes"] else 0.0
    if dom == "dense_scan" and ai > RIDGE:
        ach = kd["flops"] / sec / 1e12
        roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_F16_PEAK_TF,
                "traffic": None}
    else:
        ach = kd["bytes"] / sec / 1e9
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None}
    if dom == "bm25_scan":
        roof["algorithmic_note"] = ("algorithmic bytes = SURVEY.md 8(d): 8 B per posting touched (document id + fp32 payload; Okapi 12 B); "
                                    "the default scan shape MOVES 4 B per posting (15-bit document offset in its tile + 16-bit "
                                    "fixed-point payload), so HBM traffic is expected well below the algorithmic figure")
    roof.update({"kernel": dom, "launches": int(kd["launches"]), "avg_launch_ms": kd["ms"] / kd["launches"],
                 "algorithmic_bytes_per_launch": kd["bytes"] / kd["launches"],
                 "flops_per_launch": kd["flops"] / kd["launches"], "arithmetic_intensity": ai,
                 "hbm_equiv_gbs": kd["bytes"] / sec / 1e9 if kd["bytes"] else None})
    return roof


def run_sub(eng, classes, fn, n_queries, dom, min_seconds=0.3, sync_each=False, what="", check_each=False, path=False):
    """One sub-benchmark on the resident corpus: enough steps for >= min_seconds of timed work, kernel classes from the
    library's event timers, roofline of `dom`.  sync_each: host-visible latency of single calls (B = 1)."""
    import torch
    for i in range(2):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    est = max((time.perf_counter() - t0) / 3, 1e-5)
    steps = int(min(4000, max(10, np.ceil(min_seconds / est))))
    eng.set_profiling(True)
    eng.reset_kernel_time()
    if path:
        eng.reset_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
        if check_each:
            eng.dense_check()                  # what a caller does before it reads device outputs: synchronise + the call's flag words
        elif sync_each:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.set_profiling(False)
    kt = {name: eng.kernel_time(cls) for name, cls in classes}
    rec = {"what": what, "queries_per_step": n_queries, "steps": steps, "ms_per_step": dt / steps * 1e3,
           "value": n_queries * steps / dt, "unit": "queries/s", "timed_seconds": dt,
           "kernel_ms_per_step": {k_: v["ms"] / steps for k_, v in kt.items()}}
    if path:
        # which kernels answered the timed steps, how many queries left the pruned pipeline, and how selective the thresholds were
        rec["path"] = {k_: v for k_, v in eng.stats().items() if v}
        rec["path"]["dense_exhaustive_queries"] = eng.stat("dense_exhaustive_queries")
        rec["dense_candidates_per_query_last_step"] = eng.dense_candidates_last_call() / max(1, n_queries)
    if dom is not None:
        rec["roofline"] = roofline_of(kt[dom], dom)
        if rec["roofline"] is not None:
            lps = rec["roofline"]["launches"] / steps
            rec["roofline"]["launches_per_step"] = lps
            if lps > 1.01:
                rec["roofline"]["launch_mix"] = (f"{lps:g} launches of the class per step (seed-prefix store kernel + append scan): "
                                                 "avg_launch_ms / *_per_launch are class totals / launches; kernel_ms_per_step holds the per-step class time")
    return rec


def sub_benchmarks(eng, synth, queries_to_csr, build_index, OKAPI, q16_pool, tok_pool, csr_pool, postings, pool, flat_lens=None):
    """The BASELINE.json configurations the headline is NOT quoted on, and the reference's own call pattern (one query at
    a time), timed on the corpus that is already resident: each with its own step count (>= 0.3 s of timed work),
    kernel-class times from the library's HIP events and a roofline block.  Roughly 10 s of wall time in all."""
    from easyrag_amd._lib import ERH_K_BM25_MERGE, ERH_K_BM25_SCAN, ERH_K_DENSE_SAMPLE, ERH_K_DENSE_SCAN, ERH_K_DENSE_SELECT, ERH_K_FUSE
    classes = (("dense_scan", ERH_K_DENSE_SCAN), ("dense_select", ERH_K_DENSE_SELECT), ("bm25_scan", ERH_K_BM25_SCAN),
               ("bm25_merge", ERH_K_BM25_MERGE), ("fuse", ERH_K_FUSE), ("dense_sample", ERH_K_DENSE_SAMPLE))
    out = {}
    q256 = [q[:256].contiguous() for q in q16_pool]
    csr256 = [queries_to_csr(t[:256]) for t in tok_pool]
    out["dense_b256_top100"] = run_sub(
        eng, classes, lambda i: eng.dense_topk(q256[i % pool], 100, device_out=True), 256, "dense_scan",
        what="configs[1]: 1M x 1024 fp16, dense cosine top-100 only, batch 256")
    out["bm25_b256_top100"] = run_sub(
        eng, classes, lambda i: eng.bm25_topk(*csr256[i % pool], 100, device_out=True), 256, "bm25_scan",
        what="configs[2]: 1M chunks, BM25 (bm25s, fp32) only top-100, batch 256")
    q1 = [q[:1].contiguous() for q in q16_pool]
    csr1 = [queries_to_csr(t[:1]) for t in tok_pool]
    out["dense_b1_top288_latency"] = run_sub(
        eng, classes, lambda i: eng.dense_topk(q1[i % pool], 288, device_out=True), 1, "dense_scan", sync_each=True, check_each=True,
        what="one query per call (the reference's call pattern), dense top-288, host-visible latency per call incl. erh_dense_check")
    out["hybrid_b1_latency"] = run_sub(
        eng, classes, lambda i: eng.hybrid_topk(q1[i % pool], *csr1[i % pool], k_dense=288, k_sparse=192, K=60, topk=10,
                                                device_out=True), 1, "dense_scan", sync_each=True, check_each=True,
        what="one query per call, dense(288)+BM25(192)+RRF top-10, host-visible latency per call incl. erh_dense_check")
    # the reference's default BM25 (bm25_type 0 = rank-bm25 Okapi, float64): a second index slot over the same postings
    indptr, doc, tf, lens = postings
    idx_ok = build_index(indptr, doc, tf, lens, OKAPI, compute_payload=False)
    eng.set_bm25(idx_ok, payload_on_device=True, slot=1)
    try:
        out["hybrid_okapi_b1024"] = run_sub(
            eng, classes, lambda i: eng.hybrid_topk(q16_pool[i % pool], *csr_pool[i % pool], k_dense=288, k_sparse=192, K=60,
                                                    topk=10, device_out=True, slot=1), int(q16_pool[0].shape[0]), "dense_scan",
            what="configs[3] with the reference's default bm25_type 0 (Okapi, float64 scores)")
        kt = out["hybrid_okapi_b1024"]
        out["bm25_okapi_b256_top100"] = run_sub(
            eng, classes, lambda i: eng.bm25_topk(*csr256[i % pool], 100, device_out=True, slot=1), 256, "bm25_scan",
            what="configs[2] with Okapi (float64) scores, batch 256")
        del kt
    finally:
        eng.free_bm25_slot(1)
        eng._select(0)
    # the reference's REAL call pattern carries a `dir` filter on every query (all 103 questions of src/data/question.jsonl name their
    # document; pipeline.py:301-312 -> filter_dict / qdrant filters, retrievers.py:278,283): four dirs as four contiguous blocks of
    # documents -- `dir` is the first path component and the reference's loader walks the directories one after the other
    # (transformation.py:70, ingestion.py:79-87) --, query b asks for dir b % 4, the same column pushed down into both routes
    n_docs = int(eng.n_dense)
    eng.set_doc_meta(n_docs, None, (np.arange(n_docs) * 4 // n_docs).astype(np.int16))
    filt = (np.arange(int(q16_pool[0].shape[0])) % 4).astype(np.int16)
    try:
        eng.reset_stats()
        out["hybrid_b1024_dir_filter"] = run_sub(
            eng, classes, lambda i: eng.hybrid_topk(q16_pool[i % pool], *csr_pool[i % pool], k_dense=288, k_sparse=192, K=60, topk=10,
                                                    device_out=True, filter_dir=filt), int(q16_pool[0].shape[0]), "dense_scan",
            what="configs[3] with a per-query dir filter on both routes (4 dirs = 4 contiguous blocks of 250k chunks): the dense route as 4 groups "
                 "of 256 queries, each over its dir's block copy, all four in ONE launch per stage (dense_dir_blocks + dense_group_launch: a view "
                 "table per 256-query tile read by the store kernel, seed select, persistent scan on the 256 x 256 tile and the final kernel), "
                 "the BM25 scan over the tiles of the query's dir only")
        out["hybrid_b1024_dir_filter"]["dense_block_groups_per_step"] = eng.stat("dense_block_groups") / max(1, eng.stat("hybrid_calls"))
        out["hybrid_b1024_dir_filter"]["dense_grouped_launches_per_step"] = eng.stat("dense_grouped_launches") / max(1, eng.stat("hybrid_calls"))
        f1 = [filt[p % 4:p % 4 + 1].copy() for p in range(pool)]
        out["hybrid_b1_dir_filter_latency"] = run_sub(
            eng, classes, lambda i: eng.hybrid_topk(q1[i % pool], *csr1[i % pool], k_dense=288, k_sparse=192, K=60, topk=10,
                                                    device_out=True, filter_dir=f1[i % pool]), 1, "dense_scan", sync_each=True, check_each=True,
            what="one query per call WITH its dir filter (the reference's real call), dense(288)+BM25(192)+RRF top-10 over the dir's block and "
                 "posting tiles, host-visible latency per call incl. erh_dense_check")
    finally:
        eng.set_doc_meta(n_docs, None, None)
    # Is the timing an artefact of the synthetic distributions?    (b) first, on the resident corpus: token queries with
    # the length distribution of the reference's 103 real questions (4 ... 45 tokens, mean 10.8) instead of ten tokens each; then (a) a
    # second chunk matrix -- anisotropic and topic-sorted like real embedding corpora (two random chunks ~0.4 cosine, one topic ~0.7,
    # 2 % exact duplicates), queries near corpus members -- under the same BM25 index.  Both report the path counters and the
    # candidates per query next to the Gaussian / fixed-length figures of the headline.
    import torch
    Bq = int(q16_pool[0].shape[0])
    if flat_lens is not None:
        flat, lens_, vocab = flat_lens
        tok_ref = [synth.token_queries(flat, lens_, vocab, Bq, seed=4000 + p, lengths=synth.REF_QUESTION_LENGTHS) for p in range(pool)]
        csr_ref = [queries_to_csr(t) for t in tok_ref]
        out["hybrid_b1024_ref_query_lengths"] = run_sub(
            eng, classes, lambda i: eng.hybrid_topk(q16_pool[i % pool], *csr_ref[i % pool], k_dense=288, k_sparse=192, K=60, topk=10,
                                                    device_out=True), Bq, "dense_scan", path=True,
            what="configs[3] with token queries whose lengths follow the reference's 103 questions (4 ... 45 tokens, mean 10.8; 80 % of a "
                 "query's tokens from its target document, repeats allowed)")
        out["hybrid_b1024_ref_query_lengths"]["mean_query_tokens"] = float(np.mean([len(t) for t in tok_ref[0]]))
        out["hybrid_b1024_ref_query_lengths"]["bm25_redo_segments"] = eng.stat("bm25_redo_segments")
    n_docs, d_ = int(eng.n_dense), int(eng.d)
    xc = synth.clustered_corpus_torch(n_docs, d_, seed=21, device=q16_pool[0].device)
    qc = [synth.dense_queries_torch(xc, Bq, seed=5000 + p) for p in range(pool)]
    eng.set_dense(xc)
    out["hybrid_b1024_clustered"] = run_sub(
        eng, classes, lambda i: eng.hybrid_topk(qc[i % pool], *csr_pool[i % pool], k_dense=288, k_sparse=192, K=60, topk=10,
                                                device_out=True), Bq, "dense_scan", path=True,
        what="configs[3] on ANISOTROPIC chunk embeddings: topic-sorted rows sqrt(.4) m + sqrt(.3) c_topic + sqrt(.3) noise (2000 topics: "
             "random-pair cosine ~0.4, intra-topic ~0.7), 2 % exact duplicates, queries = noisy copies of corpus rows; same BM25 index")
    eng.dense_check()
    out["hybrid_b1024_clustered"]["dense_exhaustive_queries_last_call"] = eng.dense_diag()["exhaustive"]
    q256c = [q[:256].contiguous() for q in qc]
    out["dense_b256_top100_clustered"] = run_sub(
        eng, classes, lambda i: eng.dense_topk(q256c[i % pool], 100, device_out=True), 256, "dense_scan", path=True,
        what="configs[1] on the anisotropic corpus")
    del xc, qc, q256c
    torch.cuda.empty_cache()
    # the reference's own vector size (ref src/configs/easyrag.yaml:15-16: gte-Qwen2-7B, vector_size 3584): the same 2.05 GB of chunk
    # matrix as 285 696 x 3584 fp16, dense top-100, batch 256 -- LAST, because it replaces the resident 1M x 1024 matrix
    n35, d35 = 285_696, 3584
    x35 = synth.dense_corpus_torch(n35, d35, seed=12, device=q16_pool[0].device)
    q35 = [synth.dense_queries_torch(x35, 256, seed=3000 + p) for p in range(pool)]
    eng.set_dense(x35)
    out["dense_d3584_b256"] = run_sub(
        eng, classes, lambda i: eng.dense_topk(q35[i % pool], 100, device_out=True), 256, "dense_scan",
        what=f"the reference's vector_size: {n35} chunks x {d35}-d fp16 (2.05 GB, as configs[1]), dense cosine top-100, batch 256")
    eng.dense_check()
    out["dense_d3584_b256"]["dense_exhaustive_queries_last_call"] = eng.dense_diag()["exhaustive"]
    del x35, q35
    torch.cuda.empty_cache()
    return out


def shared_sparse_corpus(synth, n, vocab, n_global, pool, qlen, dev, rank, world):
    """The token corpus (CSR postings) and the query pools of a multi-rank run, generated ONCE per node: rank 0 builds them and writes
    them to /dev/shm (keyed by the shapes, the seeds and the job's rendezvous port), the others wait at a barrier and read the files --
    eight ranks of one node would otherwise each sort 56 M tokens and walk them on the same host cores at the same time.
    Returns (indptr, doc, tf, lens, flat or None, tok_pool_global, shared: bool); world == 1 generates in place and keeps `flat`."""
    import tempfile
    import torch
    lengths = synth.REF_QUESTION_LENGTHS if qlen == "ref" else None

    def generate():
        indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
        pools = [synth.token_queries(flat, lens, vocab, n_global, seed=2000 + p, lengths=lengths) for p in range(pool)]
        return indptr, doc, tf, lens, flat, pools

    if world == 1:
        return (*generate(), False)
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    key = f"erh_bench_n{n}_v{vocab}_q{n_global}_p{pool}_{qlen}_{os.environ.get('MASTER_PORT', '0')}"
    path = os.path.join(base, key)
    if rank == 0:
        indptr, doc, tf, lens, flat, pools = generate()
        del flat
        os.makedirs(path, exist_ok=True)
        np.save(os.path.join(path, "indptr.npy"), indptr)
        np.save(os.path.join(path, "doc.npy"), doc)
        np.save(os.path.join(path, "tf.npy"), tf)
        np.save(os.path.join(path, "lens.npy"), lens)
        for p, qs in enumerate(pools):                                   # ragged: one flat array + offsets per pool
            off = np.zeros(len(qs) + 1, np.int64)
            off[1:] = np.cumsum([len(q) for q in qs])
            np.save(os.path.join(path, f"q{p}_tok.npy"), np.concatenate(qs).astype(np.int32) if qs else np.zeros(0, np.int32))
            np.save(os.path.join(path, f"q{p}_off.npy"), off)
    torch.distributed.barrier()                                          # the files are complete
    if rank != 0:
        indptr, doc, tf, lens = (np.load(os.path.join(path, f"{name}.npy"), mmap_mode="r") for name in ("indptr", "doc", "tf", "lens"))
        pools = []
        for p in range(pool):
            tok, off = np.load(os.path.join(path, f"q{p}_tok.npy")), np.load(os.path.join(path, f"q{p}_off.npy"))
            pools.append([tok[off[i]:off[i + 1]] for i in range(off.shape[0] - 1)])
    return indptr, doc, tf, lens, None, pools, path


def spawn_ranks(args, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run on this node (what
    the driver's wrapped form does) and relay their output."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not args.share_device:
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs on this node, found {have} "
              f"(one rank per GPU; RCCL does not share a device between ranks)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    return subprocess.call(cmd, env=env)


def main(argv=None, platform=None):
    """The benchmark.  `argv` / `platform` default to the command line and the GPU (GpuPlatform); rank 0 prints ONE JSON line
    and every rank returns {"record": that line's dict (rank 0) or None, "out": the last step's (global) result}."""
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args, argv))
    import torch
    from easyrag_amd import dist as erd
    from easyrag_amd._lib import ERH_K_BM25_MERGE, ERH_K_BM25_SCAN, ERH_K_DENSE_SAMPLE, ERH_K_DENSE_SCAN, ERH_K_DENSE_SELECT, ERH_K_FUSE
    from easyrag_amd.index import BM25S, OKAPI
    plat = platform if platform is not None else GpuPlatform()
    synth, queries_to_csr, build_bm25_index_from_postings = plat.synth, plat.queries_to_csr, plat.build_index

    t_setup = time.perf_counter()
    if args.share_device and args.backend != "gloo" and args.gpus > 1:
        raise SystemExit("--share-device needs --backend gloo (RCCL does not share a device between ranks)")
    local = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    rank, world = erd.init_from_env(device_index=local, backend=args.backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = plat.set_device(local)
    n, d, vocab = args.chunks, args.dim, args.vocab
    B = args.batch or (1024 if args.workload == "hybrid" else 256)
    k_dense, k_sparse, topk = (100, 0, 100) if args.workload == "dense" else (288, 192, 10)
    if args.workload == "bm25":
        k_dense, k_sparse, topk = 0, 100, 100
    variant = BM25S if args.variant == "bm25s" else OKAPI
    n_global = B * world
    pool = max(1, args.pool)

    # ---- synthetic corpus, replicated on every rank (same seeds); the global query batches are the same on every
    # rank too, and each rank answers its contiguous shard of them -------------------------------------------------
    eng = plat.make_engine(local)
    for opt in args.option:
        name, val = opt.split("=")
        eng.set_option(name, int(val))
    shards = erd.QueryShards(n_global, rank, world, engine=eng, mode=args.gather)
    lo, hi = shards.bounds
    x = idx = None
    corpus_shared = False
    flat_keep = None
    q16_pool, csr_pool, tok_pool = [], [], []
    if args.workload in ("hybrid", "dense"):
        x = (synth.clustered_corpus_torch(n, d, seed=21, device=dev) if args.corpus == "clustered"
             else synth.dense_corpus_torch(n, d, seed=2, device=dev))
        q16_pool = [synth.dense_queries_torch(x, n_global, seed=1000 + p)[lo:hi].contiguous() for p in range(pool)]
        eng.set_dense(x)
    if args.workload in ("hybrid", "bm25"):
        indptr, doc, tf, lens, flat, tok_global, shared_path = shared_sparse_corpus(synth, n, vocab, n_global, pool, args.qlen, dev, rank, world)
        idx = build_bm25_index_from_postings(indptr, doc, tf, lens, variant, compute_payload=False)
        eng.set_bm25(idx, payload_on_device=True)            # IDF*TF/(TF + k1*lenNorm) evaluated by the GPU
        for p in range(pool):
            tok_pool.append(tok_global[p][lo:hi])
            csr_pool.append(queries_to_csr(tok_pool[-1]))
        flat_keep = (flat, lens, vocab) if (args.sub and world == 1 and args.workload == "hybrid") else None   # (the sub-benchmarks draw further queries)
        del flat, tok_global
        if shared_path:
            plat.synchronize()                               # every rank's uploads out of the shared files are done ...
            torch.distributed.barrier()
            if rank == 0:                                    # ... before rank 0 removes them
                import shutil
                shutil.rmtree(shared_path, ignore_errors=True)
            corpus_shared = True
    filt = None
    if args.dirs > 0:
        eng.set_doc_meta(n, None, (np.arange(n, dtype=np.int64) * args.dirs // n).astype(np.int16))
        filt = ((lo + np.arange(hi - lo)) % args.dirs).astype(np.int16)
    else:
        eng.set_doc_meta(n, None, None)
    plat.synchronize()
    q16 = q16_pool[0] if q16_pool else None
    queries = tok_pool[0] if tok_pool else []

    def local_step(p):
        if args.workload == "hybrid":
            qi, qt = csr_pool[p]
            if filt is not None:
                return eng.hybrid_topk(q16_pool[p], qi, qt, k_dense=k_dense, k_sparse=k_sparse, K=60, topk=topk, device_out=True, filter_dir=filt)
            return eng.hybrid_topk(q16_pool[p], qi, qt, k_dense=k_dense, k_sparse=k_sparse, K=60, topk=topk, device_out=True)
        if args.workload == "dense":
            return eng.dense_topk(q16_pool[p], k_dense, device_out=True, **({} if filt is None else {"filter_dir": filt}))
        qi, qt = csr_pool[p]
        return eng.bm25_topk(qi, qt, k_sparse, device_out=True, **({} if filt is None else {"filter_dir": filt}))

    counter = [0]

    gather_events = []

    def step():
        p = counter[0] % pool
        counter[0] += 1
        out = local_step(p)
        if world > 1:
            ev = (plat.event(), plat.event())
            ev[0].record()
            out = shards.gather(*out)                         # erh_dense_check, pack -> ONE all-gather -> unpack
            ev[1].record()
            gather_events.append(ev)
        return out

    out = None
    for _ in range(args.warmup):
        step()
    plat.synchronize()
    setup_s = time.perf_counter() - t_setup                   # process start -> ready for the timed region (corpus, index, warm-up)
    if args.workload != "bm25":
        eng.dense_check()                                     # raises on candidate overflow
    eng.set_profiling(True)
    eng.reset_kernel_time()
    if hasattr(eng, "reset_stats"):
        eng.reset_stats()                                     # the counters below cover exactly the timed steps
    gather_events.clear()
    if world > 1:
        torch.distributed.barrier()
    plat.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    plat.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    setup_all = [setup_s]
    if world > 1:
        host_side = torch.distributed.get_backend() == "gloo"       # (gloo reduces host tensors)
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if host_side else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        setup_all = [None] * world
        torch.distributed.all_gather_object(setup_all, setup_s)
    eng.set_profiling(False)
    if args.workload != "bm25":
        eng.dense_check()
    path_stats = eng.stats() if hasattr(eng, "stats") else None    # (reads the device counters: synchronises, outside the timed region)
    if path_stats is not None and args.workload != "bm25" and hasattr(eng, "dense_candidates_last_call"):
        path_stats["dense_candidates_per_query_last_step"] = eng.dense_candidates_last_call() / max(1, B)

    rec = None
    if rank == 0:
        kt = {name: eng.kernel_time(cls) for name, cls in
              (("dense_scan", ERH_K_DENSE_SCAN), ("dense_select", ERH_K_DENSE_SELECT), ("bm25_scan", ERH_K_BM25_SCAN),
               ("bm25_merge", ERH_K_BM25_MERGE), ("fuse", ERH_K_FUSE), ("dense_sample", ERH_K_DENSE_SAMPLE))}
        per_step = {k_: (v["ms"] / args.steps) for k_, v in kt.items()}
        dom = "bm25_scan" if args.workload == "bm25" else "dense_scan"
        kd = kt[dom]
        roof = roofline_of(kd, dom)
        if roof is not None:
            roof["launches_per_step"] = roof["launches"] / args.steps
            filtered4 = args.workload == "hybrid" and args.dirs == 4      # the filtered step has PMC passes of its own (bench.py --dirs 4)
            if args.dirs == 0 or filtered4:
                roof.update(pmc_traffic(args, dom, roof["algorithmic_bytes_per_launch"], roof["launches_per_step"],
                                        key="hybrid_dirs4" if filtered4 else None))
            if args.workload != "bm25" and (args.dirs == 0 or filtered4):
                key, main = (("hybrid_dirs4_b1024", "pp3") if filtered4 else ("dense_b1024", "pp5")) if args.workload == "hybrid" else ("dense_b256", "pp3")
                roof["counters"] = pmc_counters(args, key, main, per_step["dense_scan"],
                                                ("pp5",) if (args.workload == "hybrid" and not filtered4) else ("pp3", "store"))
            if dom == "dense_scan" and roof["launches"] > args.steps:
                roof["launch_mix"] = ("the dense-scan class has two launches per step -- the seed-prefix store kernel (real scan work over "
                                      "the first rows) and the append scan over the rest; per-launch figures are class totals / launches, "
                                      "rocprofv3 lists the two kernels separately.  (From 512 queries on the threshold comes from a sample "
                                      "pass with its own class, dense_sample, and the scan class is ONE kernel.)")
        cpu = None
        if world == 1 and args.cpu_queries > 0:
            payload = payload_check = None
            if idx is not None:
                # the CPU baseline scores with a payload evaluated on the HOST (easyrag_amd.index: the libraries' arithmetic in
                # numpy), not with what the GPU computed; the two must agree bit for bit, and the line says whether they did
                host_idx = build_bm25_index_from_postings(indptr, doc, tf, lens, variant, compute_payload=True)
                payload = host_idx.payload
                gpu_payload = eng.get_bm25_payload()
                payload_check = bool(payload.dtype == gpu_payload.dtype and np.array_equal(payload, gpu_payload))
                if not payload_check:
                    raise SystemExit("bench.py: the GPU's BM25 payload differs from the host evaluation (parity broken)")
                del host_idx, gpu_payload
            cpu, ref_ids = cpu_baseline(x, q16, idx, payload, queries, k_dense, k_sparse, topk, args.cpu_queries,
                                        args.workload)
            # recall@topk of the GPU result against the CPU restatement on the same sample (sets: the CPU walk
            # orders equal scores as numpy's argsort happens to, the GPU by index)
            got = local_step(0)
            plat.synchronize()
            g_ids = got[0][: len(ref_ids)].cpu().numpy()
            hit = tot = 0
            for b, want in enumerate(ref_ids):
                tot += len(want)
                hit += len(set(want) & set(int(v) for v in g_ids[b] if v >= 0))
            cpu["recall_at_topk_vs_cpu"] = (hit / tot) if tot else None
            if payload_check is not None:
                cpu["payload"] = ("host payload (numpy evaluation of idf * tf-saturation per posting, easyrag_amd.index); the GPU's "
                                  "payload array is bit-identical to it" if payload_check else "MISMATCH")
            cpu["recall_note"] = ("id sets of the two results; below 1.0 only where documents tie at the k-th score: the CPU "
                                  "walk orders equal scores as numpy's argsort happens to, the GPU by index")
        total_q = B * world * args.steps
        workload_name = {
            "hybrid": f"configs[3]: 1M chunks, dual-route dense(top-{k_dense})+BM25(top-{k_sparse}) with RRF top-{topk}",
            "dense": f"configs[1]: 1M chunks x 1024-d fp16, dense cosine top-{k_dense} only",
            "bm25": f"configs[2]: 1M chunks, BM25 only top-{k_sparse}, batch {B}"}[args.workload]
        rec = {
            "metric": "queries/sec (dense+BM25+RRF top-10) at 1Mx1024 chunks" if args.workload == "hybrid"
                      else f"queries/sec ({args.workload} only) at 1Mx1024 chunks",
            "value": total_q / dt, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if args.workload != "bm25" else ("f32" if variant == BM25S else "f64"),
            "data": "synthetic",
            "config": {"workload": workload_name, "chunks": n, "dim": d, "vocab": vocab,
                       "postings": int(idx.nnz) if idx is not None else 0,
                       "queries_per_gpu": B, "global_batch": B * world,
                       "bm25_variant": args.variant if idx is not None else None,
                       "query_batches_rotated": pool, "corpus": args.corpus, "query_lengths": args.qlen, "dir_filter_dirs": args.dirs,
                       "parallelism": f"query-sharded x{world}, corpus replicated, all-gather of fused top-k"
                                      + (f" ({shards.mode})" if world > 1 else "")},
            "multi_gpu": None if world == 1 else {
                "rccl_ranks": int(torch.distributed.get_world_size()), "backend": torch.distributed.get_backend(),
                "gather_mode": shards.mode, "gather_fallback_reason": shards.fallback_reason,
                "transport": getattr(shards, "transport", None), "shared_device": bool(args.share_device),
                "setup_s_per_rank": [round(float(v), 3) for v in setup_all],
                "allgather_ms_per_step": (sum(a.elapsed_time(b) for a, b in gather_events) / max(len(gather_events), 1)),
                "allgather_what": "rank 0: dense_check + pack kernel + all_gather_into_tensor + unpack kernel (CUDA events)",
                # the token corpus and the query pools were generated once (rank 0) and read by the other ranks from /dev/shm
                "corpus_shared": corpus_shared},
            "roofline": roof,
            "cpu_baseline": cpu,
            "kernel_ms_per_step": per_step,
            # which kernels answered the TIMED steps (erh_get_stat; rank 0): the record proves its own path -- every query of every
            # timed step stayed on the pruned dense pipeline iff dense_exhaustive_queries == 0 (the exhaustive path's later rounds
            # would run in the un-timed erh_dense_check), and no BM25 (query, segment) fell back to the exact block scan
            "path": path_stats,
        }
        default_shape = (args.chunks == 1_000_000 and args.dim == 1024 and args.vocab == 262_144 and args.batch == 0
                         and not args.option and args.variant == "bm25s" and args.corpus == "gaussian" and args.qlen == "fixed" and args.dirs == 0)
        if args.sub and world == 1 and args.workload == "hybrid" and default_shape:
            rec["sub_benchmarks"] = sub_benchmarks(eng, synth, queries_to_csr, build_bm25_index_from_postings, OKAPI,
                                                   q16_pool, tok_pool, csr_pool, (indptr, doc, tf, lens), pool, flat_lens=flat_keep)
            # the committed PMC passes of the other BASELINE configurations (same digest guard as the headline's): MFMA-busy / TD / L2 hit /
            # effective clock and the FETCH_SIZE traffic of each sub-benchmark's dominant kernel class
            for name, key, main, timed, tkey in (("dense_b256_top100", "dense_b256", "pp3", ("pp3", "store"), "dense"),
                                                 ("bm25_b256_top100", "bm25_b256", "ascan", ("ascan",), "bm25"),
                                                 ("hybrid_b1024_dir_filter", "hybrid_dirs4_b1024", "pp3", ("pp3", "store"), "hybrid_dirs4")):
                sub = rec["sub_benchmarks"].get(name)
                if not sub or not sub.get("roofline"):
                    continue
                cls = "bm25_scan" if name.startswith("bm25") else "dense_scan"
                sub["roofline"]["counters"] = pmc_counters(args, key, main, sub["kernel_ms_per_step"].get(cls), timed_classes=timed)
                sub["roofline"].update(pmc_traffic(args, cls, sub["roofline"]["algorithmic_bytes_per_launch"],
                                                   sub["roofline"].get("launches_per_step", 1.0), key=tkey))
        print(json.dumps(rec), flush=True)
    if args.dump_out and out is not None:
        plat.synchronize()
        np.savez(f"{args.dump_out}.rank{rank}.npz", **{name: (t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t))
                                                       for name, t in zip(("ids", "scores", "len"), out)})
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    eng.close()
    return {"record": rec, "out": out}


if __name__ == "__main__":
    main()
