#!/usr/bin/env python
"""Kernel micro-bench on the 1M-chunk shapes: sweeps library options (ablations, stage sizes) in ONE process
and prints per-class kernel milliseconds from the library's HIP-event timers.  Measurement tool only."""
import json
import os
import sys

os.environ["ERH_MEASURE"] = "1"          # measurement variants of the kernels (see easyrag_amd/_build.py)

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from easyrag_amd import synth  # noqa: E402
from easyrag_amd._lib import ERH_K_BM25_SCAN, ERH_K_DENSE_SCAN, ERH_K_DENSE_SELECT, ERH_K_FUSE  # noqa: E402
from easyrag_amd.engine import RetrievalEngine, queries_to_csr  # noqa: E402
from easyrag_amd.index import BM25S, OKAPI, build_bm25_index_from_postings  # noqa: E402


def timed(eng, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    eng.set_profiling(True)
    eng.reset_kernel_time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    eng.set_profiling(False)
    out = {}
    for name, cls in (("dense_scan", ERH_K_DENSE_SCAN), ("dense_select", ERH_K_DENSE_SELECT),
                      ("bm25_scan", ERH_K_BM25_SCAN), ("fuse", ERH_K_FUSE)):
        kt = eng.kernel_time(cls)
        if kt["launches"]:
            out[name] = round(kt["ms"] / reps, 4)
    return out


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    dev = torch.device("cuda", 0)
    n, d, vocab = 1_000_000, 1024, 262_144
    eng = RetrievalEngine(0)
    res = {}
    if what == "n0n1":                                       # seed / refine boundaries
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        for B, k in ((256, 100), (1024, 288)):
            q = synth.dense_queries_torch(x, B, seed=7)
            for n0, n1 in ((32768, 131072), (32768, 0), (32768, 262144), (32768, 393216), (16384, 262144), (32768, 131072), (32768, 0), (32768, 262144), (32768, 393216), (16384, 262144)):
                eng.set_option("dense_n0", n0)
                eng.set_option("dense_n1", n1)
                res[f"dense B={B} k={k} n0={n0} n1={n1} #{len(res)}"] = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
            eng.set_option("dense_n0", 32768)
            eng.set_option("dense_n1", 131072)
    if what == "n1auto":
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        for B, k in ((256, 100), (1024, 288)):
            q = synth.dense_queries_torch(x, B, seed=7)
            for auto in (0, 1, 0, 1):
                eng.set_option("dense_n1_auto", auto)
                res[f"dense B={B} k={k} n1_auto={auto} #{len(res)}"] = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
        eng.set_option("dense_n1_auto", 1)
    if what == "scale":                                      # time per chunk vs corpus size (is the scan memory-side bound?)
        for nn in (65536 + 32768, 262144 + 32768, 1_000_000):
            x = synth.dense_corpus_torch(nn, d, seed=2, device=dev)
            eng.set_dense(x)
            eng.set_option("dense_n1", 0)
            for B, k in ((256, 100), (1024, 288)):
                q = synth.dense_queries_torch(x, B, seed=7)
                for abl in (0, 7):
                    eng.set_option("dense_ablate", abl)
                    r = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
                    r["us_per_1k_chunks"] = round(1e3 * r["dense_scan"] / (nn / 1000), 4)
                    res[f"dense N={nn} B={B} pabl={abl}"] = r
            eng.set_option("dense_ablate", 0)
            eng.set_option("dense_n1", 131072)
    if what == "fin":                                        # finalize: exact (fp64 re-score) vs fast (fp32 order only)
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        for B, k in ((256, 100), (1024, 288)):
            q = synth.dense_queries_torch(x, B, seed=7)
            for mode in (0, 1, 0, 1):
                res[f"dense B={B} k={k} mode={mode} ({len(res)})"] = timed(eng, lambda: eng.dense_topk(q, k, mode=mode, device_out=True))
    if what == "pp":                                         # ping-pong scan vs the persistent kernel, with ablations
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        for B, k in ((256, 100), (1024, 288)):
            q = synth.dense_queries_torch(x, B, seed=7)
            eng.set_option("dense_pp", 0)
            for cfg in ():
                for abl in (0, 7):
                    eng.set_option("dense_cfg", cfg)
                    eng.set_option("dense_ablate", abl)
                    res[f"dense B={B} k={k} persist cfg={cfg} pabl={abl}"] = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
            eng.set_option("dense_cfg", 0)
            for rep in "ab":
                for pp, abl in ((1, 0), (1, 8), (1, 7), (1, 11), (1, 12)):
                    eng.set_option("dense_pp", pp)
                    eng.set_option("dense_ablate", abl)
                    res[f"dense B={B} k={k} pp={pp} pabl={abl} (run {rep})"] = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
            eng.set_option("dense_ablate", 0)
            eng.set_option("dense_pp", 1)
    if what in ("pp2", "pp2q"):                              # ping-pong kernels: operand-side ablations and phase clocks
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        lean = int(os.environ.get("KB_PP", "2"))
        eng.set_option("dense_pp", lean)
        for B, k in ((256, 100), (1024, 288)):
            q = synth.dense_queries_torch(x, B, seed=7)
            for rep in "ab":
                for abl in ((0, 7, 12, 11, 16, 17, 18) if what == "pp2q" else (0, 7, 12, 13, 15, 14, 11, 16, 17, 18)):
                    eng.set_option("dense_ablate", abl)
                    res[f"dense B={B} k={k} pp{lean} pabl={abl} (run {rep})"] = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
            eng.set_option("debug_counters", 1)
            eng.set_option("dense_ablate", 20)
            eng.dense_topk(q, k, device_out=True)
            torch.cuda.synchronize()
            c = eng.debug_counters().astype(np.float64)
            eng.set_option("dense_ablate", 0)
            eng.set_option("debug_counters", 0)
            names = ["matrix", "wait", "barrier", "memory", "epilogue", "epi_barrier"]
            for g in (0, 1):
                stages = c[g * 8 + 6]
                res[f"dense B={B} pp{lean} phase clocks per stage, group {g}"] = {
                    n_: round(v / stages, 1) for n_, v in zip(names, c[g * 8:g * 8 + 6])} if stages else {}
        del x
    if what == "var":                                        # lean ping-pong scan: barrier / priority variants x K-rotation
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        eng.set_option("dense_pp", 2)
        for B, k in ((256, 100), (1024, 288)):
            q = synth.dense_queries_torch(x, B, seed=7)
            combos = [(v, 0) for v in (0, 1, 2, 3)] if B == 256 else [(v, r) for r in (0, 8) for v in (0, 1, 2, 3)] + [(0, 4), (0, 16), (1, 16)]
            for rep in "ab":
                for var, rot in combos:
                    eng.set_option("dense_var", var)
                    eng.set_option("dense_rot", rot)
                    for abl in (0, 7):
                        eng.set_option("dense_ablate", abl)
                        res[f"dense B={B} var={var} rot={rot} pabl={abl} (run {rep})"] = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
            eng.set_option("dense_ablate", 0)
            for var, rot in ((0, 0), (1, 0)):
                eng.set_option("dense_var", var)
                eng.set_option("dense_rot", rot)
                eng.set_option("debug_counters", 1)
                eng.set_option("dense_ablate", 20)
                eng.dense_topk(q, k, device_out=True)
                torch.cuda.synchronize()
                c = eng.debug_counters().astype(np.float64)
                eng.set_option("dense_ablate", 0)
                eng.set_option("debug_counters", 0)
                names = ["matrix", "wait", "barrier", "memory", "epilogue", "epi_barrier"]
                for g in (0, 1):
                    stages = c[g * 8 + 6]
                    res[f"dense B={B} var={var} phase clocks per stage, group {g}"] = {
                        n_: round(v / stages, 1) for n_, v in zip(names, c[g * 8:g * 8 + 6])} if stages else {}
            eng.set_option("dense_var", 0)
            eng.set_option("dense_rot", 0)
        del x
    if what == "mm":                                         # matrix-side ablations of the lean ping-pong scan with phase clocks
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        eng.set_option("dense_pp", 2)
        names = ["matrix", "wait", "barrier", "memory", "epilogue", "epi_barrier"]
        for B, k in ((256, 100), (1024, 288)):
            q = synth.dense_queries_torch(x, B, seed=7)
            for var in (0, 1):
                eng.set_option("dense_var", var)
                for rep in "ab":
                    for abl in (7, 12, 23, 17, 18):
                        eng.set_option("dense_ablate", abl)
                        res[f"dense B={B} var={var} pabl={abl} (run {rep})"] = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
                for abl in (21, 20, 22, 24):
                    eng.set_option("debug_counters", 1)
                    eng.set_option("dense_ablate", abl)
                    eng.dense_topk(q, k, device_out=True)
                    torch.cuda.synchronize()
                    c = eng.debug_counters().astype(np.float64)
                    eng.set_option("dense_ablate", 0)
                    eng.set_option("debug_counters", 0)
                    for g in (0, 1):
                        stages = c[g * 8 + 6]
                        res[f"dense B={B} var={var} pabl={abl} phase clocks per stage, group {g}"] = {
                            n_: round(v / stages, 1) for n_, v in zip(names, c[g * 8:g * 8 + 6])} if stages else {}
            eng.set_option("dense_var", 0)
        del x
    if what == "pp4":                                        # tiled-operand ping-pong scan (dense_pp = 4) against pp3, both bench shapes
        eng.set_option("dense_tiled", 1)
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        names = ["matrix", "wait", "barrier", "memory", "epilogue", "epi_barrier"]
        for B, k in ((256, 100), (1024, 288)):
            q = synth.dense_queries_torch(x, B, seed=7)
            ref = None
            for rep in "abc":
                for pp, tiled in ((3, 0), (3, 1), (4, 1)):
                    eng.set_option("dense_pp", pp)
                    eng.set_option("dense_tiled", tiled)
                    for abl in (0, 7):
                        eng.set_option("dense_ablate", abl)
                        res[f"dense B={B} pp={pp} tiled={tiled} pabl={abl} (run {rep})"] = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
                    eng.set_option("dense_ablate", 0)
                    ids, sc, ln = eng.dense_topk(q, k)
                    if ref is None:
                        ref = (ids.copy(), sc.copy(), ln.copy())
                    else:
                        same = bool(np.array_equal(ids, ref[0]) and np.array_equal(sc, ref[1]) and np.array_equal(ln, ref[2]))
                        res[f"dense B={B} pp={pp} tiled={tiled} identical to the first result (run {rep})"] = same
            eng.set_option("dense_pp", 4)
            eng.set_option("dense_tiled", 1)
            for abl in (21, 20):
                eng.set_option("debug_counters", 1)
                eng.set_option("dense_ablate", abl)
                eng.dense_topk(q, k, device_out=True)
                torch.cuda.synchronize()
                c = eng.debug_counters().astype(np.float64)
                eng.set_option("dense_ablate", 0)
                eng.set_option("debug_counters", 0)
                for g in (0, 1):
                    stages = c[g * 8 + 6]
                    res[f"dense B={B} pp=4 pabl={abl} phase clocks per stage, group {g}"] = {
                        n_: round(v / stages, 1) for n_, v in zip(names, c[g * 8:g * 8 + 6])} if stages else {}
        del x
    if what == "b256":                                       # configs[1] (256 queries, top-100): the scan's own ablation round
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        names = ["matrix", "wait", "barrier", "memory", "epilogue", "epi_barrier"]
        q = synth.dense_queries_torch(x, 256, seed=7)
        eng.set_option("dense_pp", 3)
        eng.set_option("dense_var", 0)
        for rep in "ab":
            for abl in (0, 7, 14, 13, 15, 12, 23, 16, 11, 17, 18):
                eng.set_option("dense_ablate", abl)
                res[f"dense B=256 pp=3 pabl={abl} (run {rep})"] = timed(eng, lambda: eng.dense_topk(q, 100, device_out=True))
        for abl in (21, 20, 22, 24):
            eng.set_option("debug_counters", 1)
            eng.set_option("dense_ablate", abl)
            eng.dense_topk(q, 100, device_out=True)
            torch.cuda.synchronize()
            c = eng.debug_counters().astype(np.float64)
            eng.set_option("dense_ablate", 0)
            eng.set_option("debug_counters", 0)
            for g in (0, 1):
                stages = c[g * 8 + 6]
                res[f"dense B=256 pp=3 pabl={abl} phase clocks per stage, group {g}"] = {
                    n_: round(v / stages, 1) for n_, v in zip(names, c[g * 8:g * 8 + 6])} if stages else {}
        eng.set_option("dense_ablate", 0)
        del x
    if what == "selfseed":                                   # sample pass inside the scan kernel (dense_selfseed) against store kernel + S0 + seed select
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        for B, k in ((1024, 288), (256, 100), (512, 100), (129, 100), (65, 10)):
            q = synth.dense_queries_torch(x, B, seed=7)
            for ss, n0 in ((1, 32768), (0, 32768), (1, 16384), (1, 32768), (0, 32768), (1, 16384)):
                eng.set_option("dense_selfseed", ss)
                eng.set_option("dense_n0", n0)
                r = timed(eng, lambda: eng.dense_topk(q, k, device_out=True), 20)
                r["sum"] = round(sum(r.values()), 4)
                res[f"dense B={B} k={k} selfseed={ss} n0={n0} #{len(res)}"] = r
        eng.set_option("dense_selfseed", 1)
        eng.set_option("dense_n0", 32768)
        del x
    if what == "tile384":                                    # 384 x 256 scan tile (dense_tile384) against the 256 x 256 one, >= 512 queries
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        for B, k in ((1024, 288), (512, 100), (768, 100)):
            q = synth.dense_queries_torch(x, B, seed=7)
            for t in (1, 0, 1, 0):
                eng.set_option("dense_tile384", t)
                r = timed(eng, lambda: eng.dense_topk(q, k, device_out=True), 20)
                r["sum"] = round(sum(r.values()), 4)
                res[f"dense B={B} k={k} tile384={t} #{len(res)}"] = r
        eng.set_option("dense_tile384", 1)
        q = synth.dense_queries_torch(x, 1024, seed=7)
        for rot in (0, 1, 2, 8, 0, 1, 2, 8):                       # K rotation per query tile of the 384 x 256 scan (default 0)
            eng.set_option("dense_rot", rot)
            r = timed(eng, lambda: eng.dense_topk(q, 288, device_out=True), 20)
            res[f"dense B=1024 k=288 tile384=1 rot={rot} #{len(res)}"] = r
        eng.set_option("dense_rot", 0)
        del x
    if what == "p3":                                         # strict-alternation ping-pong (dense_pp=3) vs the lean one (2)
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        names = ["matrix", "wait", "barrier", "memory", "epilogue", "epi_barrier"]
        for B, k in ((256, 100), (1024, 288)):
            q = synth.dense_queries_torch(x, B, seed=7)
            for rep in "ab":
                for pp, var in ((2, 0), (3, 0), (3, 1)):
                    eng.set_option("dense_pp", pp)
                    eng.set_option("dense_var", var)
                    for abl in (0, 7, 12, 23, 16):
                        eng.set_option("dense_ablate", abl)
                        res[f"dense B={B} pp={pp} var={var} pabl={abl} (run {rep})"] = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
            eng.set_option("dense_pp", 3)
            eng.set_option("dense_var", 1)
            for abl in (21, 20):
                eng.set_option("debug_counters", 1)
                eng.set_option("dense_ablate", abl)
                eng.dense_topk(q, k, device_out=True)
                torch.cuda.synchronize()
                c = eng.debug_counters().astype(np.float64)
                eng.set_option("dense_ablate", 0)
                eng.set_option("debug_counters", 0)
                for g in (0, 1):
                    stages = c[g * 8 + 6]
                    res[f"dense B={B} pp=3 pabl={abl} phase clocks per stage, group {g}"] = {
                        n_: round(v / stages, 1) for n_, v in zip(names, c[g * 8:g * 8 + 6])} if stages else {}
            eng.set_option("dense_ablate", 0)
            eng.set_option("dense_var", 0)
        eng.set_option("dense_pp", 2)
        del x
    if what == "bm25w":                                      # wave-owned scan vs block scan, section clocks
        indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
        for variant, name in ((BM25S, "bm25s"), (OKAPI, "okapi")):
            idx = build_bm25_index_from_postings(indptr, doc, tf, lens, variant, compute_payload=False)
            queries = synth.token_queries(flat, lens, vocab, 1024, seed=9)
            for ws in (1, 0, 1, 0):
                eng.set_option("bm25_wscan", ws)
                eng.set_bm25(idx, payload_on_device=True)
                for Bq, k in ((1024, 192), (256, 100), (16, 192), (1, 192)):
                    qi, qt = queries_to_csr(queries[:Bq])
                    res[f"{name} wscan={ws} B={Bq} k={k} #{len(res)}"] = timed(eng, lambda: eng.bm25_topk(qi, qt, k, device_out=True), 3)
            eng.set_option("bm25_wscan", 1)
            eng.set_bm25(idx, payload_on_device=True)
            qi, qt = queries_to_csr(queries)
            eng.set_option("debug_counters", 1)
            eng.bm25_topk(qi, qt, 192, device_out=True)
            torch.cuda.synchronize()
            c = eng.debug_counters().astype(np.float64)
            eng.set_option("debug_counters", 0)
            names = ["bounds+misc", "apply(+next fetch)", "sweep", "sweep_barrier", "shrink", "final"]
            tot = c[:6].sum()
            res[f"{name} wscan sections (% of thread-0 cycles, k=192)"] = {n_: round(100 * v / tot, 1) for n_, v in zip(names, c[:6])}
            res[f"{name} wscan cycles per query (thread 0)"] = {"total": round(tot / 1024)}
    if what == "bm25segs":                                   # segments per query of the fixed-point scan (packed shape), both bench batch sizes
        indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
        idx = build_bm25_index_from_postings(indptr, doc, tf, lens, BM25S, compute_payload=False)
        queries = synth.token_queries(flat, lens, vocab, 1024, seed=9)
        eng.set_bm25(idx, payload_on_device=True)
        for Bq, k, sweep in ((1024, 192, (0, 1, 2)), (256, 100, (0, 2, 3, 4, 6, 8)), (64, 100, (0, 4, 8, 16))):
            qi, qt = queries_to_csr(queries[:Bq])
            for rep in "ab":
                for sg in sweep:
                    eng.set_option("bm25_segs", sg)
                    r = timed(eng, lambda: eng.bm25_topk(qi, qt, k, device_out=True), 10)
                    print(f"bm25s B={Bq} k={k} segs={sg} (run {rep})  {json.dumps(r)}", flush=True)
        eng.set_option("bm25_segs", 0)
    if what == "bm25p16":                                    # packed shape: 4-byte postings (bm25_post16) against the 8-byte ones
        indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
        for variant, name in ((BM25S, "bm25s"), (OKAPI, "okapi")):
            idx = build_bm25_index_from_postings(indptr, doc, tf, lens, variant, compute_payload=False)
            queries = synth.token_queries(flat, lens, vocab, 1024, seed=9)
            eng.set_bm25(idx, payload_on_device=True)
            for Bq, k in ((1024, 192), (256, 100), (64, 100)):
                qi, qt = queries_to_csr(queries[:Bq])
                for p16 in (1, 0, 1, 0):
                    eng.set_option("bm25_post16", p16)
                    r = timed(eng, lambda: eng.bm25_topk(qi, qt, k, device_out=True), 10)
                    print(f"{name} B={Bq} k={k} post16={p16}  {json.dumps(r)}", flush=True)
            eng.set_option("bm25_post16", 1)
    if what == "bm25a":                                      # fixed-point scan + exact re-score: times, ablations, section clocks
        class _Live(dict):
            def __setitem__(self, k_, v_):
                print(f"{k_:44s} {json.dumps(v_)}", flush=True)
                super().__setitem__(k_, v_)
        res = _Live()
        indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
        for variant, name in ((BM25S, "bm25s"), (OKAPI, "okapi")):
            idx = build_bm25_index_from_postings(indptr, doc, tf, lens, variant, compute_payload=False)
            queries = synth.token_queries(flat, lens, vocab, 1024, seed=9)
            eng.set_bm25(idx, payload_on_device=True)
            for rep in "ab":
                for small in (2, 1, 0):
                    eng.set_option("bm25_small", small)
                    for Bq, k in ((1024, 192), (256, 100), (16, 192), (1, 192)):
                        qi, qt = queries_to_csr(queries[:Bq])
                        res[f"{name} ascan small={small} B={Bq} k={k} (run {rep})"] = timed(eng, lambda: eng.bm25_topk(qi, qt, k, device_out=True), 3)
            eng.set_option("bm25_small", 2)
            qi, qt = queries_to_csr(queries)
            if name == "bm25s":
                for abl in (0, 1, 2, 3, 4, 8):    # 1 no adds, 4 no clear, 8 one descriptor set (cached loads, adds out of range)
                    eng.set_option("bm25_ablate", abl)
                    res[f"{name} ascan B=1024 k=192 ablate={abl}"] = timed(eng, lambda: eng.bm25_topk(qi, qt, 192, device_out=True), 3)
                eng.set_option("bm25_ablate", 0)
            eng.set_option("debug_counters", 1)
            eng.bm25_topk(qi, qt, 192, device_out=True)
            torch.cuda.synchronize()
            c = eng.debug_counters().astype(np.float64)
            eng.set_option("debug_counters", 0)
            names = ["describe", "publish+fill (+extra rounds)", "apply_barrier", "list+clear | sweep", "clear_barrier", "list shrinks", "load wait", "adds",
                     "final shrink", "re-score+rank+output"]
            tot = c[:10].sum()
            res[f"{name} ascan sections (% of thread-0 cycles, k=192)"] = {n_: round(100 * v / tot, 1) for n_, v in zip(names, c[:10])}
            res[f"{name} ascan cycles per query (thread 0)"] = {"total": round(tot / 1024)}
            fq = ["token table", "search set-up", "searches", "payloads", "sums", "rank+output"]
            res[f"{name} ascan tail (thread-0 cycles per query)"] = {n_: round(v / 1024) for n_, v in zip(fq, c[10:16])}
    if what == "bm25x":                                      # wave-owned scan: sweep / threshold crossings / free-running waves
        indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
        for variant, name in ((BM25S, "bm25s"), (OKAPI, "okapi")):
            idx = build_bm25_index_from_postings(indptr, doc, tf, lens, variant, compute_payload=False)
            queries = synth.token_queries(flat, lens, vocab, 1024, seed=9)
            eng.set_bm25(idx, payload_on_device=True)
            modes = ((0, 0), (2, 0))      # (bm25_crossing, reserved)
            for rep in "ab":
                for cx, fr in modes:
                    eng.set_option("bm25_crossing", cx)
                    for Bq, k in ((1024, 192), (256, 100)):
                        qi, qt = queries_to_csr(queries[:Bq])
                        res[f"{name} crossing={cx} free={fr} B={Bq} k={k} (run {rep})"] = timed(eng, lambda: eng.bm25_topk(qi, qt, k, device_out=True), 3)
            qi, qt = queries_to_csr(queries)
            for cx, fr in modes:
                eng.set_option("bm25_crossing", cx)
                eng.set_option("debug_counters", 1)
                eng.bm25_topk(qi, qt, 192, device_out=True)
                torch.cuda.synchronize()
                c = eng.debug_counters().astype(np.float64)
                eng.set_option("debug_counters", 0)
                names = ["bounds+misc", "apply(+next fetch)", "survivors", "barrier_wait", "shrink", "final"]
                tot = c[:6].sum()
                res[f"{name} crossing={cx} free={fr} sections (% of thread-0 cycles, k=192)"] = {n_: round(100 * v / tot, 1) for n_, v in zip(names, c[:6])}
                res[f"{name} crossing={cx} free={fr} per query (thread 0)"] = {"cycles": round(tot / 1024), "tiles_by_list": round(c[6] / 1024, 1), "crossings_wave0": round(c[7] / 1024, 1)}
            eng.set_option("bm25_crossing", 1)
    if what in ("all", "dense"):
        x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
        eng.set_dense(x)
        for B, k in ((256, 100), (1024, 288)):
            q = synth.dense_queries_torch(x, B, seed=7)
            for cfg, persist in ():
                eng.set_option("dense_cfg", cfg)
                eng.set_option("dense_persist", persist)
                res[f"dense B={B} k={k} cfg={cfg} persist={persist}"] = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
            eng.set_option("dense_cfg", 0)
            eng.set_option("dense_persist", 1)
            for rep in "ab":
                for cfg, ra, abl in ((0, 0, 0), (0, 0, 10), (2, 0, 0), (2, 1, 0), (2, 1, 7), (0, 0, 7)):
                    eng.set_option("dense_cfg", cfg)
                    eng.set_option("dense_readahead", ra)
                    eng.set_option("dense_ablate", abl)
                    res[f"dense B={B} k={k} persist cfg={cfg} ra={ra} pabl={abl} (run {rep})"] = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
            eng.set_option("dense_ablate", 0)
            eng.set_option("dense_cfg", 0)
            eng.set_option("dense_readahead", 1)
            eng.set_option("dense_persist", 0)
            for cfg in ():
                eng.set_option("dense_cfg", cfg)
                for abl in (0, 1):
                    eng.set_option("dense_ablate", abl)
                    res[f"dense B={B} k={k} cfg={cfg} ablate={abl}"] = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
                eng.set_option("dense_ablate", 0)
            eng.set_option("dense_cfg", 0)
            eng.set_option("debug_counters", 1)
            eng.set_option("dense_ablate", 5)
            eng.dense_topk(q, k, device_out=True)
            c = eng.debug_counters().astype(np.float64)[8:14]
            eng.set_option("dense_ablate", 0)
            eng.set_option("debug_counters", 0)
            names = ["wait+barrier", "dma_issue", "lds_read+mfma_issue", "epilogue", "prologue", "total"]
            res[f"dense B={B} cfg=0 sections (% of wave-0 clocks)"] = {n_: round(100 * v / c[5], 1) for n_, v in zip(names, c)}
            for n0, n1 in ((32768, 131072), (32768, 229376)):
                eng.set_option("dense_n0", n0)
                eng.set_option("dense_n1", n1)
                res[f"dense B={B} k={k} n0={n0} n1={n1} #{len(res)}"] = timed(eng, lambda: eng.dense_topk(q, k, device_out=True))
            eng.set_option("dense_n0", 32768)
            eng.set_option("dense_n1", 131072)
            eng.set_option("dense_persist", 1)
        del x
    if what in ("all", "bm25"):
        indptr, doc, tf, lens, flat = synth.token_csr_torch(n, vocab, seed=3, device=dev)
        for variant, name in ((BM25S, "bm25s"), (OKAPI, "okapi")):
            idx = build_bm25_index_from_postings(indptr, doc, tf, lens, variant, compute_payload=False)
            eng.set_bm25(idx, payload_on_device=True)
            queries = synth.token_queries(flat, lens, vocab, 1024, seed=9)
            qi, qt = queries_to_csr(queries)
            for k in (100, 192):
                for abl in (0, 0, 0):
                    eng.set_option("bm25_ablate", abl)
                    res[f"{name} B=1024 k={k} ablate={abl} #{len(res)}"] = timed(eng, lambda: eng.bm25_topk(qi, qt, k, device_out=True), 3)
                eng.set_option("bm25_ablate", 0)
            eng.set_option("debug_counters", 1)
            eng.bm25_topk(qi, qt, 192, device_out=True)
            c = eng.debug_counters().astype(np.float64)
            eng.set_option("debug_counters", 0)
            names = ["ranges", "token_loop", "shrink_in_sweep", "sweep_own", "sweep_barriers", "shrink", "final", "tile_misc"]
            tot = c[:8].sum()
            res[f"{name} sections (% of thread-0 cycles, k=192)"] = {n_: round(100 * v / tot, 1) for n_, v in zip(names, c[:8])}
            res[f"{name} cycles per query (thread 0)"] = {"total": round(tot / 1024)}
            for Bq in (256, 16):
                qi, qt = queries_to_csr(queries[:Bq])
                res[f"{name} B={Bq} k=192"] = timed(eng, lambda: eng.bm25_topk(qi, qt, 192, device_out=True), 3)
    for k_, v in res.items():
        print(f"{k_:44s} {json.dumps(v)}")
    eng.close()


if __name__ == "__main__":
    main()
