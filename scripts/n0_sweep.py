import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from easyrag_amd import synth
from easyrag_amd._lib import ERH_K_DENSE_SCAN, ERH_K_DENSE_SELECT
from easyrag_amd.engine import RetrievalEngine
dev = torch.device("cuda", 0)
n, d = 1_000_000, 1024
eng = RetrievalEngine(0)
x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
eng.set_dense(x)
for n0 in (32768, 16384, 8192, 4096):
    eng.set_option("dense_n0", n0)
    for B, k in ((1, 288), (1, 10), (16, 288), (64, 288), (256, 100), (1024, 288)):
        q = synth.dense_queries_torch(x, B, seed=7)
        for _ in range(3):
            eng.dense_topk(q, k, device_out=True)
        torch.cuda.synchronize()
        eng.set_profiling(True); eng.reset_kernel_time()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.dense_topk(q, k, device_out=True)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        eng.set_profiling(False)
        scan = eng.kernel_time(ERH_K_DENSE_SCAN)["ms"] / reps
        sel = eng.kernel_time(ERH_K_DENSE_SELECT)["ms"] / reps
        print(f"n0={n0:6d} B={B:4d} k={k:3d}: scan {scan:.3f} select {sel:.3f} wall {wall:.3f} ms  exhaustive {eng.dense_diag()['exhaustive']}")
