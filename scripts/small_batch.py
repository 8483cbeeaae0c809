#!/usr/bin/env python
"""Latency of the small-batch dense path (skinny-GEMM stream) at 1M x 1024: B = 1, 4, 16 and, for comparison, 17 (the
padded 256-query scan).  Library event timers per kernel class + wall time per call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from easyrag_amd import synth  # noqa: E402
from easyrag_amd._lib import ERH_K_DENSE_SCAN, ERH_K_DENSE_SELECT  # noqa: E402
from easyrag_amd.engine import RetrievalEngine  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n, d = 1_000_000, 1024
    eng = RetrievalEngine(0)
    x = synth.dense_corpus_torch(n, d, seed=2, device=dev)
    eng.set_dense(x)
    for B, k in ((1, 288), (1, 10), (4, 288), (16, 288), (17, 288)):
        q = synth.dense_queries_torch(x, B, seed=7)
        for _ in range(3):
            eng.dense_topk(q, k, device_out=True)
        torch.cuda.synchronize()
        eng.set_profiling(True)
        eng.reset_kernel_time()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.dense_topk(q, k, device_out=True)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        eng.set_profiling(False)
        scan = eng.kernel_time(ERH_K_DENSE_SCAN)["ms"] / reps
        sel = eng.kernel_time(ERH_K_DENSE_SELECT)["ms"] / reps
        print(f"B={B:3d} k={k:3d}: scan {scan:.3f} ms ({2.048 / scan:.2f} TB/s of the 2 GB matrix)  select {sel:.3f} ms  wall {wall:.3f} ms per call")
    eng.close()


if __name__ == "__main__":
    main()
