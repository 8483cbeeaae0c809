#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/build.log 2>&1
import __graft_entry__ as g
g.build()
PY
tail -1 gpurun_out/build.log
timeout 1700 python -m pytest tests/test_gpu_sparse_fusion.py tests/test_gpu_fullsize.py -m gpu -q -s --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "device index build|passed|failed|Error|error" gpurun_out/pytest_gpu.log | tail -12
python - <<'PY'
import sys, time
sys.path.insert(0, ".")
import torch, numpy as np
from easyrag_amd import synth
from easyrag_amd._lib import ERH_K_DENSE_SCAN, ERH_K_DENSE_SELECT
from easyrag_amd.engine import RetrievalEngine
dev = torch.device("cuda", 0)
eng = RetrievalEngine(0)
x = synth.dense_corpus_torch(1_000_000, 1024, seed=2, device=dev)
eng.set_dense(x)
for B, k in ((1, 100), (8, 100), (16, 288)):
    q = synth.dense_queries_torch(x, B, seed=7)
    for kb in (32, 16):
        for wgs in (2, 3, 4):
            for single in (1, 0):
                eng.set_option("dense_gemv_kb", kb); eng.set_option("dense_gemv_wgs", wgs); eng.set_option("dense_small_single_stage", single)
                for _ in range(3): eng.dense_topk(q, k, device_out=True)
                torch.cuda.synchronize()
                eng.set_profiling(True); eng.reset_kernel_time()
                t0 = time.perf_counter()
                for _ in range(20): eng.dense_topk(q, k, device_out=True)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 20 * 1e3
                eng.set_profiling(False)
                sc = eng.kernel_time(ERH_K_DENSE_SCAN); se = eng.kernel_time(ERH_K_DENSE_SELECT)
                print(f"B={B} k={k} kb={kb} wgs={wgs} single={single}: wall {dt:.3f} ms  scan {sc['ms']/20:.3f}  select {se['ms']/20:.3f}")
PY
