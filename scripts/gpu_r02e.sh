#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/build.log 2>&1
import __graft_entry__ as g
g.build()
PY
tail -1 gpurun_out/build.log
timeout 1700 python -m pytest tests/test_gpu_dense.py -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -8 gpurun_out/pytest_gpu.log
cat > /tmp/b1.py <<'PY'
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from easyrag_amd import synth
from easyrag_amd.engine import RetrievalEngine
dev = torch.device("cuda", 0)
eng = RetrievalEngine(0)
x = synth.dense_corpus_torch(1_000_000, 1024, seed=2, device=dev)
eng.set_dense(x)
q = synth.dense_queries_torch(x, 1, seed=7)
for _ in range(10): eng.dense_topk(q, 100, device_out=True)
torch.cuda.synchronize()
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b1 -o b1 -- python /tmp/b1.py > $GRAFT_REPO_ROOT/gpurun_out/prof_b1.log 2>&1; echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT; for f in $(find gpurun_out/prof_b1 -name "*kernel_stats.csv"); do head -24 $f | cut -c1-200; done
