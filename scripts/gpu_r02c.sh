#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/build.log 2>&1
import __graft_entry__ as g
g.build()
PY
tail -2 gpurun_out/build.log
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=8 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -30 gpurun_out/pytest_gpu.log
for wl in hybrid dense bm25; do
  timeout 600 python bench.py --workload $wl --steps 10 --warmup 2 > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err; echo "bench $wl exit $?"
  tail -c 2600 gpurun_out/bench_$wl.json; tail -3 gpurun_out/bench_$wl.err
done
timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/bench_g2.json 2> gpurun_out/bench_g2.err; echo "bench --gpus 2 exit $? (expected 2 on a 1-GPU box)"; tail -2 gpurun_out/bench_g2.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 1 --cpu-queries 0 > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; echo "torchrun x1 exit $?"; tail -c 600 gpurun_out/bench_torchrun1.json
