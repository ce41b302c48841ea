#!/bin/bash
# GPU box (via gpurun): what the driver does at round end -- full GPU suite, smoke(), bench.py the way the driver runs it
# (one process, --steps 20 --warmup 5), the same under torch.distributed.run with ONE rank, and the default bench.
#   gpurun --timeout 1200 -- 'bash tools/gpu_roundend.sh'
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -1 gpurun_out/smoke.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err; rc=$?; echo "bench rc=$rc lines=$(wc -l < gpurun_out/bench_driver.json) bytes=$(wc -c < gpurun_out/bench_driver.json)"; cat gpurun_out/bench_driver.json; echo
[ $rc -ne 0 ] && exit 1
cp gpurun_out/bench_detail.json gpurun_out/bench_detail_driver.json      # (the default bench below writes bench_detail.json again)
timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 20 --warmup 5 --only-headline --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; rc=$?; echo "torchrun rc=$rc"; head -c 900 gpurun_out/bench_torchrun1.json; echo
[ $rc -ne 0 ] && exit 1
if [ "${1:-}" = "default" ]; then
  timeout -k 10 900 python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default bench rc=$?"; head -c 300 gpurun_out/bench_default.json; echo
fi
