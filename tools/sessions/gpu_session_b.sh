#!/bin/bash
# Round-2 GPU session B: tests, driver-style bench, host-wait-mode experiment, per-workload profiles (trace + PMC).
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; date
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
echo "== driver-style bench"; date
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err; echo "bench rc=$?"; head -c 1500 gpurun_out/bench_driver.json; echo
echo "== headline only x3 (default host wait mode)"
for i in 1 2 3; do timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['timed_region_ms'], d['kernel_ms'])"; done
echo "== headline only x3 (HSA_ENABLE_INTERRUPT=0)"
for i in 1 2 3; do HSA_ENABLE_INTERRUPT=0 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['timed_region_ms'], d['kernel_ms'])"; done
echo "== profiles"; date
timeout -k 10 400 bash tools/profile_bench.sh r2_tron_n20_t20 --steps 20 --warmup 20 > gpurun_out/prof_r2_tron_n20_t20.log 2>&1; echo "t20 rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_tron_n20 --steps 16384 --warmup 8192 > gpurun_out/prof_r2_tron_n20.log 2>&1; echo "n20 rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_step_api --only-step-api > gpurun_out/prof_r2_step_api.log 2>&1; echo "step_api rc=$?"
date
