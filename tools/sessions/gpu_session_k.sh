#!/bin/bash
# Refresh the per-step API profile (fused calls of all three games, coalesced ranking kernel) and the driver-style record.
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 bash tools/profile_bench.sh r2_step_api --only-step-api > gpurun_out/prof_r2_step_api.log 2>&1; echo "step_api rc=$?"
grep -E "n= " gpurun_out/prof_r2_step_api/summary.txt | cut -c1-170 | head -24
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err; echo "bench rc=$?"; head -c 300 gpurun_out/bench_driver.json; echo
