#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 bash tools/profile_bench.sh r2_step_api --only-step-api > gpurun_out/prof_r2_step_api.log 2>&1; echo "step_api rc=$?"
grep -E "n= " gpurun_out/prof_r2_step_api/summary.txt | grep -E "ttt_|blokus_|tron_" | cut -c1-170
