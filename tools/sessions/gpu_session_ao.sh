#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -16 gpurun_out/pytest_gpu.log
