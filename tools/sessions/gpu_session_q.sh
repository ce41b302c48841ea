#!/bin/bash
# quad (lane-per-player) Tron kernel: parity first, then A/B against the lane-per-game kernels
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_tron.py -m gpu -x -q -k "quad" > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 tools/kernel_ab.py 20 8192
timeout -k 10 300 python3 tools/kernel_ab.py 20 256
timeout -k 10 300 python3 tools/kernel_ab.py 20 20
