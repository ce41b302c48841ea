#!/bin/bash
# first runs of the lane-per-player bitboard kernel (qbits): Tron parity tests, then the A/B at 40x40 and 20x20
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_tron.py tests/test_gpu_soak.py tests/test_gpu_abi_properties.py -m gpu -x -q > gpurun_out/pytest_tron.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 gpurun_out/pytest_tron.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 tools/kernel_ab.py 40 8192 && timeout -k 10 300 python3 tools/kernel_ab.py 20 8192 && timeout -k 10 300 python3 tools/kernel_ab.py 40 256
