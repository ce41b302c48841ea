#!/bin/bash
# long randomised soak of all rollout kernels (Tron, TicTacToe, Blokus) and the per-step API twins against the oracle
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 500 python3 tools/debug/tron_fuzz.py 9000 31415 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/fuzz_long.log
timeout -k 10 200 python3 tools/debug/ttt_blokus_fuzz.py 6000 60 27182 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a gpurun_out/fuzz_long.log
timeout -k 10 200 python3 tools/debug/step_api_fuzz.py 6000 16180 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a gpurun_out/fuzz_long.log
