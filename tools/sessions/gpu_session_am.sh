#!/bin/bash
# randomised differential soak of the TicTacToe and Blokus rollouts against the oracle
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1000 python3 tools/debug/ttt_blokus_fuzz.py 1500 25 99 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ttt_blokus_fuzz.log | tail -25
