#!/bin/bash
# randomised differential soak of all Tron rollout kernels against the oracle
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1000 python3 tools/debug/tron_fuzz.py 3000 20261004 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tron_fuzz.log | tail -20
