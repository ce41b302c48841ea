#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 1000 python3 tools/debug/step_api_fuzz.py 4000 77 2>&1 | grep -v amdgpu.ids | tee gpurun_out/step_api_fuzz.log | tail -25
