#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/debug/qbits_diff.py qbits bytes 2>&1 | tee gpurun_out/qbits_diff.log
