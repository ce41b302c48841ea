#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout -k 10 120 python3 tools/debug/sync_latency.py 0 2>&1 | grep -v amdgpu.ids
