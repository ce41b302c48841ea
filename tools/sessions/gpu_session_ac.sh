#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for f in 0 1 2 4; do timeout -k 10 120 python3 tools/debug/sync_latency.py $f 2>&1 | grep -v amdgpu.ids; done
