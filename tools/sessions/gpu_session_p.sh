#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_obs/trace -- python3 bench.py --only-step-api > gpurun_out/prof_obs.log 2>&1; echo "rc=$?"
python3 tools/summarize_prof.py gpurun_out/prof_obs > gpurun_out/prof_obs/summary.txt 2>&1; grep -E "n= " gpurun_out/prof_obs/summary.txt | grep -E "blokus_|ttt_" | cut -c1-170
