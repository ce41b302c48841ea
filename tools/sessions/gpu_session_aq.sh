#!/bin/bash
# Blokus profile after the last count-pass changes
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 500 bash tools/profile_bench.sh r2_blokus --workload blokus_p4_b16384 --steps 4096 --warmup 2048 > gpurun_out/prof_r2_blokus.log 2>&1; echo "blokus rc=$?"
grep "rollout" gpurun_out/prof_r2_blokus/summary.txt | head -6 | cut -c1-400
find gpurun_out -name "*_kernel_trace.csv" -delete; find gpurun_out -name "*_counter_collection.csv" -delete
