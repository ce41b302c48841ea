#!/bin/bash
# TicTacToe profiles after the running-game ply
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 bash tools/profile_bench.sh r2_ttt_5x5 --workload ttt_p3_5x5_k4_b262144 --steps 8192 --warmup 2048 > gpurun_out/prof_r2_ttt_5x5.log 2>&1; echo "ttt5 rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_ttt_3x5 --workload ttt_p3_3x5_k3_b262144 --steps 8192 --warmup 2048 > gpurun_out/prof_r2_ttt_3x5.log 2>&1; echo "ttt3 rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_ttt_3x3x3 --workload ttt_p4_3x3x3_b262144 --steps 8192 --warmup 2048 > gpurun_out/prof_r2_ttt_3x3x3.log 2>&1; echo "ttt333 rc=$?"
for f in ttt_5x5 ttt_3x5 ttt_3x3x3; do grep "rollout" gpurun_out/prof_r2_$f/summary.txt | head -1 | cut -c1-200; done
find gpurun_out -name "*_kernel_trace.csv" -delete; find gpurun_out -name "*_counter_collection.csv" -delete
