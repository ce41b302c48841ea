#!/bin/bash
# profiles: Blokus rollout, TicTacToe rollouts (nth_set_bit change), the step API; then the default bench line
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 500 bash tools/profile_bench.sh r2_blokus --workload blokus_p4_b16384 --steps 4096 --warmup 2048 > gpurun_out/prof_r2_blokus.log 2>&1; echo "blokus rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_ttt_5x5 --workload ttt_p3_5x5_k4_b262144 --steps 8192 --warmup 2048 > gpurun_out/prof_r2_ttt_5x5.log 2>&1; echo "ttt5 rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_ttt_3x5 --workload ttt_p3_3x5_k3_b262144 --steps 8192 --warmup 2048 > gpurun_out/prof_r2_ttt_3x5.log 2>&1; echo "ttt3 rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_ttt_3x3x3 --workload ttt_p4_3x3x3_b262144 --steps 8192 --warmup 2048 > gpurun_out/prof_r2_ttt_3x3x3.log 2>&1; echo "ttt333 rc=$?"
timeout -k 10 500 bash tools/profile_bench.sh r2_step_api --only-step-api > gpurun_out/prof_r2_step_api.log 2>&1; echo "step api rc=$?"
for f in blokus ttt_5x5 ttt_3x5 ttt_3x3x3; do grep "rollout" gpurun_out/prof_r2_$f/summary.txt | head -1 | cut -c1-200; done
# raw traces are bulky (gpurun copies back at most 64 MiB): keep the summaries and per-kernel statistics only
find gpurun_out -name "*_kernel_trace.csv" -delete; find gpurun_out -name "*_counter_collection.csv" -delete; du -sh gpurun_out
