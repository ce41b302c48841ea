#!/bin/bash
# rocprof + PMC passes for the two lane-per-player kernels (steady state), and the 20-step launch shape
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 bash tools/profile_bench.sh r2_tron_n40 --workload tron_p4_n40_b65536 --steps 16384 --warmup 8192 > gpurun_out/prof_r2_tron_n40.log 2>&1; echo "n40 rc=$?"
grep "rollout" gpurun_out/prof_r2_tron_n40/summary.txt | cut -c1-900
timeout -k 10 400 bash tools/profile_bench.sh r2_tron_n20 --steps 16384 --warmup 8192 > gpurun_out/prof_r2_tron_n20.log 2>&1; echo "n20 rc=$?"
grep "rollout" gpurun_out/prof_r2_tron_n20/summary.txt | cut -c1-900
timeout -k 10 400 bash tools/profile_bench.sh r2_tron_n20_t20 --steps 20 --warmup 20 > gpurun_out/prof_r2_tron_n20_t20.log 2>&1; echo "t20 rc=$?"
grep "rollout" gpurun_out/prof_r2_tron_n20_t20/summary.txt | cut -c1-900
