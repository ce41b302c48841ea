#!/bin/bash
# round-end style validation + fresh profiles of what changed since the last profile set (TicTacToe x3, quad 20-step shape, step API)
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout -k 10 400 bash tools/profile_bench.sh r2_ttt_5x5 --workload ttt_p3_5x5_k4_b262144 --steps 8192 --warmup 2048 > gpurun_out/prof_r2_ttt_5x5.log 2>&1; echo "ttt5 rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_ttt_3x5 --workload ttt_p3_3x5_k3_b262144 --steps 8192 --warmup 2048 > gpurun_out/prof_r2_ttt_3x5.log 2>&1; echo "ttt3 rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_ttt_3x3x3 --workload ttt_p4_3x3x3_b262144 --steps 8192 --warmup 2048 > gpurun_out/prof_r2_ttt_3x3x3.log 2>&1; echo "ttt333 rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_tron_n20_t20 --steps 20 --warmup 20 > gpurun_out/prof_r2_tron_n20_t20.log 2>&1; echo "t20 rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_tron_n20 --steps 16384 --warmup 8192 > gpurun_out/prof_r2_tron_n20.log 2>&1; echo "n20 rc=$?"
for f in ttt_5x5 ttt_3x5 ttt_3x3x3 tron_n20_t20 tron_n20; do grep "rollout" gpurun_out/prof_r2_$f/summary.txt | head -1 | cut -c1-200; done
