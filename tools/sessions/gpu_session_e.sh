#!/bin/bash
# Round-2 GPU session E: refresh the Tron 40x40 profile (double-slab bitboard kernel) and the driver-style record.
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 bash tools/profile_bench.sh r2_tron_n40 --workload tron_p4_n40_b65536 --steps 16384 --warmup 8192 > gpurun_out/prof_r2_tron_n40.log 2>&1; echo "n40 rc=$?"
grep "rollout" gpurun_out/prof_r2_tron_n40/summary.txt | cut -c1-700
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err; echo "bench rc=$?"; head -c 600 gpurun_out/bench_driver.json; echo
