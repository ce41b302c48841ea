#!/bin/bash
# full GPU suite with qbits as the 40x40 default, its rocprof / PMC passes, then the default bench line
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 bash tools/profile_bench.sh r2_tron_n40 --workload tron_p4_n40_b65536 --steps 16384 --warmup 8192 > gpurun_out/prof_r2_tron_n40.log 2>&1; echo "n40 rc=$?"
grep "rollout" gpurun_out/prof_r2_tron_n40/summary.txt | cut -c1-400
timeout -k 10 900 python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default bench rc=$?"; head -c 600 gpurun_out/bench_default.json; echo
