#!/bin/bash
# quad kernel as the default for boards <= 20x20, P <= 4: full suite, A/B, fresh profiles of both launch shapes
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 tools/kernel_ab.py 20 8192
timeout -k 10 300 python3 tools/kernel_ab.py 20 20
timeout -k 10 300 bash tools/profile_bench.sh r2_tron_n20_t20 --steps 20 --warmup 20 > gpurun_out/prof_r2_tron_n20_t20.log 2>&1; echo "t20 rc=$?"
timeout -k 10 300 bash tools/profile_bench.sh r2_tron_n20 --steps 16384 --warmup 8192 > gpurun_out/prof_r2_tron_n20.log 2>&1; echo "n20 rc=$?"
grep "rollout_quad" gpurun_out/prof_r2_tron_n20/summary.txt | cut -c1-500
for i in 1 2 3; do timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g'%d['value'], round(d['timed_region_ms']*1e3,1), round(d['kernel_ms']*1e3,1))"; done
