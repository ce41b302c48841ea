#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_tron.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
for i in 1 2 3; do
timeout -k 10 300 python3 bench.py --workload tron_p4_n40_b65536 --steps 32768 --warmup 8192 --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('n40', '%.4g'%d['value'], d['roofline']['launch_ms'])"
done
timeout -k 10 300 bash tools/profile_bench.sh r2_tron_n40 --workload tron_p4_n40_b65536 --steps 16384 --warmup 8192 > gpurun_out/prof_r2_tron_n40.log 2>&1; echo "n40 rc=$?"
grep "rollout" gpurun_out/prof_r2_tron_n40/summary.txt | cut -c1-700
timeout -k 10 300 python3 tools/kernel_ab.py 20 8192
timeout -k 10 300 python3 tools/kernel_ab.py 20 256
timeout -k 10 300 python3 tools/kernel_ab.py 40 8192
