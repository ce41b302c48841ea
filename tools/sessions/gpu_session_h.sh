#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_blokus.py tests/test_gpu_abi_properties.py tests/test_gpu_rccl.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
for rep in 1 2 3; do
timeout -k 10 300 python3 bench.py --workload blokus_p4_b16384 --steps 4096 --warmup 2048 --only-headline --no-cpu-baseline 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('blokus', round(d['value']/1e6,1), 'M env-steps/s')"
done
bash tools/blokus_stamps.sh 2>&1 | tail -10
