#!/bin/bash
# row-wise board copy of the quad kernel at 20x20 + unconditional state loads: parity, then the 20-step launch and the A/B
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_tron.py tests/test_gpu_soak.py tests/test_gpu_abi_properties.py -m gpu -x -q > gpurun_out/pytest_tron.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_tron.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 tools/kernel_ab.py 20 20 && timeout -k 10 300 python3 tools/kernel_ab.py 20 8192 && timeout -k 10 300 python3 tools/kernel_ab.py 40 8192
for i in 1 2 3; do timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g'%d['value'], round(d['timed_region_ms']*1e3,1), round(d['kernel_ms']*1e3,1))"; done
