#!/bin/bash
# Blokus rollout: other players' rows derived lazily (before the move is placed): parity, fuzz, workload
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_blokus.py tests/test_gpu_soak.py tests/test_gpu_abi_properties.py -m gpu -x -q -k "not tron and not ttt" > gpurun_out/pytest_blk.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_blk.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python3 tools/debug/ttt_blokus_fuzz.py 0 60 555 2>&1 | grep -v amdgpu.ids | tail -2
for i in 1 2; do timeout -k 10 300 python3 bench.py --workload blokus_p4_b16384 --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '%.4g'%d['value'])"; done
