#!/bin/bash
# TicTacToe rollout: the ply of a running game (no legality checks): parity incl. finished-state entry, fuzz, workloads
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_ttt.py tests/test_gpu_soak.py tests/test_gpu_abi_properties.py -m gpu -x -q -k "not tron and not blokus" > gpurun_out/pytest_ttt.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_ttt.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python3 tools/debug/ttt_blokus_fuzz.py 1500 0 4242 2>&1 | grep -v amdgpu.ids | tail -2
for w in ttt_p3_5x5_k4_b262144 ttt_p3_3x5_k3_b262144 ttt_p4_3x3x3_b262144; do
  timeout -k 10 300 python3 bench.py --workload $w --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '%.4g'%d['value'])"
done
