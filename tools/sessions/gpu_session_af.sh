#!/bin/bash
# nth_set_bit by bit-field extracts (TicTacToe), popcount accumulation inside v_bcnt (Blokus count pass): parity + workloads
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_ttt.py tests/test_gpu_blokus.py tests/test_gpu_soak.py tests/test_gpu_abi_properties.py -m gpu -x -q -k "not tron" > gpurun_out/pytest_tb.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_tb.log
[ $rc -ne 0 ] && exit 1
for w in ttt_p3_5x5_k4_b262144 ttt_p3_3x5_k3_b262144 ttt_p4_3x3x3_b262144 blokus_p4_b16384; do
  timeout -k 10 300 python3 bench.py --workload $w --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '%.4g'%d['value'])"
done
