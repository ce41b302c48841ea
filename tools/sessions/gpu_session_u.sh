#!/bin/bash
# Philox products as v_mad_u64_u32 (all games): full GPU suite, Tron A/B, one steady-state line per other workload
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 tools/kernel_ab.py 20 8192
for w in tron_p4_n40_b65536 ttt_p3_5x5_k4_b262144 ttt_p3_3x5_k3_b262144 ttt_p4_3x3x3_b262144 blokus_p4_b16384; do
  timeout -k 10 300 python3 bench.py --workload $w --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '%.4g'%d['value'], 'steady %.4g' % d.get('steady_state',{}).get('value',0))"
done
