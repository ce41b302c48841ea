#!/bin/bash
# Round-2 GPU session D: tests + Tron 40x40 bitboard kernel A/B + step API rates.
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
for i in 1 2 3; do
timeout -k 10 300 python3 bench.py --workload tron_p4_n40_b65536 --steps 32768 --warmup 8192 --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('n40', '%.4g'%d['value'], d['roofline']['launch_ms'])"
done
timeout -k 10 300 python3 bench.py --workload tron_p4_n20_b65536 --steps 32768 --warmup 8192 --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('n20', '%.4g'%d['value'], d['roofline']['launch_ms'])"
timeout -k 10 300 python3 bench.py --only-step-api 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read())['step_api']
for k,v in d.items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if a!='what'})"
