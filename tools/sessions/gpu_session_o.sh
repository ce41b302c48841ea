#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 bench.py --only-step-api 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read())
print(d['copy_peak_GBs'], d['write_stream_peak_GBs'])
for k,v in d['step_api'].items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in v.items() if a!='what'})"
