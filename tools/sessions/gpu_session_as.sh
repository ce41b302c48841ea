#!/bin/bash
# driver-style headline on a fresh box, several processes in a row (is the first one still slower?)
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for i in 1 2 3 4; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g'%d['value'], round(d['timed_region_ms']*1e3,1), round(d['kernel_ms']*1e3,1), d.get('device_warmup_ms'))"; done
