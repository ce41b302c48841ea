#!/bin/bash
# driver-style headline after a quiet spell (sleep), several processes: does the first timed region still run slow?
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for i in 1 2 3; do sleep 8; python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g'%d['value'], round(d['timed_region_ms']*1e3,1), round(d['kernel_ms']*1e3,1), d.get('device_warmup_ms'), ['%.3g'%v['value'] for k,v in d['seeds'].items() if k!='spread'])"; done
