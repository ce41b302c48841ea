#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout -k 10 300 python3 bench.py --only-step-api --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
for k, v in d['step_api'].items():
    print(k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items() if a in ('us_per_call', 'gpu_us_per_call', 'env_steps_per_s', 'error', 'plies_per_s', 'frac_of_copy')})
"
