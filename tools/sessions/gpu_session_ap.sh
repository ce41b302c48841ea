#!/bin/bash
# Blokus: what does the serial anchor walk of the select pass cost?  (shipped vs a build that walks all anchors once more)
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
WALK=$(bash tools/diag_build.sh fullwalk -DBLK_DIAG_FULLWALK) || exit 1
for lib in shipped "$WALK" shipped "$WALK"; do
  [ "$lib" = shipped ] && unset CRL_LIB_PATH || export CRL_LIB_PATH=$lib
  timeout -k 10 300 python3 bench.py --workload blokus_p4_b16384 --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib'[-30:], '%.4g'%d['value'])"
done
bash tools/blokus_stamps.sh 2>&1 | grep -v amdgpu.ids | tail -9
