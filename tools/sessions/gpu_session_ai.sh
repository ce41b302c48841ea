#!/bin/bash
# upper bound of what a cheaper join around the interaction path could buy: the lane-per-player kernels with that path compiled out
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
NOSLOW=$(bash tools/diag_build.sh noslow -DCRL_DIAG_NO_SLOW) || exit 1
DETECT=$(bash tools/diag_build.sh detect -DCRL_DIAG_DETECT_ONLY) || exit 1
cat > /tmp/t.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from colosseumrl_amd.batched import TronBatch
for N in (20, 40):
    tb = TronBatch(N, 4, 65536)
    tb.rollout(8192, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        tb.rollout(8192, 0)
    e1.record(); torch.cuda.synchronize()
    print(N, "%.3f ms" % (e0.elapsed_time(e1) / 5))
PY
for rep in 1; do
  echo "detection only"; CRL_LIB_PATH=$DETECT timeout -k 10 120 python3 /tmp/t.py
  echo "shipped"; timeout -k 10 120 python3 /tmp/t.py
  echo "no slow path"; CRL_LIB_PATH=$NOSLOW timeout -k 10 120 python3 /tmp/t.py
done
