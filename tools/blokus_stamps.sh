#!/bin/bash
# Diagnostic (GPU box): build blokus.hip with -DBLK_STAMPS, run a rollout, print per-phase cycle shares.
# The instrumented library is built into a scratch directory and loaded through CRL_LIB_PATH: the in-tree objects and
# the shipped libcolosseum_hip.so are never touched.
set -euo pipefail
export CRL_LIB_PATH=$(tools/diag_build.sh stamps -DBLK_STAMPS)
python3 - <<'PY'
import ctypes as C, torch
from colosseumrl_amd import _native
from colosseumrl_amd.batched import BlokusBatch
bb = BlokusBatch(16384)
bb.rollout(32, 1)
buf = (C.c_uint64 * 8)()
_native.lib().crl_blokus_stamps(buf, 1)
bb.rollout(256, 1)
torch.cuda.synchronize()
_native.lib().crl_blokus_stamps(buf, 1)
names = ["loop/outcome", "prep", "count", "rng+select", "apply", "exists", "rest", "-"]   # BLK_STAMP(k) closes segment k
tot = sum(buf)
for n, v in zip(names, buf):
    print("%-14s %6.1f %%   %8.0f cycles/wave-step" % (n, 100.0 * v / tot, v / 16384 / 256))
print("total cycles/wave-step %.0f" % (tot / 16384 / 256))
PY
