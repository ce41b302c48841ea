#!/bin/bash
# Diagnostic (GPU box): per-phase cycle shares of a Blokus rollout step from a -DBLK_STAMPS build.
# Build the variant in the build container first (it travels with the snapshot):  tools/lib_variant.sh blkstamps blokus -DBLK_STAMPS
set -euo pipefail
export CRL_LIB_PATH=${CRL_LIB_PATH:-build/ab_blkstamps/libcolosseum_hip.so}
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
names = ["loop/outcome", "prologue: rows, pre-shifted table, work list", "count: row range, scan of the piece counts", "rng+select", "apply", "exists", "count: shape x row batches", "-"]   # BLK_STAMP(k) closes segment k
tot = sum(buf)
for n, v in zip(names, buf):
    print("%-48s %6.1f %%   %8.0f cycles/wave-step" % (n, 100.0 * v / tot, v / 16384 / 256))
print("total cycles/wave-step %.0f" % (tot / 16384 / 256))
PY
