#!/usr/bin/env python3
"""Phase timeline of ONE launch of the lane-per-player byte kernel (diagnostic build: tools/lib_variant.sh stamps tron
-DCRL_QUAD_STAMPS [-DCRL_QUAD_SKEW=n]; run with CRL_LIB_PATH=build/ab_stamps/libcolosseum_hip.so):
when, relative to the first wave's entry, the waves enter, have their boards in LDS, finish stepping and end.
    python tools/debug/quad_phases.py [steps [board width [kernel]]]
(board widths 21..40 with kernel "qbits": the same four stamps in the lane-per-player bitboard kernel -- entry, bits laid out,
steps done, end of the replay)"""
import ctypes as C
import os
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from colosseumrl_amd import _native  # noqa: E402
from colosseumrl_amd.batched import TronBatch  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20           # board width
KERNEL = sys.argv[3] if len(sys.argv) > 3 else ("quad" if N <= 20 else "qbits")
lib = _native.lib()
tb = TronBatch(N, 4, 65536)
for _ in range(5):
    tb.rollout(T, 0, kernel=KERNEL)
torch.cuda.synchronize()
buf = (C.c_uint64 * (4096 * 4))()
for trial in range(3):
    tb.rollout(T, 0, kernel=KERNEL)
    torch.cuda.synchronize()
    assert lib.crl_diag_quad_stamps(buf, 4096 * 4) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4).astype(np.int64)
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0                       # 100 MHz wall clock
    names = ["entry", "boards in LDS", "steps done", "end"]
    print("T=%d trial %d: kernel span %.2f us" % (T, trial, us[:, 3].max()))
    for i, n in enumerate(names):
        c = us[:, i]
        print("  %-14s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (n, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
    d = us[:, 1:] - us[:, :-1]
    print("  per wave: copy-in %.2f  stepping %.2f  copy-out %.2f (medians)" % tuple(np.median(d, axis=0)))
    for grp, sel in (("even waves", slice(0, None, 2)), ("odd waves", slice(1, None, 2))):
        print("  %-10s entry %.2f  LDS %.2f  stepped %.2f  end %.2f (medians)" % ((grp,) + tuple(np.median(us[sel], axis=0))))
