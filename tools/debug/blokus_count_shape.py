#!/usr/bin/env python3
"""Diagnostic (GPU box): the shape of the Blokus count pass over random games, from a -DBLK_COUNTERS build
(tools/lib_variant.sh blkcnt blokus -DBLK_COUNTERS): work items, origin rows, row trips taken and the trips an even spread of
the (shape, row) pairs over 64 lanes would take."""
import ctypes as C
import os
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("CRL_LIB_PATH", "build/ab_blkcnt/libcolosseum_hip.so")
import torch
from colosseumrl_amd import _native
from colosseumrl_amd.batched import BlokusBatch
bb = BlokusBatch(16384)
bb.rollout(64, 1)
buf = (C.c_uint64 * 8)()
_native.lib().crl_blokus_stamps(buf, 1)
bb.rollout(512, 1)
torch.cuda.synchronize()
_native.lib().crl_blokus_stamps(buf, 1)
n = max(1, buf[0])
print("count passes %d (of %d plies)" % (buf[0], 16384 * 512))
print("work items per pass           %.1f" % (buf[1] / n))
print("origin rows per pass          %.2f   (anchor rows span %.2f)" % (buf[2] / n, buf[4] / n))
print("row trips per pass            %.2f" % (buf[3] / n))
print("... at an even spread         %.2f   (rows = anchor span + 3: %.2f)" % (buf[5] / n, buf[6] / n))
print("passes with <= 32 items       %.1f %%" % (100.0 * buf[7] / n))
