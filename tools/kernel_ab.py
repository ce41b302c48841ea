#!/usr/bin/env python3
"""A/B of the interchangeable Tron rollout kernels on one device (interleaved rounds in one process):
python3 tools/kernel_ab.py [N] [steps per launch]"""
import os
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from colosseumrl_amd.batched import TronBatch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
B = 65536
tbs = {k: TronBatch(N, 4, B) for k in (("quad", "bytes", "bits", "qbits") if N <= 20 else ("bytes", "bits", "qbits"))}
for k, tb in tbs.items():
    tb.rollout(T, 0, kernel=k)
torch.cuda.synchronize()
res = {k: [] for k in tbs}
for rnd in range(5):
    for k, tb in tbs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            tb.rollout(T, 0, kernel=k)
        e1.record()
        torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 3)
for k, v in res.items():
    v = sorted(v)
    print("N=%d T=%d %-5s median %.3f ms  min %.3f ms  -> %.4g env-steps/s" % (N, T, k, v[len(v) // 2], v[0], B * T / (v[len(v) // 2] * 1e-3)))
for k in tbs:
    assert torch.equal(tbs["bytes"].board, tbs[k].board) and torch.equal(tbs["bytes"].ret_sum, tbs[k].ret_sum), k
