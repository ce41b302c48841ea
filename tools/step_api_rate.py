#!/usr/bin/env python3
"""Throughput of the per-step batched API (external actions, all outputs written every step): eager launches and a
captured HIP graph of 64 steps.  Run on the GPU box: python tools/step_api_rate.py"""
import os
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from colosseumrl_amd.batched import BlokusBatch, TronBatch, TTTBatch


def rate(fn, steps, games):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return games * steps / (time.perf_counter() - t0)


def main():
    B = 65536
    tb = TronBatch(20, 4, B)
    acts = [torch.randint(-1, 2, (4, B), dtype=torch.int8, device="cuda") for _ in range(64)]
    i = [0]

    def eager():
        tb.step(acts[i[0] & 63], auto_reset=True)
        i[0] += 1
    print("tron 20x20 P4 B=65536 step(auto_reset), eager   : %.3g env-steps/s" % rate(eager, 2000, B))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(64):
            tb.step(acts[k], auto_reset=True)
    print("tron 20x20 P4 B=65536 step(auto_reset), 64-step graph: %.3g env-steps/s" % rate(g.replay, 100, B * 64))
    pl = torch.zeros((B,), dtype=torch.int8, device="cuda")
    print("tron observe (state_to_observation)               : %.3g obs/s" % rate(lambda: tb.observe(pl), 500, B))
    buf = tb.observe_all()
    rr = rate(lambda: tb.observe_all(buf), 500, B)
    print("tron observe_all (4 observers, fused)             : %.3g games/s = %.2f TB/s (N*N in + 4*N*N out)" % (rr, rr * 5 * 400 / 1e12))
    print("tron ranking (compute_ranking)                    : %.3g games/s" % rate(lambda: tb.ranking(), 500, B))
    tt = TTTBatch((3, 5), 3, 3, 262144)
    a = torch.randint(0, 15, (262144,), dtype=torch.int8, device="cuda")
    print("ttt 3x5 P3 B=262144 step(auto_reset), eager      : %.3g env-steps/s" % rate(lambda: tt.step(a, auto_reset=True), 2000, 262144))
    bb = BlokusBatch(16384)
    cnt = bb.valid()
    print("blokus B=16384 valid (count only)                 : %.3g games/s" % rate(lambda: bb.valid(), 200, 16384))
    passes = torch.full((16384,), -1, dtype=torch.int32, device="cuda")
    print("blokus B=16384 step (pass actions)                : %.3g env-steps/s" % rate(lambda: bb.step(passes), 200, 16384))
    print("blokus observe                                    : %.3g obs/s" % rate(lambda: bb.observe(torch.zeros((16384,), dtype=torch.int8, device='cuda')), 200, 16384))


if __name__ == "__main__":
    main()
