#!/usr/bin/env python3
"""Throughput of a workload's rollout kernel against the batch size (= resident waves per SIMD): is the kernel bound by
instruction issue (throughput flat beyond the workload's own batch) or by the latency of its dependent chains (throughput
rises with more waves per SIMD)?
    python tools/debug/batch_sweep.py <workload> <steps per launch> <batch>[,<batch>...]"""
import os
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

wl, T = sys.argv[1], int(sys.argv[2])
game, kw = bench.WORKLOADS[wl][:2]
for B in [int(x) for x in sys.argv[3].split(",")]:
    st = bench.make_stepper(game, kw, B, torch.device("cuda", 0), 0)
    for _ in range(3):
        st.rollout(T, 0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            st.rollout(T, 0)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ts.sort()
    print("%s T=%d B=%-8d %8.4f ms per launch  %.4g env-steps/s" % (wl, T, B, ts[2], B * T / (ts[2] * 1e-3)), flush=True)
    del st
