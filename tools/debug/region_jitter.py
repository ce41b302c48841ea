#!/usr/bin/env python3
"""Distribution of the contract's timed region (one 20-step launch of the headline workload between two synchronisations)
over many repetitions, with and without a short busy-wait between the opening synchronise and the clock's start."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
game, kw, batch, chunk = bench.WORKLOADS[bench.HEADLINE][:4]
st = bench.make_stepper(game, kw, batch, torch.device("cuda", 0), 0)
for _ in range(50):
    st.rollout(20, 0)
torch.cuda.synchronize()

def run(spin_us, n=400, gap_ms=0.0):
    ts = []
    for _ in range(n):
        if gap_ms:
            time.sleep(gap_ms * 1e-3)
        st.rollout(5, 0)
        torch.cuda.synchronize()
        if spin_us:
            t = time.perf_counter()
            while time.perf_counter() - t < spin_us * 1e-6:
                pass
        t0 = time.perf_counter()
        st.rollout(20, 0)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    return ts[len(ts) // 2], ts[len(ts) // 10], ts[9 * len(ts) // 10], ts[-1]

for gap in (0.0, 2.0):
    for spin in (0, 20, 100):
        med, p10, p90, mx = run(spin, gap_ms=gap)
        print("gap %4.1f ms  spin %3d us: median %.1f  p10 %.1f  p90 %.1f  max %.1f us" % (gap, spin, med, p10, p90, mx), flush=True)
