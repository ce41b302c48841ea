#!/usr/bin/env python3
"""Host-side latency of launch + synchronize for a 20-step headline launch, with the HIP schedule flag given as argv[1]
(0 auto, 1 spin, 2 yield, 4 blocking sync); bring-up aid for bench.py's timed region."""
import ctypes, os, sys, time
flag = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if flag:
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(ctypes.c_uint(flag)))
from colosseumrl_amd.batched import TronBatch
tb = TronBatch(20, 4, 65536)
for _ in range(5):
    tb.rollout(20, 0)
torch.cuda.synchronize()
ts = []
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tb.rollout(20, 0)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts.sort()
print("flag %d: launch+sync median %.1f us, p10 %.1f us, min %.1f us" % (flag, ts[100] * 1e6, ts[20] * 1e6, ts[0] * 1e6))
ts = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(); tb.rollout(20, 0); e1.record()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts.sort()
print("flag %d: with two event records: median %.1f us, min %.1f us; events say %.1f us" % (flag, ts[100] * 1e6, ts[0] * 1e6, e0.elapsed_time(e1) * 1e3))
# the same launch replayed from a one-node HIP graph
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        tb.rollout(20, 0)
torch.cuda.current_stream().wait_stream(side)
ts = []
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts.sort()
print("flag %d: graph replay + sync median %.1f us, min %.1f us" % (flag, ts[100] * 1e6, ts[0] * 1e6))
