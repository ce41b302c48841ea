#!/usr/bin/env python3
"""Host-side cost of the pieces of the contract's timed region (one 20-step Tron launch + gather + synchronise), warm:
the raw ctypes call, TronBatch.rollout, ShardedRollout.rollout, the gather without a process group, an idle synchronise,
and the whole region against ctypes call + synchronise alone (what the Python layers above the C ABI add: ~1.3 us).
    python tools/debug/host_latency.py"""
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
import os
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from colosseumrl_amd.batched import TronBatch, _stream, _DevGuard
from colosseumrl_amd.parallel import ShardedRollout
tb = TronBatch(20, 4, 65536)
sr = ShardedRollout(lambda batch, first_env_id: TronBatch(20, 4, batch, first_env_id=first_env_id), 65536)
st = sr.stepper
for _ in range(50):
    st.rollout(20, 0); torch.cuda.synchronize()
def t(fn, n=2000):
    fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
print("perf_counter pair           %.2f us" % t(lambda: time.perf_counter()))
print("_stream()                   %.2f us" % t(_stream))
g = _DevGuard(st.device)
def guard():
    with g: pass
print("_DevGuard enter/exit        %.2f us" % t(guard))
print("torch.cuda.synchronize idle %.2f us" % t(torch.cuda.synchronize))
args = st._rollout_args
lib = st._lib
s = _stream()
def raw():
    lib.crl_tron_rollout(st._ctx.handle, st.B, 0, st.first_env_id, 20, *args, 0, s)
# launch only (async) then sync outside the timing: measure launch call cost with queue draining every 20
def launch_cost(fn, n=400):
    tot = 0.0
    for i in range(n):
        t0 = time.perf_counter(); fn(); tot += time.perf_counter() - t0
        torch.cuda.synchronize()
    return tot / n * 1e6
print("ctypes crl_tron_rollout     %.2f us" % launch_cost(raw))
print("TronBatch.rollout           %.2f us" % launch_cost(lambda: st.rollout(20, 0)))
print("ShardedRollout.rollout      %.2f us" % launch_cost(lambda: sr.rollout(20, 0, 8192)))
print("sr.gather(copy=False)       %.2f us" % t(lambda: sr.gather(copy=False)))
def region():
    sr.rollout(20, 0, 8192); sr.gather(copy=False); torch.cuda.synchronize()
print("region (rollout+gather+sync) %.2f us" % t(region, 500))
def region_raw():
    raw(); torch.cuda.synchronize()
print("region raw (ctypes+sync)     %.2f us" % t(region_raw, 500))
