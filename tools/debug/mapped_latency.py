#!/usr/bin/env python3
"""Latency of ONE B = 1 Tron step through the C ABI, three ways (bring-up aid for the drop-in path):
 (a) state in device tensors, blocking torch copies each way (round 2's drop-in path),
 (b) state in crl_host_alloc memory, kernels reading / writing it over PCIe, crl_stream_synchronize,
 (c) as (b) but spinning on a byte the kernel's last store... (not possible without a kernel change: skipped)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from colosseumrl_amd import _native  # noqa: E402
from colosseumrl_amd.envs.tron import layout  # noqa: E402

lib = _native.require_gpu()
N, P = 20, 4
heads, dirs = layout.start_positions(N, P, 1, [2] * P)
handle = C.c_void_p()
_native.check(lib.crl_tron_create(N, P, (C.c_int16 * P)(*heads), (C.c_int8 * P)(*dirs), C.byref(handle)))
host, dev = C.c_void_p(), C.c_void_p()
nbytes = 4096
_native.check(lib.crl_host_alloc(nbytes, C.byref(host), C.byref(dev)))
print("host %x device %x" % (host.value, dev.value))
buf = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(host.value))
NN = N * N
off = {"board": 0, "heads": 512, "dirs": 512 + 16, "deaths": 512 + 32, "actions": 512 + 48, "rewards": 512 + 64,
       "terminal": 512 + 80, "winners": 512 + 96}
board = buf[0:NN].view(np.int8)
hv = buf[512:512 + 2 * P].view(np.int16)
dv = buf[528:528 + P].view(np.int8)
kv = buf[544:544 + P].view(np.int8)
av = buf[560:560 + P].view(np.int8)
rv = buf[576:576 + P].view(np.int8)
stream = C.c_void_p()
_native.check(lib.crl_stream_create(C.byref(stream)))
d = dev.value


def reset():
    board[:] = 0
    for p in range(P):
        board[heads[p]] = p + 1
    hv[:] = heads
    dv[:] = dirs
    kv[:] = 0


def step_mapped(sync=True):
    rc = lib.crl_tron_step(handle, 1, d + 0, d + 512, d + 528, d + 544, d + 560, d + 576, d + 592, d + 608, 0, stream)
    assert rc == 0, lib.crl_last_error()
    if sync:
        lib.crl_stream_synchronize(stream)


reset()
av[:] = [0, 1, -1, 0]
step_mapped()
print("after one step: heads", hv.tolist(), "dirs", dv.tolist(), "deaths", kv.tolist(), "rewards", rv.tolist())
ts = []
for i in range(2000):
    if i % 6 == 0:
        reset()
    t0 = time.perf_counter()
    step_mapped()
    ts.append(time.perf_counter() - t0)
ts.sort()
print("mapped host memory: launch + sync median %.1f us, p10 %.1f, min %.1f" % (ts[1000] * 1e6, ts[200] * 1e6, ts[0] * 1e6))
# null stream
stream0 = stream
stream = C.c_void_p(0)
ts = []
for i in range(2000):
    if i % 6 == 0:
        reset()
    t0 = time.perf_counter()
    step_mapped()
    ts.append(time.perf_counter() - t0)
ts.sort()
print("mapped, NULL stream: median %.1f us, min %.1f" % (ts[1000] * 1e6, ts[0] * 1e6))
stream = stream0
# device tensors + blocking copies (what round 2's drop-in classes did)
from colosseumrl_amd.batched import TronBatch  # noqa: E402
tb = TronBatch(N, P, 1)
act = torch.zeros((P, 1), dtype=torch.int8)
ts = []
for i in range(500):
    if i % 6 == 0:
        tb.reset()
    t0 = time.perf_counter()
    b = tb.board.cpu(); h = tb.heads.cpu(); dd = tb.dirs.cpu(); k = tb.deaths.cpu()
    tb.board.copy_(b); tb.heads.copy_(h); tb.dirs.copy_(dd); tb.deaths.copy_(k)
    a = act.to(tb.device)
    r, t, w = tb.step(a)
    b = tb.board.cpu(); h = tb.heads.cpu(); dd = tb.dirs.cpu(); k = tb.deaths.cpu(); r.cpu(); t.cpu(); w.cpu()
    ts.append(time.perf_counter() - t0)
ts.sort()
print("device tensors + copies: median %.1f us, min %.1f" % (ts[250] * 1e6, ts[0] * 1e6))
# two launches back to back then one sync (pipelining of async launches over mapped memory)
ts = []
for i in range(1000):
    reset()
    t0 = time.perf_counter()
    step_mapped(False)
    step_mapped(False)
    step_mapped(True)
    ts.append(time.perf_counter() - t0)
ts.sort()
print("mapped, three launches + one sync: median %.1f us, min %.1f" % (ts[500] * 1e6, ts[0] * 1e6))
