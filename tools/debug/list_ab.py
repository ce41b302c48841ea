#!/usr/bin/env python3
"""A/B of blokus_list_kernel builds on the GPU box: one child process per library (CRL_LIB_PATH), each times valid_list on the
16,384 mid-game positions bench.py uses and checks the lists against the first library's.
usage: python tools/debug/list_ab.py <lib.so> [<lib.so> ...]"""
import os
import subprocess
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, hashlib
sys.path.insert(0, %r)
import torch
from colosseumrl_amd.batched import BlokusBatch
bb = BlokusBatch(16384)
bb.rollout(24, 5)
count, ids = bb.valid_list(2048)
for _ in range(5): bb.valid_list(2048, out=ids)
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): bb.valid_list(2048, out=ids)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
h = hashlib.sha1(ids.cpu().numpy().tobytes() + count.cpu().numpy().tobytes()).hexdigest()[:12]
print("%%-40s %%7.1f us (min %%.1f)  mean legal %%.0f  sha %%s" %% (sys.argv[1], sorted(ts)[2], min(ts), count.float().mean().item(), h))
""" % ROOT

for lib in sys.argv[1:]:
    env = dict(os.environ, CRL_LIB_PATH=os.path.abspath(lib))
    subprocess.run([sys.executable, "-c", CHILD, os.path.relpath(lib, ROOT)], env=env, check=False)
