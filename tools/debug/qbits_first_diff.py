#!/usr/bin/env python3
"""First rollout step at which two Tron kernels part on a small board with many interactions (bring-up aid):
    python tools/debug/qbits_first_diff.py <kernel a> <kernel b> [N P B seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from colosseumrl_amd.batched import TronBatch
a_k, b_k = sys.argv[1], sys.argv[2]
N, P, B, seed = (int(x) for x in (sys.argv[3:7] + ["10", "3", "2048", "5"][len(sys.argv) - 3:]))
a, b = TronBatch(N, P, B), TronBatch(N, P, B)
for t in range(400):
    prev = {k: getattr(b, k).clone() for k in ("board", "heads", "dirs", "deaths", "tstep")}
    a.rollout(1, seed, kernel=a_k)
    b.rollout(1, seed, kernel=b_k)
    bad = None
    for k in ("heads", "dirs", "deaths", "n_episodes", "tstep", "board"):
        x, y = getattr(a, k), getattr(b, k)
        if not torch.equal(x, y):
            d = x != y
            g = int((d.reshape(d.shape[0], -1).any(1) if k in ("board", "n_episodes", "tstep") else d.any(0)).nonzero().flatten()[0])
            bad = (k, g)
            break
    if bad:
        k, g = bad
        print("step %d: %s differs first in game %d" % (t, k, g))
        print(" before: heads %s dirs %s deaths %s tstep %d" % (prev["heads"][:, g].tolist(), prev["dirs"][:, g].tolist(), prev["deaths"][:, g].tolist(), int(prev["tstep"][g])))
        for nm, o in ((a_k, a), (b_k, b)):
            print(" %-6s: heads %s dirs %s deaths %s tstep %d n_ep %d" % (nm, o.heads[:, g].tolist(), o.dirs[:, g].tolist(), o.deaths[:, g].tolist(), int(o.tstep[g]), int(o.n_episodes[g])))
        brd = prev["board"][g].reshape(N, N).tolist()
        for row in brd:
            print("   " + " ".join(str(c) for c in row))
        sys.exit(1)
print("no difference in 400 single-step launches")
