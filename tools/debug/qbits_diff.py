#!/usr/bin/env python3
"""Field-by-field difference of two Tron rollout kernels on small batches (bring-up aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from colosseumrl_amd.batched import TronBatch

a_k, b_k = sys.argv[1], sys.argv[2]
for N, P, B, chunks in [(20, 4, 64, (64,)), (20, 4, 64, (5,)), (20, 4, 64, (1,)), (20, 4, 200, (64, 1, 31)), (40, 4, 100, (300,)), (12, 2, 70, (60,))]:
    a, b = TronBatch(N, P, B), TronBatch(N, P, B)
    for T in chunks:
        a.rollout(T, 7, kernel=a_k)
        b.rollout(T, 7, kernel=b_k)
    torch.cuda.synchronize()
    print("== N=%d P=%d B=%d chunks=%s" % (N, P, B, chunks))
    for k in ("board", "heads", "dirs", "deaths", "tcount", "tstep", "n_episodes", "win_count", "len_sum", "ret_sum", "last_winners", "last_len"):
        x, y = getattr(a, k), getattr(b, k)
        if not torch.equal(x, y):
            d = (x != y)
            games = d.reshape(d.shape[0], -1).any(1).nonzero().flatten() if k == "board" or d.dim() == 1 else d.any(0).nonzero().flatten()
            print("  %-12s differs: %d entries, games %s" % (k, int(d.sum()), games[:12].tolist()))
            if k == "board":
                e = int(games[0])
                idx = d[e].nonzero().flatten()[:10].tolist()
                print("    game %d cells %s  %s=%s  %s=%s  tstep=%d heads=%s/%s" % (e, idx, a_k, x[e, idx].tolist(), b_k, y[e, idx].tolist(), int(b.tstep[e]), a.heads[:, e].tolist(), b.heads[:, e].tolist()))
