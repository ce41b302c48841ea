#!/usr/bin/env python3
"""Randomised differential run of every Tron rollout kernel against the CPU oracle (bring-up / soak aid, not a test):
random board sizes, player counts, ragged batches, split launches.  usage: tron_fuzz.py [n_cases] [seed]"""
import os, sys, time
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from colosseumrl_amd.batched import TronBatch
from oracle import oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
for case in range(n_cases):
    N = int(rng.integers(4, 41)) if rng.random() < 0.9 else int(rng.integers(41, 70))    # (above 40x40: the global-memory kernels)
    P = int(rng.integers(2, 9))
    while P * 2 > N * 2:                      # (start ring needs room)
        P -= 1
    B = int(rng.integers(1, 3000))
    chunks = [int(rng.integers(1, 700)) if rng.random() < 0.7 else int(rng.integers(1, 30)) for _ in range(int(rng.integers(1, 4)))]
    if rng.random() < 0.15:
        chunks.append(int(rng.integers(16384, 20000)))
        B = min(B, 300)
    seed, first = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 40))
    try:
        sh, sd = O.tron_start_positions(N, P)
    except Exception:
        continue
    ost = O.TronState(N, P, B)
    O.tron_reset(ost, sh, sd)
    for T in chunks:
        O.tron_rollout(ost, seed, first, T, sh, sd, n_threads=16)
    for kernel in ("auto", "quad", "pair", "qbits", "bytes", "bits", "global", "gquad"):
        tb = TronBatch(N, P, B, first_env_id=first)
        for T in chunks:
            tb.rollout(T, seed, kernel=kernel)
        torch.cuda.synchronize()
        for k in ("board", "heads", "dirs", "deaths", "tcount", "tstep", "n_episodes", "win_count", "len_sum", "ret_sum", "last_winners", "last_len"):
            want = getattr(ost, k)
            if not np.array_equal(getattr(tb, k).cpu().numpy().view(want.dtype), want):
                bad += 1
                print("MISMATCH case %d N=%d P=%d B=%d chunks=%s kernel=%s field=%s seed=%d first=%d" % (case, N, P, B, chunks, kernel, k, seed, first), flush=True)
                break
    if case % 10 == 9:
        print("case %d done, %.0f s, mismatches %d" % (case + 1, time.time() - t0, bad), flush=True)
print("fuzz: %d cases, %d mismatches" % (n_cases, bad))
if os.environ.get("CRL_EXPECT_BOUNDS_BUILD") == "1":         # tools/gpu_bounds.sh: this run is on the bounds-assert build
    from colosseumrl_amd import _native
    rep = _native.bounds_report()
    wrong = (not rep["compiled"]) or any(rep[k]["failures"] for k in ("tron", "ttt", "blokus"))
    print("bounds asserts: %s %s" % ("FAILED" if wrong else "clean", rep))
    bad += 1 if wrong else 0
sys.exit(1 if bad else 0)
