#!/usr/bin/env python3
"""Randomised differential run of the TicTacToe and Blokus rollouts against the CPU oracle (soak aid, not a test):
random board shapes / K / player counts, ragged batches, split launches.  usage: ttt_blokus_fuzz.py [n_ttt] [n_blokus] [seed]"""
import os, sys, time
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from colosseumrl_amd.batched import TTTBatch, BlokusBatch
from oracle import oracle as O

n_ttt = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_blk = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
bad, t0 = 0, time.time()
for case in range(n_ttt):
    nd = int(rng.integers(1, 4))
    while True:
        dims = tuple(int(rng.integers(1, 9)) for _ in range(nd))
        if 1 <= int(np.prod(dims)) <= 32:
            break
    K, P, B = int(rng.integers(1, 7)), int(rng.integers(2, 9)), int(rng.integers(1, 5000))
    chunks = [int(rng.integers(1, 900)) for _ in range(int(rng.integers(1, 4)))]
    seed, first = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 40))
    try:
        ost = O.TTTState(dims, K, P, B)
        tb = TTTBatch(dims, K, P, B, first_env_id=first)
    except Exception as exc:
        continue
    for T in chunks:
        O.ttt_rollout(ost, seed, first, T, n_threads=16)
        tb.rollout(T, seed)
    for k in ("occ", "winner", "to_move", "tcount", "tstep", "n_episodes", "win_count", "draw_count", "len_sum"):
        want = getattr(ost, k)
        if not np.array_equal(getattr(tb, k).cpu().numpy().view(want.dtype), want):
            bad += 1
            print("TTT MISMATCH case %d dims=%s K=%d P=%d B=%d chunks=%s field=%s" % (case, dims, K, P, B, chunks, k), flush=True)
            break
    if case % 100 == 99:
        print("ttt case %d done, %.0f s, mismatches %d" % (case + 1, time.time() - t0, bad), flush=True)
for case in range(n_blk):
    B = int(rng.integers(1, 700))
    chunks = [int(rng.integers(1, 120)) for _ in range(int(rng.integers(1, 4)))]
    seed, first = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 40))
    bb = BlokusBatch(B, first_env_id=first)
    ost = O.BlokusState(B)
    for T in chunks:
        bb.rollout(T, seed)
        O.blokus_rollout(ost, seed, first, T, n_threads=16)
    for k in ("occ", "inv", "score", "round", "to_move", "tcount", "tstep", "n_episodes", "win_count", "len_sum", "score_sum"):
        want = getattr(ost, k)
        if not np.array_equal(getattr(bb, k).cpu().numpy().view(want.dtype), want):
            bad += 1
            print("BLOKUS MISMATCH case %d B=%d chunks=%s field=%s" % (case, B, chunks, k), flush=True)
            break
    print("blokus case %d done, %.0f s, mismatches %d" % (case + 1, time.time() - t0, bad), flush=True)
print("fuzz: %d + %d cases, %d mismatches" % (n_ttt, n_blk, bad))
if os.environ.get("CRL_EXPECT_BOUNDS_BUILD") == "1":         # tools/gpu_bounds.sh: this run is on the bounds-assert build
    from colosseumrl_amd import _native
    rep = _native.bounds_report()
    wrong = (not rep["compiled"]) or any(rep[k]["failures"] for k in ("tron", "ttt", "blokus"))
    print("bounds asserts: %s %s" % ("FAILED" if wrong else "clean", rep))
    bad += 1 if wrong else 0
sys.exit(1 if bad else 0)
