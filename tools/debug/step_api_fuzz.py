#!/usr/bin/env python3
"""Randomised twin run of the per-step API (soak aid): the fused step_observe calls against their unfused equivalents and,
through the rollout equivalence (sample + step T times == rollout(T)), against the rollout kernels, on random shapes and
ragged batches.  usage: step_api_fuzz.py [n_cases] [seed]"""
import os, sys, time
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ctypes as C
from colosseumrl_amd._native import check
from colosseumrl_amd.batched import TronBatch, TTTBatch

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad, t0 = 0, time.time()
for case in range(n_cases):
    N, P, B = int(rng.integers(4, 41)), int(rng.integers(2, 9)), int(rng.integers(1, 2500))
    P = min(P, 4) if N == 4 else P
    if rng.random() < 0.35:
        B = (B // 16 + 1) * 16                                  # whole 16-game groups: the flat-stream kernels of boards like 19 x 19
    T, seed, first = int(rng.integers(1, 40)), int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 40))
    a, b, c = (TronBatch(N, P, B, first_env_id=first) for _ in range(3))
    step_kernel = ("auto", "bytes", "staged")[case % 3]        # crl_tron_step's interchangeable kernels (staged: where the board allows)
    for t in range(T):
        # (shapes the 16-byte fused kernel cannot take -- N*N % 16 != 0, P = 8 -- run the one-game-per-workgroup kernel)
        oa = a.step_observe(None, seed=seed)
        acts = b.sample(seed)
        if B <= 300 and t < 6:                                  # the Cython signature: int64 reference layout, in place, B games
            brd64 = b.board.to(torch.int64).contiguous()
            vec64 = [x.t().to(torch.int64).contiguous() for x in (b.heads, b.dirs, b.deaths, acts)]
            rew64 = torch.zeros((B, P), dtype=torch.int64, device="cuda")
            trm64 = torch.zeros((B,), dtype=torch.uint8, device="cuda")
            ob64 = torch.zeros((B, P, N * N), dtype=torch.int64, device="cuda")
            ov64 = [torch.zeros((B, P, P), dtype=torch.int64, device="cuda") for _ in range(3)]
            ptr = lambda x: C.c_void_p(x.data_ptr())                                  # noqa: E731
            check(b._lib.crl_tron_next_state_inplace64(b._ctx.handle, B, ptr(brd64), ptr(vec64[0]), ptr(vec64[1]), ptr(vec64[2]), ptr(vec64[3]),
                                                       ptr(rew64), ptr(trm64), None, ptr(ob64), ptr(ov64[0]), ptr(ov64[1]), ptr(ov64[2]), None),
                  "crl_tron_next_state_inplace64")
            torch.cuda.synchronize()
        else:
            brd64 = None
        rew, term, _ = b.step(acts, auto_reset=brd64 is None, kernel=step_kernel)
        if brd64 is not None:
            same = (torch.equal(brd64, b.board.to(torch.int64)) and torch.equal(vec64[0], b.heads.t().to(torch.int64))
                    and torch.equal(vec64[1], b.dirs.t().to(torch.int64)) and torch.equal(vec64[2], b.deaths.t().to(torch.int64))
                    and torch.equal(rew64, rew.t().to(torch.int64)) and torch.equal(trm64, term))
            pre = b.observe_all()                               # observations of the un-reset new state: what the int64 entry wrote
            same = same and all(torch.equal(ob64[:, p], pre["board"][p].reshape(B, -1).to(torch.int64)) for p in range(P)) \
                and all(torch.equal(ov64[0][:, p].t(), pre["heads"][p].to(torch.int64)) for p in range(P))
            if not same:
                bad += 1
                print("TRON next_state_inplace64 MISMATCH case %d N=%d P=%d B=%d t=%d" % (case, N, P, B, t), flush=True)
            b.reset(term)                                       # step + masked reset == step with auto-reset
        ob = b.observe_all()
        for k in ("board", "heads", "directions", "deaths"):
            if not torch.equal(oa[k], ob[k]):
                bad += 1
                print("TRON step_observe MISMATCH case %d N=%d P=%d B=%d t=%d field=%s" % (case, N, P, B, t, k), flush=True)
                break
    c.rollout(T, seed)
    for k in ("board", "heads", "dirs", "deaths", "tcount"):
        if not (torch.equal(getattr(a, k), getattr(c, k)) and torch.equal(getattr(b, k), getattr(c, k))):
            bad += 1
            print("TRON step-vs-rollout MISMATCH case %d N=%d P=%d B=%d T=%d field=%s" % (case, N, P, B, T, k), flush=True)
            break
    # TicTacToe
    nd = int(rng.integers(1, 4))
    while True:
        dims = tuple(int(rng.integers(1, 9)) for _ in range(nd))
        if 1 <= int(np.prod(dims)) <= 32:
            break
    K, Pt, Bt = int(rng.integers(1, 7)), int(rng.integers(2, 9)), int(rng.integers(1, 4000))
    try:
        x, y, z = (TTTBatch(dims, K, Pt, Bt, first_env_id=first) for _ in range(3))
    except Exception:
        continue
    for t in range(T):
        ox = x.step_observe(None, seed=seed)
        y.step(y.sample(seed), auto_reset=True)
        vb, bb = y.valid_mask(), y.board(y.to_move, Pt)
        if not (torch.equal(ox["valid"], vb) and torch.equal(ox["board"], bb)):
            bad += 1
            print("TTT step_observe MISMATCH case %d dims=%s K=%d P=%d B=%d t=%d" % (case, dims, K, Pt, Bt, t), flush=True)
            break
    z.rollout(T, seed)
    for k in ("occ", "winner", "to_move", "tcount"):
        if not (torch.equal(getattr(x, k), getattr(z, k)) and torch.equal(getattr(y, k), getattr(z, k))):
            bad += 1
            print("TTT step-vs-rollout MISMATCH case %d dims=%s K=%d P=%d B=%d T=%d field=%s" % (case, dims, K, Pt, Bt, T, k), flush=True)
            break
    if case % 20 == 19:
        print("case %d done, %.0f s, mismatches %d" % (case + 1, time.time() - t0, bad), flush=True)
print("fuzz: %d cases, %d mismatches" % (n_cases, bad))
if os.environ.get("CRL_EXPECT_BOUNDS_BUILD") == "1":         # tools/gpu_bounds.sh: this run is on the bounds-assert build
    from colosseumrl_amd import _native
    rep = _native.bounds_report()
    wrong = (not rep["compiled"]) or any(rep[k]["failures"] for k in ("tron", "ttt", "blokus"))
    print("bounds asserts: %s %s" % ("FAILED" if wrong else "clean", rep))
    bad += 1 if wrong else 0
sys.exit(1 if bad else 0)
