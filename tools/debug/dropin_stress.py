#!/usr/bin/env python3
"""Stress of the single-state drop-in path (host-mapped staging, end-of-call wait on mapped memory): T threads, each with
its own env instances, play random games back to back for `calls` next_state calls per env kind and check every step
against the CPU oracle (Tron) or against invariants (TicTacToe / Blokus); reports calls per second and the slowest call.
    python tools/debug/dropin_stress.py [threads=4] [calls=20000]"""
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
import os
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from colosseumrl_amd import get_environment  # noqa: E402
from oracle import oracle as O  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
CODE = {"forward": 0, "right": 1, "left": -1}
errors, stats = [], []


def tron(tid):
    rng = np.random.default_rng(tid)
    env = get_environment("tron")("13;4")
    sh, sd = O.tron_start_positions(13, 4)
    n, worst = 0, 0.0
    while n < CALLS:
        state, players = env.new_state()
        st = O.TronState(13, 4, 1)
        O.tron_reset(st, sh, sd)
        while n < CALLS:
            acts = [("forward", "right", "left")[int(rng.integers(0, 3))] for _ in players]
            a = np.zeros((4, 1), np.int8)
            for p, s in zip(players, acts):
                a[p, 0] = CODE[s]
            t0 = time.perf_counter()
            state, players, rewards, term, winners = env.next_state(state, players, acts)
            worst = max(worst, time.perf_counter() - t0)
            r2, t2, _ = O.tron_step(st, a)
            n += 1
            if not (np.array_equal(state[0].reshape(-1), st.board[0]) and np.array_equal(state[1], st.heads[:, 0])
                    and rewards.tolist() == r2[:, 0].tolist() and bool(term) == bool(t2[0])):
                errors.append(("tron", tid, n))
                return
            if term:
                break
    stats.append(("tron", tid, n, worst))


def ttt(tid):
    rng = np.random.default_rng(100 + tid)
    env = get_environment("tictactoe_3p")()
    n, worst = 0, 0.0
    while n < CALLS:
        state, players = env.new_state()
        filled = 0
        while n < CALLS:
            va = env.valid_actions(state, players[0])
            if len(va) != 15 - filled:
                errors.append(("ttt valid count", tid, n))
                return
            t0 = time.perf_counter()
            state, players, rewards, term, winners = env.next_state(state, players, [va[int(rng.integers(0, len(va)))]])
            worst = max(worst, time.perf_counter() - t0)
            n += 1
            filled += 1
            if int((state[0] >= 0).sum()) != filled:
                errors.append(("ttt board", tid, n))
                return
            if term:
                break
    stats.append(("ttt", tid, n, worst))


def blokus(tid):
    rng = np.random.default_rng(200 + tid)
    env = get_environment("blokus")()
    n, worst = 0, 0.0
    while n < CALLS // 10:
        state, players = env.new_state()
        cells = 0
        while n < CALLS // 10:
            va = env.valid_actions(state, players[0])
            pick = va[int(rng.integers(0, len(va)))]
            t0 = time.perf_counter()
            state, players, rewards, term, winners = env.next_state(state, players, [pick])
            worst = max(worst, time.perf_counter() - t0)
            n += 1
            if pick:
                cells += {"mon": 1, "dom": 2, "tro": 3, "tet": 4, "pen": 5}[pick[:3]]
            if int((state[0].board_contents != 0).sum()) != cells:
                errors.append(("blokus cells", tid, n))
                return
            if term:
                break
    stats.append(("blokus", tid, n, worst))


t0 = time.perf_counter()
threads = [threading.Thread(target=f, args=(i,)) for i in range(T) for f in (tron, ttt, blokus)]
for th in threads:
    th.start()
for th in threads:
    th.join()
dt = time.perf_counter() - t0
total = sum(s[2] for s in stats)
for kind in ("tron", "ttt", "blokus"):
    rows = [s for s in stats if s[0] == kind]
    if rows:
        print("%-7s %d threads, %d calls, slowest call %.0f us" % (kind, len(rows), sum(r[2] for r in rows), max(r[3] for r in rows) * 1e6))
print("dropin stress: %d next_state calls on %d threads in %.1f s (%.0f calls/s), %d errors %s" % (total, len(threads), dt, total / dt, len(errors), errors[:3]))
sys.exit(1 if errors or len(stats) != len(threads) else 0)
