#!/usr/bin/env python3
"""Per-call latency of the REFERENCE's single-state API (build container only; the reference cannot travel).

    python tools/time_reference_dropin.py            # prints one JSON object and writes profiles/reference_dropin.json

Times new_state / next_state / valid_actions / state_to_observation of the reference envs imported from
/root/reference (oracle/ref_loader.py) on one core, random play with restarts at terminal: the figures bench.py's
`dropin` section prints beside this package's own per-call latencies (it reads profiles/reference_dropin.json).
TicTacToe is driven with canonical '(r, c)' strings (under numpy 2 the reference's own valid_actions strings
do not parse), chosen from the empty cells; Blokus runs un-jitted (numba absent) and is sampled for a few plies only.
"""
import json
import os
import random
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, n):
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


def play(env, pick, n_next, n_other, rng):
    out = {}
    out["new_state"] = timeit(lambda: env.new_state(), max(3, n_other // 4))
    state, players = env.new_state()
    t_next = t_valid = t_obs = 0.0
    k_valid = k_obs = 0
    for i in range(n_next):
        if i < n_other:
            t0 = time.perf_counter()
            va = env.valid_actions(state, players[0])
            t_valid += time.perf_counter() - t0
            k_valid += 1
            t0 = time.perf_counter()
            env.state_to_observation(state, players[0])
            t_obs += time.perf_counter() - t0
            k_obs += 1
        else:
            va = None
        actions = pick(env, state, players, va, rng)
        t0 = time.perf_counter()
        state, players, rewards, terminal, winners = env.next_state(state, players, actions)
        t_next += time.perf_counter() - t0
        if terminal:
            state, players = env.new_state()
    out["next_state"] = t_next / n_next * 1e6
    out["valid_actions"] = t_valid / max(k_valid, 1) * 1e6
    out["state_to_observation"] = t_obs / max(k_obs, 1) * 1e6
    return out


def pick_tron(env, state, players, va, rng):
    return [rng.choice(["forward", "right", "left"]) for _ in players]


def pick_ttt(env, state, players, va, rng):
    import numpy as np
    cells = np.argwhere(state[0] == -1)
    if len(cells) == 0:
        return [""]
    c = cells[rng.randrange(len(cells))]
    return [str(tuple(int(x) for x in c))]


def pick_blokus(env, state, players, va, rng):
    if va is None:
        va = env.valid_actions(state, players[0])
    return [rng.choice(va)]


def main():
    from oracle import ref_loader
    R = ref_loader.load()
    rng = random.Random(0)
    out = {}
    out["tron"] = play(R["tron"]("20;4"), pick_tron, 20000, 2000, rng)
    out["tictactoe"] = play(R["ttt2"](), pick_ttt, 4000, 1000, rng)
    out["tictactoe_3p"] = play(R["ttt3"](), pick_ttt, 4000, 1000, rng)
    out["tictactoe_4p"] = play(R["ttt4"](), pick_ttt, 1000, 300, rng)
    out["blokus"] = play(R["blokus"](), pick_blokus, 12, 12, rng)
    out = {k: {m: round(v, 1) for m, v in d.items()} for k, d in out.items()}
    out["where"] = "build container, one core, reference imported from /root/reference (numba absent: Blokus un-jitted)"
    out["script"] = "tools/time_reference_dropin.py"
    with open(os.path.join(ROOT, "profiles", "reference_dropin.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
