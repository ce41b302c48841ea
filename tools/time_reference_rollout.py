#!/usr/bin/env python3
"""env-steps/s of the REFERENCE's own random-agent rollout loop (build container only; the reference cannot travel).

    python tools/time_reference_rollout.py [--seconds S] [--out profiles/reference_python.json]

The loop BASELINE.md section 2 describes, on the reference envs imported from /root/reference (oracle/ref_loader.py;
Tron through the reference's own CyTronGrid.pyx compiled by oracle/build_ref.py), one process, one core: new_state,
then next_state with uniformly random actions for every current player until terminal, then new_state again -- resets
are inside the clock, exactly as the auto-reset is inside the GPU rollout's.  One env-step = one next_state call.
TicTacToe is driven with canonical '(r, c)' strings chosen from the empty cells (under numpy 2 the
reference's own valid_actions strings do not parse); Blokus calls valid_actions + next_state per ply, un-jitted (numba
is absent from the image) and is bounded to one short game.  Writes profiles/reference_python.json: per bench.py
workload {value, steps, seconds, episodes, mean_episode_len} + nproc, versions, "no numba".  bench.py's `cpu_baseline.
reference_python` reads that file; there is no constant to fall back to.
"""
import argparse
import json
import os
import platform
import random
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without importing anything heavy (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pick_tron(env, state, players, rng):
    return [rng.choice(("forward", "right", "left")) for _ in players]


def pick_ttt(env, state, players, rng):
    import numpy as np
    cells = np.argwhere(state[0] == -1)
    if len(cells) == 0:
        return [""]
    return [str(tuple(int(x) for x in cells[rng.randrange(len(cells))]))]


def pick_blokus(env, state, players, rng):
    return [rng.choice(env.valid_actions(state, players[0]))]


def rollout(env, pick, seconds, rng, max_steps=None):
    """Random-agent loop with restarts for ~`seconds` (or `max_steps`); the clock covers new_state, the agent's pick and
    next_state -- everything a rollout worker on the reference spends per env-step."""
    steps = episodes = ep_steps = 0
    t0 = time.perf_counter()
    state, players = env.new_state()
    while True:
        state, players, _, terminal, _ = env.next_state(state, players, pick(env, state, players, rng))
        steps += 1
        ep_steps += 1
        if terminal:
            episodes += 1
            ep_steps = 0
            state, players = env.new_state()
        if (steps & 63) == 0 or max_steps:
            dt = time.perf_counter() - t0
            if dt >= seconds or (max_steps and steps >= max_steps):
                break
    dt = time.perf_counter() - t0
    finished = steps - ep_steps
    return {"value": steps / dt, "unit": "env-steps/s", "steps": steps, "seconds": round(dt, 2), "episodes": episodes,
            "mean_episode_len": round(finished / episodes, 2) if episodes else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--blokus-steps", type=int, default=24)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "reference_python.json"))
    args = ap.parse_args()
    from oracle import ref_loader
    R = ref_loader.load()
    import numpy
    import scipy
    rng = random.Random(0)
    wl = {}
    wl["tron_p4_n20_b65536"] = dict(rollout(R["tron"]("20;4"), pick_tron, args.seconds, rng), env="tron '20;4'")
    wl["tron_p4_n40_b65536"] = dict(rollout(R["tron"]("40;4"), pick_tron, args.seconds, rng), env="tron '40;4'")
    wl["ttt_p2_3x3_k3"] = dict(rollout(R["ttt2"](), pick_ttt, args.seconds / 2, rng), env="tictactoe (2p 3x3)")
    wl["ttt_p3_3x5_k3_b262144"] = dict(rollout(R["ttt3"](), pick_ttt, args.seconds / 2, rng), env="tictactoe_3p (3x5)")
    wl["ttt_p4_3x3x3_b262144"] = dict(rollout(R["ttt4"](), pick_ttt, args.seconds / 2, rng), env="tictactoe_4p (3x3x3)")
    wl["blokus_p4_b16384"] = dict(rollout(R["blokus"](), pick_blokus, 1e9, rng, max_steps=args.blokus_steps),
                                  env="blokus (valid_actions + next_state per ply, un-jitted)")
    try:
        import numba                                          # noqa: F401
        jit = "numba %s" % numba.__version__                 # (oracle/ref_loader.py's identity-decorator shell has no version)
    except (ImportError, AttributeError):
        jit = "no numba (the reference's documented fallback: computation.py:15-20): Blokus runs as plain Python"
    out = {"workloads": wl,
           "what": "the reference's own envs (imported from /root/reference), random agent incl. new_state resets, one process, one core",
           "where": "build container (the Python reference cannot travel to the GPU box)",
           "nproc": len(os.sched_getaffinity(0)), "cores_used": 1, "jit": jit,
           "versions": {"python": platform.python_version(), "numpy": numpy.__version__, "scipy": scipy.__version__},
           "script": "tools/time_reference_rollout.py --seconds %g --blokus-steps %d" % (args.seconds, args.blokus_steps)}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: round(v["value"], 2) for k, v in wl.items()}))


if __name__ == "__main__":
    main()
