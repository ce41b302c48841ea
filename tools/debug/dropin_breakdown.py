#!/usr/bin/env python3
"""Where the microseconds of the single-state Tron `next_state` go (B = 1, host-mapped memory, one launch + one wait):
    python tools/debug/dropin_breakdown.py [calls]
prints one JSON object (us per call, mean over `calls` after a warm-up):
  launch_wait        the bare floor: crl_tron_next_state_inplace64 through ctypes with pre-bound arguments + the wait on
                     mapped memory (crl_stream_wait_mapped), nothing else -- no state copied in or out
  launch_wait_obs    the same with the fused observations of all P players (the kernel stages the board, writes P copies)
  launch_sync        the same launch ended by hipStreamSynchronize instead
  stepper            SingleTron.next_state64: + the five copies into the mapped block, through the ONE-call form
                     (crl_tron_next_state_inplace64_host: player vectors by value, the kernel publishes completion);
                     stepper_two_calls: the launch + crl_stream_wait_mapped form on the same state
  env_next_state     TronGridEnvironment.next_state without fused observations (what is left is numpy: copies out, np.where,
                     the moves loop); env_next_state_obs with them (+ one 13 KB snapshot + the state's bytes as its identity)
  env_observation    state_to_observation served from the fused launch; env_observation_gpu on a state it has not seen
                     (crl_tron_relative_player_inplace64 + wait)
The reference's own next_state (Python + Cython, build container, profiles/reference_dropin.json) is ~10 us: the floor of a
GPU call alone is about that, so the single-state class cannot win on Tron -- it exists so that agents drop in unchanged.
"""
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
import json
import os
import random
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from colosseumrl_amd.config import get_environment  # noqa: E402


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    env = get_environment("tron")("20;4")
    rng = random.Random(0)
    state, players = env.new_state()
    st = env._single()
    lib = st._lib

    def mean_us(fn, n=calls):
        for _ in range(50):
            fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return round((time.perf_counter() - t0) / n * 1e6, 2)

    out = {}

    def raw(args):
        def f():
            lib.crl_tron_next_state_inplace64(*args)
            st.sync()
        return f
    st.next_state64(*state, np.zeros(4, np.int64), True)          # a valid state in the mapped block
    out["launch_wait"] = mean_us(raw(st._a_next64))
    out["launch_wait_obs"] = mean_us(raw(st._a_next64_obs))

    def launch_sync():
        lib.crl_tron_next_state_inplace64(*st._a_next64)
        lib.crl_stream_synchronize(st._stream)
    out["launch_sync"] = mean_us(launch_sync)
    acts = np.zeros(4, np.int64)
    out["stepper"] = mean_us(lambda: st.next_state64(*state, acts, False))
    out["stepper_obs"] = mean_us(lambda: st.next_state64(*state, acts, True))
    if st._unified:                                               # the same through the two-call form (launch, then signal kernel + spin)
        st._unified = False
        out["stepper_two_calls"] = mean_us(lambda: st.next_state64(*state, acts, False))
        out["stepper_two_calls_obs"] = mean_us(lambda: st.next_state64(*state, acts, True))
        st._unified = True

    cur = [state, players]

    def env_step():
        s, p = cur
        s, p, _, term, _ = env.next_state(s, p, [rng.choice(("forward", "right", "left")) for _ in p])
        if term:
            s, p = env.new_state()
        cur[0], cur[1] = s, p
    env._obs_idle = env.OBS_IDLE_STEPS                           # nobody asks for observations: the launch does not fuse them
    out["env_next_state"] = mean_us(env_step)

    def env_step_obs():
        env._obs_idle = 0
        env_step()
    out["env_next_state_obs"] = mean_us(env_step_obs)
    env._obs_idle = 0
    env_step()
    s = cur[0]
    out["env_observation"] = mean_us(lambda: env.state_to_observation(s, 1))
    other = tuple(a.copy() for a in s)
    other[0][0, 0] = 0 if other[0][0, 0] else 1

    def obs_gpu():
        env.state_to_observation(other, 1)
    out["env_observation_gpu"] = mean_us(obs_gpu)
    out["new_state"] = mean_us(lambda: env.new_state(), 300)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
