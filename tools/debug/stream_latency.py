#!/usr/bin/env python3
"""Launch + synchronise latency of a B = 1 TicTacToe step on mapped memory, per stream, for the first 8 non-blocking
streams a process creates (bring-up aid: the drop-in classes own one stream each; HIP maps streams onto a few hardware
queues)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from colosseumrl_amd.single import SingleTTT  # noqa: E402

envs = [SingleTTT((3, 5), 3, 3, 3) for _ in range(8)]
for rnd in range(2):
    for i, st in enumerate(envs):
        st.load([-1] * 15, None, 0)
        for _ in range(50):
            st.step(-1)
        ts = []
        for _ in range(2000):
            t0 = time.perf_counter()
            st.step(-1)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print("round %d stream %d (handle %x): step median %.1f us  p10 %.1f  p90 %.1f" % (rnd, i, st._stream.value, ts[1000] * 1e6, ts[200] * 1e6, ts[1800] * 1e6), flush=True)
