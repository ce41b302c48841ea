import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from oracle import oracle as O
from backends import HipTron
O.build()
def run(N, P, B, chunks, kernel):
    sh, sd = O.tron_start_positions(N, P)
    hip = HipTron(N, P, B, sh, sd); hip.tb.first_env_id = 123456
    ost = O.TronState(N, P, B); O.tron_reset(ost, sh, sd)
    seed = 0xC0FFEE12345
    for T in chunks:
        hip.tb.rollout(T, seed, kernel=kernel)
        O.tron_rollout(ost, seed, 123456, T, sh, sd, n_threads=8)
    bad = {}
    for k in ("board", "heads", "dirs", "deaths", "tstep", "n_episodes", "ret_sum"):
        want = getattr(ost, k); have = getattr(hip.tb, k).cpu().numpy().view(want.dtype)
        if not np.array_equal(have, want):
            idx = np.argwhere(have != want)
            bad[k] = (len(idx), idx[:3].tolist())
    print(N, P, B, chunks, kernel, "OK" if not bad else bad, "max n_ep", int(ost.n_episodes.max()))
for args in [(40, 4, 2048, (100,)), (40, 4, 64, (3,)), (40, 4, 64, (30,)), (36, 4, 64, (30,)), (24, 4, 64, (30,))]:
    run(*args, "bits")
