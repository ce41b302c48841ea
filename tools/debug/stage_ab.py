import sys, time, random
sys.path.insert(0, "/root/repo")
from colosseumrl_amd import get_environment
def run(name, cfg, skip, n=3000):
    env = get_environment(name)(*cfg)
    rng = random.Random(0)
    state, players = env.new_state()
    t_next = 0.0
    for i in range(n + 200):
        va = env.valid_actions(state, players[0])
        acts = [rng.choice(va) for _ in players]
        if not skip:
            env._staged = None
        t0 = time.perf_counter()
        state, players, _, term, _ = env.next_state(state, players, acts)
        if i >= 200:
            t_next += time.perf_counter() - t0
        if term:
            state, players = env.new_state()
    return t_next / n * 1e6
for name, cfg, n in (("tron", ("20;4",), 3000), ("tictactoe_3p", (), 3000), ("blokus", (), 400)):
    for rnd in range(2):
        print(name, "skip-staging %.1f us   always-stage %.1f us" % (run(name, cfg, True, n), run(name, cfg, False, n)), flush=True)
