#!/usr/bin/env python3
"""B games resident on the GPU: fused random-agent rollouts, and a policy in the loop through the vector-env adapter.

    python examples/batched_rollout.py [games=65536]
"""
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:
    print(__doc__)
    sys.exit(0)
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from colosseumrl_amd.batched import TronBatch  # noqa: E402
from colosseumrl_amd.vector import BlokusVectorEnv, TronVectorEnv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
# 1. fused rollouts: T env-steps of every game per launch, uniform random agent, auto-reset; statistics stay on the device
tron = TronBatch(board_size=20, num_players=4, batch=B)
tron.rollout(64, seed=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
tron.rollout(8192, seed=0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
rows = tron.results()                                  # int32 [B, 3 + 2P]: n_episodes, len_sum, last_winners, win_count[P], ret_sum[P]
print("Tron 20x20 P4, %d games: %.3g env-steps/s, %d episodes, mean length %.2f"
      % (B, B * 8192 / dt, int(rows[:, 0].sum()), rows[:, 1].sum().item() / max(1, rows[:, 0].sum().item())))
# 2. a policy in the loop: one launch per step returns every player's observation of every game
env = TronVectorEnv(board_size=20, num_players=4, batch=4096)
obs = env.reset()
for _ in range(50):
    actions = torch.randint(-1, 2, (4, 4096), dtype=torch.int8, device="cuda")     # your policy(obs) goes here
    obs, rewards, done, info = env.step(actions)
print("TronVectorEnv: observation of player 0", tuple(obs[0]["board"].shape), "rewards", tuple(rewards.shape), "done", int(done.sum()))
# 3. Blokus: the ordered legal-action ids come back with the step
blokus = BlokusVectorEnv(batch=1024)
obs, mover, n_valid = blokus.reset()
ids = blokus.valid_list(2048)[1]
for _ in range(20):
    pick = (torch.rand(1024, device="cuda") * n_valid.clamp(min=1)).long()                 # a random index into each legal list
    action = torch.where(n_valid > 0, ids.gather(1, pick[:, None])[:, 0], torch.full_like(n_valid, -1))
    obs, mover, n_valid, reward, done, info = blokus.step(action.int(), list_cap=2048)
    ids = info["ids"]
print("BlokusVectorEnv: 20 plies of 1024 games, legal actions of the movers now: mean %.0f" % n_valid.float().mean().item())
