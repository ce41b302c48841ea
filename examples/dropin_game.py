#!/usr/bin/env python3
"""One game through the drop-in classes, exactly as with the reference package -- only the import differs:

    from colosseumrl.config import get_environment        # reference
    from colosseumrl_amd.config import get_environment    # this package (needs an MI355X)

    python examples/dropin_game.py [tron|tictactoe|tictactoe_3p|tictactoe_4p|blokus] [seed]
"""
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:
    print(__doc__)
    sys.exit(0)
import os
import random

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colosseumrl_amd.config import get_environment  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tron"
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
env = get_environment(name)("15;4") if name == "tron" else get_environment(name)()
state, players = env.new_state()
steps, terminal, winners = 0, False, None
while not terminal:
    actions = []
    for player in players:
        legal = env.valid_actions(state, player)
        action = rng.choice(legal)
        assert action == "" or env.is_valid_action(state, player, action)
        actions.append(action)
    state, players, rewards, terminal, winners = env.next_state(state, players, actions)
    steps += 1
ranking = env.compute_ranking(state, list(players), list(winners) if winners is not None else [])
print("%s: %d steps, winners %s, ranking %s" % (name, steps, list(winners) if winners is not None else None, dict(ranking)))
