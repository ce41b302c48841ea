"""Tron behind the multi-agent dict API (colosseumrl/envs/tron/TronRllibEnvironment.py:10-38): integer actions
0 / 1 / anything else = forward / right / left, a player is done when it is no longer among the players to move."""
import numpy as np

from ...BaseEnvironment import BaseEnvironment
from ..wrappers import RllibWrapper
from ..wrappers.spaces import Box, Dict, Discrete
from .TronGridEnvironment import TronGridEnvironment


def tron_observation_space(board_size: int, num_players: int):
    return Dict({
        "board": Box(0, num_players, shape=(board_size, board_size)),
        "heads": Box(0, np.inf, shape=(num_players,)),
        "directions": Box(0, 4, shape=(num_players,)),
        "deaths": Box(0, num_players, shape=(num_players,)),
    })


class TronRllibEnvironment(RllibWrapper):
    def create_env(self, *args, **kwargs) -> BaseEnvironment:
        return TronGridEnvironment.create(*args, **kwargs)

    def create_observation_space(self, *args, **kwargs):
        return tron_observation_space(self.env.N, self.env.num_players)

    def create_action_space(self, *args, **kwargs):
        return Discrete(3)

    def create_done_dict(self, state, players, rewards, terminal, action_dict):
        alive = {str(p) for p in players}
        return {key: bool(terminal or key not in alive) for key in action_dict}

    def action_map(self, action):
        return ("forward", "right")[action] if action in (0, 1) else "left"
