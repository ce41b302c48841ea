"""Vector-environment adapters over the batched steppers.

The reference exposes single games to RL libraries through ``RllibWrapper`` / ``TronRayEnvironment``
(colosseumrl/envs/wrappers/rllib.py:29-55, envs/tron/rllib.py:30-57): ``reset() -> obs`` and
``step(action_dict) -> obs, rewards, dones, infos`` per agent.  The adapters below give the same
reset/step contract for B games at once, with tensors instead of dicts of Python objects: actions in,
observations / rewards / dones out, everything staying on the GPU, games auto-resetting when they end.
"""
from typing import Dict, Tuple

import torch

from .batched import TronBatch, TTTBatch


class TronVectorEnv:
    """B simultaneous-move Tron games.  ``step`` takes int8 actions [P, B] in {0 forward, 1 right, -1 left}."""

    def __init__(self, board_size: int = 19, num_players: int = 4, batch: int = 1024, device="cuda"):
        self.batch = TronBatch(board_size, num_players, batch, device=device)
        self.num_players, self.num_envs = num_players, batch

    def observe(self, player: int) -> Dict[str, torch.Tensor]:
        pl = torch.full((self.num_envs,), player, dtype=torch.int8, device=self.batch.device)
        return self.batch.observe(pl)

    def reset(self) -> Dict[int, Dict[str, torch.Tensor]]:
        self.batch.reset()
        return {p: self.observe(p) for p in range(self.num_players)}

    def step(self, actions: torch.Tensor) -> Tuple[Dict[int, Dict[str, torch.Tensor]], torch.Tensor, torch.Tensor, Dict]:
        """-> (obs per player of the state AFTER auto-reset, rewards int8 [P, B], done uint8 [B], info)."""
        rewards, terminal, winners = self.batch.step(actions, auto_reset=True)
        rewards, done, winners = rewards.clone(), terminal.clone(), winners.clone()
        return {p: self.observe(p) for p in range(self.num_players)}, rewards, done, {"winners": winners}


class TicTacToeVectorEnv:
    """B turn-based TicTacToe games.  ``step`` takes int8 cell indices [B] for the player to move (-1 = pass)."""

    def __init__(self, dims=(3, 3), k: int = 3, num_players: int = 2, batch: int = 1024, device="cuda"):
        self.batch = TTTBatch(dims, k, num_players, batch, device=device)
        self.num_players, self.num_envs = num_players, batch

    def reset(self):
        self.batch.reset()
        return self.batch.observe(self.batch.to_move), self.batch.to_move.clone(), self.batch.valid_mask()

    def step(self, action: torch.Tensor):
        """-> (obs for the next mover, next mover int8 [B], empties bitmask int32 [B], reward int8 [B] of the
        player who just moved, done uint8 [B], info)."""
        reward, terminal, winners = self.batch.step(action, auto_reset=True)
        reward, done, winners = reward.clone(), terminal.clone(), winners.clone()
        mover = self.batch.to_move
        return self.batch.observe(mover), mover.clone(), self.batch.valid_mask(), reward, done, {"winners": winners}
