"""Vector-environment adapters over the batched steppers.

The reference exposes single games to RL libraries through ``RllibWrapper`` / ``TronRayEnvironment``
(colosseumrl/envs/wrappers/rllib.py:29-55, envs/tron/rllib.py:30-57): ``reset() -> obs`` and
``step(action_dict) -> obs, rewards, dones, infos`` per agent.  The adapters below give the same
reset/step contract for B games at once, with tensors instead of dicts of Python objects: actions in,
observations / rewards / dones out, everything staying on the GPU, games auto-resetting when they end.
"""
from typing import Dict, Tuple

import torch

from .batched import BlokusBatch, TronBatch, TTTBatch


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
        """-> (obs per player of the state AFTER auto-reset, rewards int8 [P, B], done uint8 [B], info).
        One launch: next_state and the observations of all players come out of the fused ``step_observe`` call."""
        o = self.batch.step_observe(actions, auto_reset=True)
        obs = {p: {"board": o["board"][p], "heads": o["heads"][p], "directions": o["directions"][p], "deaths": o["deaths"][p]}
               for p in range(self.num_players)}
        return obs, o["rewards"].clone(), o["terminal"].clone(), {"winners": o["winners"].clone()}


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
        player who just moved, done uint8 [B], info).  One launch (``TTTBatch.step_observe``)."""
        o = self.batch.step_observe(action, auto_reset=True)
        return ({"board": o["board"]}, o["mover"].clone(), o["valid"], o["reward"].clone(), o["terminal"].clone(),
                {"winners": o["winners"].clone()})


class BlokusVectorEnv:
    """B turn-based Blokus games (4 players, 20x20).  ``step`` takes int32 dense action ids [B] for the player to move
    (``envs.blokus.actions``: ``((piece*400 + y*20 + x)*8 + orientation)*5 + shift``; -1 = pass, the reference's '').
    Like the reference's ``next_state`` it does not validate actions (``match_server`` does, via ``is_valid_action``):
    pick them from ``valid_list`` / ``select`` / ``sample_valid`` (or test them with ``is_valid``)."""

    def __init__(self, batch: int = 1024, device="cuda"):
        self.batch = BlokusBatch(batch, device=device)
        self.num_players, self.num_envs = 4, batch

    def _mover(self) -> torch.Tensor:
        return self.batch.to_move.to(torch.int8)

    def reset(self):
        """-> (obs for the player to move, mover int8 [B], number of legal actions int32 [B])."""
        self.batch.reset()
        mover = self._mover()
        return self.batch.observe(mover), mover, self.batch.valid()

    def valid_mask(self) -> torch.Tensor:
        """Dense legal-action bitmap int32 [B, 10500] of the player to move (42 KB per game: meant for small B)."""
        return self.batch.valid(want_mask=True)[1]

    def valid_list(self, cap: int = 2048, out=None):
        """(count int32 [B], ids int32 [B, cap]): the ORDERED legal action ids of the player to move, compacted -- what the
        reference's ``valid_actions`` returns (BlokusEnvironment.py:453-500), for every game; ``ids[b, :count[b]]`` ascending
        = reference order, -1 beyond.  1,693 is the longest list seen in reference-played games."""
        return self.batch.valid_list(cap, out=out)

    def select(self, rank: torch.Tensor) -> torch.Tensor:
        """Dense id of the rank[b]-th legal action (reference order) of the player to move, -1 where rank is outside
        [0, count): lets a policy that emits an index into the legal list act without materialising the list."""
        return self.batch.select(rank)[0]

    def is_valid(self, action: torch.Tensor) -> torch.Tensor:
        """uint8 [B]: 1 iff action[b] is a legal action of the player to move (``is_valid_action`` for all games)."""
        return self.batch.is_valid(action)

    def sample_valid(self, seed: int = 0) -> torch.Tensor:
        """A uniformly drawn legal action id per game (-1 where the mover must pass)."""
        return self.batch.sample(seed)

    def step(self, action: torch.Tensor, list_cap: int = 0):
        """-> (obs for the next mover, next mover int8 [B], its number of legal actions int32 [B], reward int8 [B] of the
        player who just moved, done uint8 [B], info); finished games restart (obs / mover / counts are of the new game).
        One launch (``BlokusBatch.step_observe``); with ``list_cap`` > 0 a second one right behind it leaves the next mover's
        ordered legal ids in ``info['ids']`` (int32 [B, list_cap], -1 beyond the count)."""
        o = self.batch.step_observe(action, auto_reset=True, list_cap=list_cap)
        obs = {"board": o["board"], "pieces": o["pieces"], "score": o["score"], "player": o["player"]}
        info = {"winners": o["winners"].clone()}
        if list_cap > 0:
            info["ids"] = o["ids"]
        return obs, o["player"].view(-1), o["n_valid"], o["reward"].clone(), o["terminal"].clone(), info
