"""Blokus -- drop-in for ``colosseumrl.envs.blokus.BlokusEnvironment``.

Same state tuple ``(Board, round_count, [AI x 4])``, action strings and return types as the reference
(colosseumrl/envs/blokus/BlokusEnvironment.py:188-768).  Move generation and the step rule run on the GPU
through a B=1 ``BlokusBatch`` (HIP kernels behind the C ABI); without an MI355X these methods raise.
For throughput use ``colosseumrl_amd.batched.BlokusBatch`` directly.
"""
from typing import Dict, List, Tuple, Union

import dill
import numpy as np

from ...BaseEnvironment import BaseEnvironment
from . import actions as A
from .actions import action_to_string, string_to_action  # noqa: F401  (module-level API of the reference)
from .ai import AI
from .board import (BOARD_TO_PLAYER_OBSERVATION_ROTATION_MATRICES, PLAYER_OBSERVATION_TO_BOARD_ROTATION_MATRICES,
                    Board)

PLAYER_TO_COLOR = {0: 1, 1: 2, 2: 3, 3: 4}
COLOR_TO_PLAYER = {1: 0, 2: 1, 3: 2, 4: 3, 0: -1}
PIECE_NAME_TO_INDEX = dict(A.PIECE_INDEX)
PIECE_TYPES = {name: None for name in A.PIECE_NAMES}      # key order = piece order (cell tables live on the GPU)
ORIENTATIONS = A.ORIENTATIONS
State = object


def print_board(state: object):
    """Board to stdout, -1 = empty, else the player number (reference :166-185)."""
    print(state[0].board_contents - 1)


class BlokusEnvironment(BaseEnvironment):

    def __init__(self, config: str = "", device="cuda"):
        super().__init__(config)
        self._device = device
        self._stepper = None

    @property
    def min_players(self) -> int:
        return 4

    @property
    def max_players(self) -> int:
        return 4

    @property
    def observation_shape(self) -> Dict[str, Tuple[int, ...]]:
        return {"board": (20, 20), "pieces": (4, 21), "score": (4,), "player": (1,)}

    @staticmethod
    def observation_names():
        return ["board", "pieces", "score", "player"]

    @staticmethod
    def all_piece_types():
        return PIECE_TYPES.keys()

    @staticmethod
    def all_orientations() -> List[str]:
        return ORIENTATIONS

    # ---- device plumbing ------------------------------------------------------------------
    def _batch(self):
        if self._stepper is None:
            from ...batched import BlokusBatch
            self._stepper = BlokusBatch(1, device=self._device)
        return self._stepper

    def _upload(self, state, mover: int):
        import torch
        board, round_count, players = state
        bb = self._batch()
        cells = np.asarray(board.board_contents)
        occ = np.zeros((1, 4, 20), dtype=np.uint32)
        weights = (np.uint32(1) << np.arange(20, dtype=np.uint32))
        for c in range(4):
            occ[0, c] = ((cells == c + 1).astype(np.uint32) * weights[None, :]).sum(axis=1)
        bb.occ.copy_(torch.from_numpy(occ.view(np.int32)))
        inv = np.array([[sum(1 << PIECE_NAME_TO_INDEX[p] for p in ai.current_pieces) for ai in players]], dtype=np.int32)
        bb.inv.copy_(torch.from_numpy(inv))
        bb.score.copy_(torch.from_numpy(np.array([[ai.player_score for ai in players]], dtype=np.int32)))
        bb.round.fill_(int(round_count))
        bb.to_move.fill_(int(mover))
        return bb

    def _download(self, bb):
        board = Board()
        board.board_contents = bb.board().cpu().numpy().astype(np.int64).reshape(20, 20)
        inv = bb.inv.cpu().numpy().view(np.uint32)[0]
        score = bb.score.cpu().numpy()[0]
        players = []
        for c in range(4):
            ai = AI(board, c + 1)
            ai.player_score = int(score[c])
            ai.current_pieces = [name for i, name in enumerate(A.PIECE_NAMES) if (int(inv[c]) >> i) & 1]
            players.append(ai)
        return board, int(bb.round.cpu().numpy()[0]), players

    def _legal_ids(self, state, player: int) -> np.ndarray:
        """Dense ids of every legal action of `player`, ascending (= reference order)."""
        import torch
        bb = self._upload(state, player)
        count, mask = bb.valid(player=torch.tensor([player], dtype=torch.int8, device=bb.device), want_mask=True)
        bits = np.unpackbits(mask.cpu().numpy().view(np.uint8)[0], bitorder="little")
        ids = np.nonzero(bits)[0]
        assert len(ids) == int(count.cpu().numpy()[0])
        return ids

    # ---- dynamics ---------------------------------------------------------------------------
    def new_state(self, num_players: int = 4) -> State:
        if num_players is None:
            num_players = 4
        assert num_players == 4
        board = Board()
        return (board, 0, [AI(board, c) for c in (1, 2, 3, 4)]), [0]

    @staticmethod
    def serializable() -> bool:
        return True

    @staticmethod
    def serialize_state(state: object) -> bytearray:
        from ... import compat                         # reference class paths when compat.reference_wire_format() is on
        return compat.dumps_blokus_state(state)

    @staticmethod
    def deserialize_state(serialized_state: bytearray) -> State:
        from ... import compat
        return compat.loads_blokus_state(serialized_state)

    def current_rewards(self, state: object) -> List[float]:
        return [p.player_score for p in state[2]]

    def next_state(self, state: object, players: List[int], actions: List[str]):
        """One move (or pass ``''``) of ``players[0]`` (reference :357-451), evaluated by the HIP kernel.

        As in the reference the action is NOT checked for legality here (callers use ``is_valid_action``);
        a piece missing from the mover's inventory raises ``ValueError`` like ``list.remove`` does there.
        """
        import torch
        player_num, action = players[0], actions[0]
        action_id = A.string_to_id(action) if len(action) > 0 else A.PASS
        if action_id >= 0:
            piece_name = A.PIECE_NAMES[action_id // 16000]
            if piece_name not in state[2][player_num].current_pieces:
                raise ValueError("list.remove(x): x not in list")
        bb = self._upload(state, player_num)
        reward, terminal, winners = bb.step(torch.tensor([action_id], dtype=torch.int32, device=bb.device))
        new_state = self._download(bb)
        term = bool(terminal.cpu().numpy()[0])
        wmask = int(winners.cpu().numpy()[0])
        win = [p for p in range(4) if (wmask >> p) & 1] if term else None
        return new_state, [int(bb.to_move.cpu().numpy()[0])], [int(reward.cpu().numpy()[0])], term, win

    def valid_actions(self, state: object, player: int) -> List[str]:
        """Every legal action string in the reference's order, or ``['']`` (reference :453-500)."""
        ids = self._legal_ids(state, player)
        return [A.id_to_string(i) for i in ids] if len(ids) else [""]

    def valid_actions_dict(self, state: object, player: int) -> Dict[str, Dict[Tuple[int, int], List[str]]]:
        """``{piece: {(x, y): [orientation+shift, ...]}}`` (reference :630-665, board.py:183-193)."""
        out: Dict[str, Dict[Tuple[int, int], List[str]]] = {}
        for i in self._legal_ids(state, player):
            piece, x, y, o, k = A.decode(int(i))
            out.setdefault(A.PIECE_NAMES[piece], {}).setdefault((x, y), []).append(ORIENTATIONS[o] + str(k))
        return out

    def player_perspective_valid_actions(self, state: object, player: int) -> List[str]:
        moves = [self.convert_real_action_to_player_perspective_action(a, player) for a in self.valid_actions(state, player)]
        return moves if moves else [""]

    def convert_real_action_to_player_perspective_action(self, action: str, player: int) -> str:
        """Rotate the index about (9.5, 9.5) and the orientation by two steps per player (reference :553-590)."""
        if not action:
            return ""
        piece_type, index, orientation = string_to_action(action)
        index = tuple((np.matmul(BOARD_TO_PLAYER_OBSERVATION_ROTATION_MATRICES[player],
                                 (np.asarray(index) - 9.5)) + 9.5).astype(np.int32))
        name, offset = A.separate_offset_from_orientation(orientation)
        name = ORIENTATIONS[(ORIENTATIONS.index(name) + player * 2) % len(ORIENTATIONS)]
        return action_to_string(piece_type, index, name + offset)

    def convert_player_perspective_action_to_real_action(self, player_action: str, player: int) -> str:
        """Inverse of the above (reference :592-628)."""
        if not player_action:
            return ""
        piece_type, index, orientation = string_to_action(player_action)
        index = tuple((np.matmul(PLAYER_OBSERVATION_TO_BOARD_ROTATION_MATRICES[player],
                                 (np.asarray(index) - 9.5)) + 9.5).astype(np.int32))
        name, offset = A.separate_offset_from_orientation(orientation)
        name = ORIENTATIONS[(ORIENTATIONS.index(name) - player * 2) % len(ORIENTATIONS)]
        return action_to_string(piece_type, index, name + offset)

    def is_valid_action(self, state: object, player: int, action: str) -> bool:
        """Membership in the legal-move set of ``player`` (reference :667-719); ``''`` is never valid."""
        if len(action) == 0:
            return False
        piece_type, index, orientation = string_to_action(action)
        try:
            o = A.ORIENTATION_INDEX[orientation[:-1]]
            wanted = A.encode(PIECE_NAME_TO_INDEX[piece_type], index[0], index[1], o, int(orientation[-1]))
        except (KeyError, ValueError, IndexError):
            return False
        if not (0 <= index[0] < 20 and 0 <= index[1] < 20):
            return False
        return bool(np.isin(wanted, self._legal_ids(state, player)))

    def state_to_observation(self, state: object, player: int) -> Dict[str, np.ndarray]:
        """Relative player ids (-1 empty), board rotated into the observer's viewpoint, inventories as a
        (4, 21) uint8 matrix in relative player order, scores rolled (reference :721-768); evaluated by
        ``crl_blokus_observe``."""
        import torch
        bb = self._upload(state, player)
        obs = bb.observe(torch.tensor([player], dtype=torch.int8, device=bb.device))
        return {"board": obs["board"].cpu().numpy().astype(np.int64).reshape(20, 20),
                "pieces": obs["pieces"].cpu().numpy().reshape(4, 21),
                "score": obs["score"].cpu().numpy().astype(np.int64).reshape(4),
                "player": np.array([player])}
