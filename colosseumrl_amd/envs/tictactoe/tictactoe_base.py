"""Shared host logic of the n-player TicTacToe drop-ins.

The three reference envs (colosseumrl/envs/tictactoe/tictactoe_{2,3,4}p_env.py) are the
same program with a different board shape and player count; here they are one
class parameterised by ``SHAPE``/``PLAYERS``/``K`` whose rules run on the GPU through
``colosseumrl_amd.single.SingleTTT`` (the reference's own board layout in host memory the GPU maps; one launch and one
synchronise per call, no copies).

State: ``(board int8[SHAPE] with -1 = empty, winner: None | int)`` as in the reference
(tictactoe_2p_env.py:165-169).  Canonical action string: ``str(tuple_of_python_ints)``,
e.g. ``'(0, 1)'`` -- what the reference emitted under numpy 1.x and what its own
``string_to_action`` parses.
"""
from typing import Dict, List, Tuple, Union

import dill
import numpy as np

from ...BaseEnvironment import BaseEnvironment

State = object


def action_to_string(index) -> str:
    """Cell index tuple -> action string (reference 2p:30-42), with plain Python ints."""
    return str(tuple(int(i) for i in index))


def string_to_action(action_str: str) -> Union[Tuple[int, ...], None]:
    """Action string -> index tuple; ``''`` -> None (reference 2p:45-63)."""
    if action_str == "":
        return None
    return tuple(map(int, action_str.replace("(", "").replace(")", "").split(",")))


class TicTacToeEnvBase(BaseEnvironment):
    SHAPE: Tuple[int, ...] = (3, 3)
    PLAYERS: int = 2
    K: int = 3
    REL_MOD: int = 2          # modulus of _relative_player_id (4p uses 3 in the reference, 4p:50)

    def __init__(self, config: str = "", device="cuda"):
        super().__init__(config)
        self._device = device
        self._stepper = None
        self._seen = None      # (state key, empties mask, mover, observation) of the state the last next_state returned
        self._staged = None    # key of the state the staging block holds
        self._n_cells = int(np.prod(self.SHAPE))
        self._cell_strings = [action_to_string(np.unravel_index(c, self.SHAPE)) for c in range(self._n_cells)]

    @property
    def min_players(self) -> int:
        return self.PLAYERS

    @property
    def max_players(self) -> int:
        return self.PLAYERS

    @property
    def observation_shape(self) -> Dict[str, Tuple[int, ...]]:
        return {"board": tuple(self.SHAPE)}

    @staticmethod
    def observation_names():
        return ["board"]

    # ---- device plumbing ------------------------------------------------------------------
    def _single(self):
        """The single-state HIP stepper behind this instance (``colosseumrl_amd.single.SingleTTT``: host-mapped staging,
        private stream; created on first use; raises without a GPU)."""
        if self._stepper is None:
            from ...single import SingleTTT
            self._stepper = SingleTTT(self.SHAPE, self.K, self.PLAYERS, self.REL_MOD)
        return self._stepper

    @staticmethod
    def _key(state):
        try:
            return (state[0].tobytes(), state[1])
        except AttributeError:
            return None

    def _stage(self, state, mover: int):
        """The stepper with `state` (and `mover`) in its staging block; a state that is already there -- what the last
        ``next_state`` produced -- is not written again (compared by value)."""
        st = self._single()
        key = self._key(state)
        if key is None or key != self._staged:
            st.load(state[0], state[1], mover)
            self._staged = key
        else:
            st.v["to_move"][0] = mover
        return st

    def _valid_mask(self, state) -> int:
        """Empties bit mask of ``state`` (computed on the GPU; for the state the last ``next_state`` returned the fused
        launch has already produced it)."""
        seen = self._seen
        if seen is not None and seen[0] == self._key(state):
            return seen[1]
        return self._stage(state, 0).valid()

    def _cell_of(self, index) -> int:
        """Flat cell of an index tuple with Python indexing rules (negative wraps, out of range raises)."""
        if len(index) != len(self.SHAPE):
            raise IndexError("too many indices for array" if len(index) > len(self.SHAPE) else "index has too few dimensions")
        fixed = []
        for i, n in zip(index, self.SHAPE):
            if i < -n or i >= n:
                raise IndexError("index {} is out of bounds for axis with size {}".format(i, n))
            fixed.append(i + n if i < 0 else i)
        return int(np.ravel_multi_index(tuple(fixed), self.SHAPE))

    # ---- dynamics ---------------------------------------------------------------------------
    def new_state(self, num_players: int = None) -> Tuple[State, List[int]]:
        if num_players is None:
            num_players = self.PLAYERS
        assert num_players == self.PLAYERS
        return (np.full(self.SHAPE, -1, np.int8), None), [0]

    @staticmethod
    def serializable() -> bool:
        return True

    @staticmethod
    def serialize_state(state: object) -> bytearray:
        return dill.dumps(state)

    @staticmethod
    def deserialize_state(serialized_state: bytearray) -> State:
        return dill.loads(serialized_state)

    def current_rewards(self, state: object) -> List[float]:
        """+1 for the winner, -1 for the others, 0 while undecided (reference 2p:219-238)."""
        _, winner = state
        if winner is not None:
            return [1 if p == winner else -1 for p in range(self.max_players)]
        return [0 for _ in range(self.max_players)]

    def next_state(self, state: object, players: List[int], actions: List[str]):
        """One move of ``players[0]`` (reference 2p:240-315), evaluated by the HIP kernel: one launch on host-mapped
        memory (``crl_ttt_step_board``: the move, the win test, and -- for the new state -- the empties mask and the
        next mover's observation, which ``valid_actions`` / ``state_to_observation`` then serve without a GPU call)."""
        action, player_num = actions[0], players[0]
        cell = -1
        if len(action) > 0:
            cell = self._cell_of(string_to_action(action))     # same exceptions as the reference's board[index]
        st = self._stage(state, player_num)
        st.step(cell)
        v = st.v
        board = v["board"].copy().reshape(self.SHAPE)
        w, ws, mover = int(v["winner"][0]), int(v["winners"][0]), int(v["to_move"][0])
        new_state = (board, None if w < 0 else w)
        self._staged = self._key(new_state)
        self._seen = (self._staged, int(v["valid"][0]), mover, v["obs_board"].copy())
        return (new_state, [mover], [int(v["reward"][0])], bool(v["terminal"][0]), None if ws < 0 else [ws])

    def valid_actions(self, state: object, player: int) -> List[str]:
        """Every empty cell in row-major order, or ``['']`` when the board is full (reference 2p:317-348)."""
        mask = self._valid_mask(state)
        out = [self._cell_strings[c] for c in range(self._n_cells) if (mask >> c) & 1]
        return out if out else [""]

    def is_valid_action(self, state: object, player_num: int, action: str) -> bool:
        """Target cell is empty; ``''`` is never valid (reference 2p:350-380)."""
        if len(action) == 0:
            return False
        cell = self._cell_of(string_to_action(action))
        return bool((self._valid_mask(state) >> cell) & 1)

    def state_to_observation(self, state: object, player: int) -> Dict[str, np.ndarray]:
        """Board with ids relative to the observer (reference 2p:382-407; modulus per file, see REL_MOD)."""
        seen = self._seen
        if seen is not None and seen[2] == player and seen[0] == self._key(state):
            return {"board": seen[3].copy().reshape(self.SHAPE)}
        st = self._stage(state, 0)
        st.observe(player)
        return {"board": st.v["obs_board"].copy().reshape(self.SHAPE)}
