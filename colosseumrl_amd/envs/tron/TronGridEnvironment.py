"""Tron on a finite grid -- drop-in for ``colosseumrl.envs.tron.TronGridEnvironment``.

Same constructor config string (``"board;players;window;remove_on_death"``),
state tuple ``(board int64[N,N], heads int64[P], directions int64[P],
deaths int64[P])``, action strings and return types as the reference
(colosseumrl/envs/tron/TronGridEnvironment.py:61-508).  The game rules are NOT
evaluated in Python: ``next_state``/``new_state``/``state_to_observation``
write the state into host memory the GPU maps (``colosseumrl_amd.single.SingleTron``) and run the HIP kernels on it --
no copies, one launch and one synchronise per call; without an MI355X they raise.  For throughput use ``colosseumrl_amd.batched.TronBatch``
directly -- this class exists so existing agents and servers drop in unchanged.
"""
from time import time
from typing import Dict, List, Tuple

import numpy as np
from dill import dumps, loads

from ...BaseEnvironment import BaseEnvironment
from . import layout


def create_tron_config(*args) -> str:
    """Join constructor options into the ``a;b;c`` config string (reference :12-25)."""
    return ";".join(str(a) for a in args)


def parse_tron_config(config: str) -> Tuple:
    """Config string -> (board_size, num_players, observation_window, remove_on_death).

    Integers parse as ints, anything else as ``text.lower() == "true"``; missing trailing
    options default to 4, -1, False; the empty string means ``19;4;-1;False`` (reference :28-58).
    """
    if len(config) == 0:
        return 19, 4, -1, False

    def one(text):
        try:
            return int(text)
        except ValueError:
            return text.lower() == "true"

    options = [one(t) for t in config.split(";")]
    for default in ((4, -1, False)[len(options) - 1:] if len(options) < 4 else ()):
        options.append(default)
    return options


class TronGridEnvironment(BaseEnvironment):
    STRING_TO_ACTION = {"": 0, "forward": 0, "right": 1, "left": -1}   # reference :62-67

    @staticmethod
    def create(board_size: int = 19, num_players: int = 4, observation_window: int = -1,
               remove_on_death: bool = False) -> "TronGridEnvironment":
        """Keyword-argument constructor (reference :69-90)."""
        return TronGridEnvironment(create_tron_config(board_size, num_players, observation_window, remove_on_death))

    def __init__(self, config: str = "", device="cuda"):
        super().__init__(config)
        board_size, num_players, observation_window, remove_on_death = parse_tron_config(config)
        self.N = board_size
        self.num_players = num_players
        self.observation_window = observation_window
        self.fully_observable = observation_window < 0
        self.remove_on_death = remove_on_death       # parsed but without effect, as in the reference
        self.player_array = np.arange(num_players)
        self.move_array = ["forward", "right", "left"]
        # persistent move buffer: an alive player missing from `players` replays its last move (:118,297-298)
        self._moves = np.zeros(num_players, dtype=np.int64)
        self._device = device
        self._stepper = None
        self._start_boards = {}                      # spawn layout -> start board (new_state)
        self._observed = None                        # (state key, observations of all players) of the last next_state
        self._staged = None                          # key of the state the staging block holds

    def __repr__(self):
        return ("Tron Finite Grid Environment\n" + "=" * 50 + "\n"
                + "\tSize: {0}x{0}\n".format(self.N)
                + "\tNumber of players: {}\n".format(self.num_players)
                + "\tFully Observable: {}\n".format("Yes" if self.fully_observable else "No")
                + "\tRemove old players: {}\n".format("Yes" if self.remove_on_death else "No")
                + "-" * 50 + "\n")

    __str__ = __repr__

    @property
    def min_players(self) -> int:
        return self.num_players

    @property
    def max_players(self) -> int:
        return self.num_players

    @staticmethod
    def observation_names() -> List[str]:
        return ["board", "heads", "directions", "deaths"]

    @property
    def observation_shape(self) -> Dict[str, tuple]:
        p = (self.num_players,)
        return {"board": (self.N, self.N), "heads": p, "directions": p, "deaths": p}

    # ---- device plumbing ------------------------------------------------------------------
    def _single(self):
        """The single-state HIP stepper behind this instance: host-mapped staging + private stream
        (``colosseumrl_amd.single.SingleTron``; created on first use; raises without a GPU)."""
        if self._stepper is None:
            from ...single import SingleTron
            # (the context's spawn layout only matters to auto-reset, which the single-state calls never ask for)
            self._stepper = SingleTron(self.N, self.num_players, list(range(self.num_players)), [0] * self.num_players)
        return self._stepper

    def _stage(self, state):
        """The stepper with `state` in its staging block.  A state that is already there -- the one the last
        ``next_state`` produced, the usual case in a game loop -- is not written again (compared by value)."""
        st = self._single()
        key = self._key(state)
        if key is None or key != self._staged:
            st.load(*state)
            self._staged = key
        return st

    @staticmethod
    def _key(state):
        """Value identity of a state (bytes of its four arrays), or None for anything that is not four numpy arrays."""
        try:
            return (state[0].tobytes(), state[1].tobytes(), state[2].tobytes(), state[3].tobytes())
        except AttributeError:
            return None

    # ---- dynamics ---------------------------------------------------------------------------
    def generate_start_positions(self, ring_offset: int = 1, spawn_offset=0):
        """(heads, directions) of the spawn ring (reference :183-226).

        A tuple ``spawn_offset`` draws one offset per player from numpy's global generator, as
        the reference does (``np.random.randint(lo, hi)`` per player, :222-224).
        """
        if isinstance(spawn_offset, int):
            spawn_offset = (spawn_offset, spawn_offset + 1)
        offsets = [int(np.random.randint(*spawn_offset)) for _ in range(self.num_players)]
        heads, dirs = layout.start_positions(self.N, self.num_players, ring_offset, offsets)
        return np.asarray(heads, dtype=np.int64), np.asarray(dirs, dtype=np.int64)

    def new_state(self, num_players: int = None, ring_offset: int = 1, spawn_offset=2):
        """Initial state and the acting players (reference :228-263).  The board of a given spawn layout is produced
        once by the GPU's reset kernel (``crl_tron_reset``) and served from a per-instance cache afterwards."""
        num_players = self.num_players if num_players is None else num_players
        assert num_players == self.num_players, "Do not change the number of players from the game configuration."
        np.random.seed(int(time()))                   # reference :255 (observable only with tuple offsets)
        heads, directions = self.generate_start_positions(ring_offset, spawn_offset)
        key = (heads.tobytes(), directions.tobytes())
        board = self._start_boards.get(key)
        if board is None:
            from ...single import SingleTron
            fresh = SingleTron(self.N, self.num_players, heads.tolist(), directions.tolist())
            fresh.reset()
            board = fresh.state64()[0]
            if len(self._start_boards) < 4096:
                self._start_boards[key] = board
        return (board.copy(), heads, directions, np.zeros(self.num_players, dtype=np.int64)), self.player_array

    def next_state(self, state: object, players: List[int], actions: List[str]):
        """One simultaneous move of every listed player (reference :265-323), evaluated on the GPU: one fused launch
        (next_state + the observations of every player of the new state, which ``state_to_observation`` then serves
        without another GPU call), no copies, one synchronise."""
        for player, action in zip(players, actions):
            self._moves[player] = self.STRING_TO_ACTION[action]      # KeyError on an unknown string, like the reference
        st = self._stage(state)
        st.step_observe(self._moves)
        new_state = st.state64()
        v = st.v
        self._staged = self._key(new_state)           # the staging block now holds the new state
        self._observed = (self._staged, v["obs_board"].copy(), v["obs_heads"].copy(), v["obs_dirs"].copy(),
                          v["obs_deaths"].copy())
        new_players = np.where(new_state[3] == 0)[0]
        rewards = v["rewards"].astype(np.int64)
        term = np.bool_(bool(v["terminal"][0]))
        return new_state, new_players, rewards, term, (new_players if term else None)

    def valid_actions(self, state: object, player: int) -> List[str]:
        return self.move_array                        # every move is always allowed (reference :325-341)

    def is_valid_action(self, state: object, player: int, action: str) -> bool:
        return True                                   # reference :343-361

    def state_to_observation(self, state: object, player: int) -> Dict[str, np.ndarray]:
        """Board relabelled so the observer is player 1; per-player vectors rolled (reference :363-420).  For the
        state the last ``next_state`` returned the observation is already there (the fused launch wrote it for
        every player); any other state takes one ``crl_tron_observe`` call."""
        P, N = self.num_players, self.N
        NN = N * N
        seen = self._observed
        # Any integer is an observer in the reference: the vectors roll by (arange + player) % P (:393) and the board goes
        # through C's remainder (CyTronGrid.pyx:71, cdivision=True) -- up to player == P that is observer `player % P`,
        # beyond P low trail ids come out <= 0; crl_tron_observe reproduces both.  int8 on the wire: fold the id into range
        # without changing either result (2P + player % P keeps the sign of every v - (player + 1) + P and the class mod P).
        player = int(player)
        if player < 0:
            player %= P
        elif player >= 2 * P:
            player = 2 * P + player % P
        pl = player % P
        if player <= P and seen is not None and seen[0] == self._key(state):
            board = seen[1][pl * NN:(pl + 1) * NN].astype(np.int64).reshape(N, N)
            heads = seen[2][pl * P:(pl + 1) * P].astype(np.int64)
            directions = seen[3][pl * P:(pl + 1) * P].astype(np.int64)
            deaths = seen[4][pl * P:(pl + 1) * P].astype(np.int64)
        else:
            st = self._stage(state)
            st.observe(player)
            v = st.v
            board = v["obs_board"][:NN].astype(np.int64).reshape(N, N)
            heads = v["obs_heads"][:P].astype(np.int64)
            directions = v["obs_dirs"][:P].astype(np.int64)
            deaths = v["obs_deaths"][:P].astype(np.int64)
        if self.fully_observable:
            return {"board": board, "heads": heads, "directions": directions, "deaths": deaths}
        # unfinished window branch of the reference (:407-420): no 'directions', python slice semantics
        head = heads[0]
        x, y, delta = head % self.N, head // self.N, self.observation_window
        return {"board": board[y - delta:y + delta, x - delta:x + delta], "heads": heads, "deaths": deaths}

    # ---- transport ---------------------------------------------------------------------------
    @staticmethod
    def serializable() -> bool:
        return False                                  # reference :422-431

    @staticmethod
    def serialize_state(state: object) -> bytearray:
        return dumps(state)                           # reference :433-447

    @staticmethod
    def deserialize_state(serialized_state: bytearray) -> object:
        return loads(serialized_state)                # reference :449-463

    # ---- helpers ---------------------------------------------------------------------------
    @staticmethod
    def next_cell(x, y, direction, board_size: int = None):
        """Neighbour of (x, y) in ``direction``; clamped to the board when a size is given (:466-481)."""
        # (int(): `direction` is a numpy scalar when it comes out of an observation, and numpy booleans do not subtract)
        x += int(direction == 1) - int(direction == 3)
        y += int(direction == 2) - int(direction == 0)
        if board_size:
            x = max(min(x, board_size - 1), 0)
            y = max(min(y, board_size - 1), 0)
        return x, y

    def compute_ranking(self, state: object, players: List[int], winners: List[int]) -> Dict[int, int]:
        """Competition ranking by trail length with the mutual-kill tie rule (reference :483-508), on the GPU
        (``crl_tron_ranking``).  Keys are numpy integers ordered by rank like the reference's ``most_common()`` walk."""
        st = self._stage(state)
        ranks = st.ranking().copy()
        order = sorted(range(len(ranks)), key=lambda p: (int(ranks[p]), p))
        return {np.int64(p): int(ranks[p]) for p in order}
