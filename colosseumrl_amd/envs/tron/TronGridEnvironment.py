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
        self._observed = None                        # (value identity of the state, observations of all players) of the last next_state
        self._obs_idle = 0                           # next_state calls since somebody last asked for an observation
        self._staged = None                          # key of the state the int8 staging block holds (compute_ranking)

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

    OBS_IDLE_STEPS = 8    # next_state calls in a row whose fused observations nobody asked for before the launch stops writing them

    def next_state(self, state: object, players: List[int], actions: List[str]):
        """One simultaneous move of every listed player (reference :265-323), evaluated on the GPU through the Cython
        module's own signature (``crl_tron_next_state_inplace64``: the reference's int64 arrays, stepped in place in
        host-mapped memory -- no conversion of the state either way): the four arrays are copied where the GPU reads them
        (the reference copies them too, :301-304), ONE launch steps them and writes rewards / terminal -- and, while
        somebody consumes them, the observations of every player of the new state, which ``state_to_observation`` then
        serves without another GPU call (a match loop asks for all P of them per tick, match_server.py:218) -- one wait."""
        for player, action in zip(players, actions):
            self._moves[player] = self.STRING_TO_ACTION[action]      # KeyError on an unknown string, like the reference
        board, heads, directions, deaths = state
        st = self._single()
        fuse = self._obs_idle < self.OBS_IDLE_STEPS
        st.next_state64(board, heads, directions, deaths, self._moves, fuse)
        v = st.s64
        new_board, new_heads = v["board"].copy(), v["heads"].copy()
        new_directions, new_deaths = v["dirs"].copy(), v["deaths"].copy()
        if fuse:
            self._obs_idle += 1
            self._observed = (v["all"].tobytes(), v["obs"].copy())   # value identity of the new state, its P observations
        else:
            self._observed = None
        new_players = np.where(new_deaths == 0)[0]
        rewards = v["rewards"].copy()
        term = np.bool_(v["terminal"][0] != 0)
        return (new_board, new_heads, new_directions, new_deaths), new_players, rewards, term, (new_players if term else None)

    def valid_actions(self, state: object, player: int) -> List[str]:
        return self.move_array                        # every move is always allowed (reference :325-341)

    def is_valid_action(self, state: object, player: int, action: str) -> bool:
        return True                                   # reference :343-361

    def state_to_observation(self, state: object, player: int) -> Dict[str, np.ndarray]:
        """Board relabelled so the observer is player 1; per-player vectors rolled (reference :363-420).  For the
        state the last ``next_state`` returned the observation is already there (the fused launch wrote it for
        every player); any other state takes one ``crl_tron_relative_player_inplace64`` call (the Cython function the
        reference calls here, :390) and numpy's roll of the three vectors (:393-397)."""
        P, N = self.num_players, self.N
        NN = N * N
        seen = self._observed
        # Any integer is an observer in the reference: the vectors roll by (arange + player) % P with numpy's modulo (:393),
        # the board goes through C's remainder (CyTronGrid.pyx:71, cdivision=True) -- for 0 <= player <= P both are
        # observer `player % P`, which the fused launch of next_state has written; everything else takes the Cython
        # function's own arithmetic on the GPU (crl_tron_relative_player_inplace64) and numpy's roll here.
        try:
            hit = seen is not None and 0 <= player <= P and seen[0] == b"".join(a.tobytes() for a in state)
        except (AttributeError, TypeError):
            hit = False
        if hit:
            self._obs_idle = 0
            pl, ob = int(player) % P, seen[1]
            board = ob[pl * NN:(pl + 1) * NN].reshape(N, N).copy()
            o = P * NN + pl * P
            heads, directions, deaths = ob[o:o + P].copy(), ob[o + P * P:o + P * P + P].copy(), ob[o + 2 * P * P:o + 2 * P * P + P].copy()
        else:
            self._obs_idle = 0                         # somebody wants observations: the next next_state fuses them again
            board = self._single().relative_board64(state[0], player + 1)
            rolled_idx = (np.arange(P) + player) % P
            heads, directions, deaths = state[1][rolled_idx], state[2][rolled_idx], state[3][rolled_idx]
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
