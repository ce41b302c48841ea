"""ctypes/numpy front end of liboracle.so -- TEST INFRASTRUCTURE (the checker, never the product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
All arrays are host numpy arrays in the same SoA layout the HIP kernels use.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TAG_TRON, TAG_TTT, TAG_BLOKUS = 0x54520000, 0x54540000, 0x424C0000


def build(force=False):
    """Build (when stale) and return the library to load: liboracle.so, or the Makefile target named by ORACLE_LIB
    (`liboracle_asan.so`: the ASan + UBSan build tests/test_oracle_asan.py runs the fixtures through)."""
    target = os.path.basename(os.environ.get("ORACLE_LIB") or "liboracle.so")
    if target not in ("liboracle.so", "liboracle_asan.so"):
        raise ValueError("ORACLE_LIB must name a target of oracle/Makefile, got %r" % target)
    so = os.path.join(HERE, target)
    srcs = [os.path.join(HERE, f) for f in ("crl_oracle.c", "blokus_oracle.c", "crl_oracle.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", HERE, "-B", target], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _chk(a, dtype, shape=None):
    assert isinstance(a, np.ndarray) and a.dtype == np.dtype(dtype) and a.flags.c_contiguous, (a.dtype, dtype)
    if shape is not None:
        assert tuple(a.shape) == tuple(shape), (a.shape, shape)
    return a


# ------------------------------------------------------------------ RNG
def philox4x32(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32).copy()
    k = np.asarray(key, dtype=np.uint32).copy()
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32(_p(c), _p(k), _p(out))
    return out


# ------------------------------------------------------------------ Tron
def tron_start_positions(N, P, ring_offset=1, spawn_offset=2):
    offs = np.asarray([spawn_offset] * P if np.isscalar(spawn_offset) else spawn_offset, dtype=np.int32)
    heads = np.zeros(P, dtype=np.int16)
    dirs = np.zeros(P, dtype=np.int8)
    rc = lib().orc_tron_start_positions(C.c_int(N), C.c_int(P), C.c_int(ring_offset), _p(offs), _p(heads), _p(dirs))
    if rc != 0:
        raise ValueError("orc_tron_start_positions rc=%d" % rc)
    return heads, dirs


class TronState:
    """SoA host buffers for B Tron games."""

    def __init__(self, N, P, B):
        self.N, self.P, self.B = N, P, B
        self.board = np.zeros((B, N * N), dtype=np.int8)
        self.heads = np.zeros((P, B), dtype=np.int16)
        self.dirs = np.zeros((P, B), dtype=np.int8)
        self.deaths = np.zeros((P, B), dtype=np.int8)
        # rollout bookkeeping
        self.tcount = np.zeros(B, dtype=np.uint32)
        self.tstep = np.zeros(B, dtype=np.uint32)
        self.n_episodes = np.zeros(B, dtype=np.uint32)
        self.win_count = np.zeros((P, B), dtype=np.uint32)
        self.len_sum = np.zeros(B, dtype=np.uint32)
        self.ret_sum = np.zeros((P, B), dtype=np.int32)
        self.last_winners = np.zeros(B, dtype=np.uint8)
        self.last_len = np.zeros(B, dtype=np.uint16)

    def copy(self):
        o = TronState(self.N, self.P, self.B)
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                setattr(o, k, v.copy())
        return o


class _TronStats(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("tcount", "tstep", "n_episodes", "win_count", "len_sum", "ret_sum", "last_winners", "last_len")]


def tron_reset(st, start_heads, start_dirs):
    sh = _chk(np.ascontiguousarray(start_heads, dtype=np.int16), np.int16, (st.P,))
    sd = _chk(np.ascontiguousarray(start_dirs, dtype=np.int8), np.int8, (st.P,))
    lib().orc_tron_reset(C.c_int(st.N), C.c_int(st.P), C.c_int64(st.B), _p(sh), _p(sd),
                         _p(st.board), _p(st.heads), _p(st.dirs), _p(st.deaths))


def tron_step(st, actions):
    a = _chk(np.ascontiguousarray(actions, dtype=np.int8), np.int8, (st.P, st.B))
    rewards = np.zeros((st.P, st.B), dtype=np.int8)
    terminal = np.zeros(st.B, dtype=np.uint8)
    winners = np.zeros(st.B, dtype=np.uint8)
    lib().orc_tron_step(C.c_int(st.N), C.c_int(st.P), C.c_int64(st.B),
                        _p(st.board), _p(st.heads), _p(st.dirs), _p(st.deaths),
                        _p(a), _p(rewards), _p(terminal), _p(winners))
    return rewards, terminal, winners


def tron_rollout(st, seed, first_env_id, T, start_heads, start_dirs, n_threads=1):
    sh = np.ascontiguousarray(start_heads, dtype=np.int16)
    sd = np.ascontiguousarray(start_dirs, dtype=np.int8)
    stats = _TronStats(*[_p(getattr(st, n)) for n, _ in _TronStats._fields_])
    f = lib().orc_tron_rollout
    f.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_uint64, C.c_uint64, C.c_int,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _TronStats, C.c_int]
    f.restype = None
    f(st.N, st.P, st.B, seed, first_env_id, T, _p(sh), _p(sd),
      _p(st.board), _p(st.heads), _p(st.dirs), _p(st.deaths), stats, n_threads)


def tron_observe(st, player):
    pl = _chk(np.ascontiguousarray(player, dtype=np.int8), np.int8, (st.B,))
    ob = np.zeros_like(st.board)
    oh = np.zeros_like(st.heads)
    od = np.zeros_like(st.dirs)
    ok = np.zeros_like(st.deaths)
    lib().orc_tron_observe(C.c_int(st.N), C.c_int(st.P), C.c_int64(st.B), _p(st.board), _p(st.heads),
                           _p(st.dirs), _p(st.deaths), _p(pl), _p(ob), _p(oh), _p(od), _p(ok))
    return ob, oh, od, ok


def tron_ranking(N, P, board, deaths):
    """board int8 [B, N*N], deaths int8 [P, B] -> rank int8 [P, B] (0 = best)"""
    board = _chk(np.ascontiguousarray(board, dtype=np.int8), np.int8)
    deaths = _chk(np.ascontiguousarray(deaths, dtype=np.int8), np.int8)
    B = board.shape[0]
    rank = np.zeros((P, B), dtype=np.int8)
    lib().orc_tron_ranking(C.c_int(N), C.c_int(P), C.c_int64(B), _p(board), _p(deaths), _p(rank))
    return rank


# ------------------------------------------------------------------ TicTacToe
def ttt_lines(D0, D1, D2, K):
    buf = np.zeros(256, dtype=np.uint32)
    n = lib().orc_ttt_lines(C.c_int(D0), C.c_int(D1), C.c_int(D2), C.c_int(K), _p(buf))
    if n < 0:
        raise ValueError("orc_ttt_lines rc=%d" % n)
    return buf[:n].copy()


class TTTState:
    def __init__(self, dims, K, P, B):
        self.dims, self.K, self.P, self.B = tuple(dims), K, P, B
        d = (1,) * (3 - len(dims)) + tuple(dims)
        self.n_cells = int(np.prod(d))
        self.lines = ttt_lines(d[0], d[1], d[2], K)
        self.occ = np.zeros((P, B), dtype=np.uint32)
        self.winner = np.full(B, -1, dtype=np.int8)
        self.to_move = np.zeros(B, dtype=np.int8)
        self.tcount = np.zeros(B, dtype=np.uint32)
        self.tstep = np.zeros(B, dtype=np.uint32)
        self.n_episodes = np.zeros(B, dtype=np.uint32)
        self.win_count = np.zeros((P, B), dtype=np.uint32)
        self.draw_count = np.zeros(B, dtype=np.uint32)
        self.len_sum = np.zeros(B, dtype=np.uint32)

    def board(self):
        """int8 [B, n_cells] board in the reference encoding (-1 empty, else player id)."""
        bd = np.full((self.B, self.n_cells), -1, dtype=np.int8)
        for p in range(self.P):
            bits = (self.occ[p][:, None] >> np.arange(self.n_cells, dtype=np.uint32)[None, :]) & 1
            bd[bits.astype(bool)] = p
        return bd


class _TTTStats(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("tcount", "tstep", "n_episodes", "win_count", "draw_count", "len_sum")]


def ttt_step(st, action):
    a = _chk(np.ascontiguousarray(action, dtype=np.int8), np.int8, (st.B,))
    reward = np.zeros(st.B, dtype=np.int8)
    terminal = np.zeros(st.B, dtype=np.uint8)
    winners = np.zeros(st.B, dtype=np.int8)
    lib().orc_ttt_step(C.c_int(st.n_cells), C.c_int(st.P), C.c_int(len(st.lines)), _p(st.lines), C.c_int64(st.B),
                       _p(st.occ), _p(st.winner), _p(st.to_move), _p(a), _p(reward), _p(terminal), _p(winners))
    return reward, terminal, winners


def ttt_rollout(st, seed, first_env_id, T, n_threads=1):
    stats = _TTTStats(*[_p(getattr(st, n)) for n, _ in _TTTStats._fields_])
    f = lib().orc_ttt_rollout
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_int,
                  C.c_void_p, C.c_void_p, C.c_void_p, _TTTStats, C.c_int]
    f.restype = None
    f(st.n_cells, st.P, len(st.lines), _p(st.lines), st.B, seed, first_env_id, T,
      _p(st.occ), _p(st.winner), _p(st.to_move), stats, n_threads)


# ------------------------------------------------------------------ Blokus
BLOKUS_PIECES = ["monomino1", "domino1", "trominoe1", "trominoe2", "tetrominoes1", "tetrominoes2", "tetrominoes3",
                 "tetrominoes4", "tetrominoes5", "pentominoe1", "pentominoe2", "pentominoe3", "pentominoe4",
                 "pentominoe5", "pentominoe6", "pentominoe7", "pentominoe8", "pentominoe9", "pentominoe10",
                 "pentominoe11", "pentominoe12"]
BLOKUS_ORIENTATIONS = ["north", "northeast", "east", "southeast", "south", "southwest", "west", "northwest"]


def blokus_encode(piece, x, y, orient, shift):
    return ((piece * 400 + y * 20 + x) * 8 + orient) * 5 + shift


def blokus_decode(aid):
    shift, o, cell, piece = aid % 5, (aid // 5) % 8, (aid // 40) % 400, aid // 16000
    return piece, cell % 20, cell // 20, o, shift


def blokus_action_string(aid):
    if aid < 0:
        return ""
    piece, x, y, o, shift = blokus_decode(int(aid))
    return "%s;(%d, %d);%s%d" % (BLOKUS_PIECES[piece], x, y, BLOKUS_ORIENTATIONS[o], shift)


def blokus_action_id(s):
    if s == "":
        return -1
    name, idx, orient = s.split(";")
    x, y = [int(v) for v in idx.replace("(", "").replace(")", "").split(",")]
    return blokus_encode(BLOKUS_PIECES.index(name), x, y, BLOKUS_ORIENTATIONS.index(orient[:-1]), int(orient[-1]))


BLOKUS_EXT_BASE = 336000          # ids from here on: index anywhere in [-20, 20) x [-20, 20) (blokus_oracle.c decode_action)
BLOKUS_STATUS = {0: None, -1: IndexError, -2: ValueError, -3: KeyError}


def blokus_step_action_id(s):
    """What the reference's next_state does with an action string BEFORE it touches the board, exceptions in its order:
    string_to_action (BlokusEnvironment.py:83-106: split / int -> ValueError), PIECE_TYPES[piece] (board.py:93: KeyError),
    int(orientation[-1]) (IndexError on '', ValueError on a non-digit), an orientation name that is none of the eight falls
    to rotate_piece's default branch = east (computation.py:85-86).  Returns the id orc_blokus_step takes; an index outside
    [-20, 20) cannot be encoded and raises the IndexError numpy would (the piece's index cell itself is always written)."""
    if s == "":
        return -1
    name, idx, orient = s.split(";")
    xy = tuple(map(int, idx.replace("(", "").replace(")", "").split(",")))
    piece = {n: i for i, n in enumerate(BLOKUS_PIECES)}[name]
    shift = int(orient[-1])
    o = BLOKUS_ORIENTATIONS.index(orient[:-1]) if orient[:-1] in BLOKUS_ORIENTATIONS else 2
    if shift >= lib().orc_blokus_piece_cells(C.c_int(piece)):
        raise IndexError("shift id names no cell of the piece")
    x, y = xy[0], xy[1]
    if not (-20 <= x < 20 and -20 <= y < 20):
        raise IndexError("index out of bounds")
    if 0 <= x < 20 and 0 <= y < 20:
        return blokus_encode(piece, x, y, o, shift)
    return BLOKUS_EXT_BASE + ((piece * 1600 + (y + 20) * 40 + (x + 20)) * 8 + o) * 5 + shift


def blokus_placement(piece, orient, shift):
    cells = np.zeros((5, 2), dtype=np.int8)
    lib().orc_blokus_placement(C.c_int(piece), C.c_int(orient), C.c_int(shift), _p(cells))
    return cells[:lib().orc_blokus_piece_cells(C.c_int(piece))]


class BlokusState:
    def __init__(self, B):
        self.B = B
        self.occ = np.zeros((B, 4, 20), dtype=np.uint32)
        self.inv = np.zeros((B, 4), dtype=np.uint32)
        self.score = np.zeros((B, 4), dtype=np.int32)
        self.round = np.zeros(B, dtype=np.int32)
        self.to_move = np.zeros(B, dtype=np.int32)
        self.tcount = np.zeros(B, dtype=np.uint32)
        self.tstep = np.zeros(B, dtype=np.uint32)
        self.n_episodes = np.zeros(B, dtype=np.uint32)
        self.win_count = np.zeros((4, B), dtype=np.uint32)
        self.len_sum = np.zeros(B, dtype=np.uint32)
        self.score_sum = np.zeros((4, B), dtype=np.int32)
        blokus_reset(self)

    @property
    def board(self):
        out = np.zeros((self.B, 20, 20), dtype=np.int8)
        lib().orc_blokus_board(C.c_int64(self.B), _p(self.occ), _p(out))
        return out

    def set_board(self, board):
        """board int [B,20,20] with 0 empty / colour 1..4 -> bitboards"""
        board = np.asarray(board)
        self.occ[:] = 0
        for c in range(4):
            bits = (board == c + 1).astype(np.uint32) << np.arange(20, dtype=np.uint32)[None, None, :]
            self.occ[:, c, :] = bits.sum(axis=2).astype(np.uint32)


def blokus_observe(st, player):
    pl = _chk(np.ascontiguousarray(player, dtype=np.int8), np.int8, (st.B,))
    ob = np.zeros((st.B, 20, 20), dtype=np.int8)
    op = np.zeros((st.B, 4, 21), dtype=np.uint8)
    osc = np.zeros((st.B, 4), dtype=np.int32)
    lib().orc_blokus_observe(C.c_int64(st.B), _p(st.occ), _p(st.inv), _p(st.score), _p(pl), _p(ob), _p(op), _p(osc))
    return ob, op, osc


class _BlokusStats(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("tcount", "tstep", "n_episodes", "win_count", "len_sum", "score_sum")]


def blokus_reset(st):
    lib().orc_blokus_reset(C.c_int64(st.B), _p(st.occ), _p(st.inv), _p(st.score), _p(st.round), _p(st.to_move))


def blokus_valid(st, player=None, cap=0, n_threads=1):
    """(count[B], ids[B, cap] or None) for the player to move (or player[b])."""
    count = np.zeros(st.B, dtype=np.int32)
    ids = np.full((st.B, cap), -1, dtype=np.int32) if cap > 0 else None
    pl = None if player is None else np.ascontiguousarray(player, dtype=np.int8)
    lib().orc_blokus_valid(C.c_int64(st.B), _p(st.occ), _p(st.inv), _p(st.score), _p(st.round), _p(st.to_move),
                           None if pl is None else _p(pl), _p(count), None if ids is None else _p(ids),
                           C.c_int(cap), C.c_int(n_threads))
    return count, ids


def blokus_step(st, action, n_threads=1):
    a = _chk(np.ascontiguousarray(action, dtype=np.int32), np.int32, (st.B,))
    reward = np.zeros(st.B, dtype=np.int8)
    terminal = np.zeros(st.B, dtype=np.uint8)
    winners = np.zeros(st.B, dtype=np.uint8)
    lib().orc_blokus_step(C.c_int64(st.B), _p(st.occ), _p(st.inv), _p(st.score), _p(st.round), _p(st.to_move),
                          _p(a), _p(reward), _p(terminal), _p(winners), C.c_int(n_threads))
    return reward, terminal, winners


def blokus_valid_corner_grid(board):
    """Board.check_valid_corner on every (colour, row, col) of int8 boards [K][20][20] -> uint8 [K][4][20][20]."""
    b = np.ascontiguousarray(board, dtype=np.int8).reshape(-1, 20, 20)
    grid = np.zeros((b.shape[0], 4, 20, 20), dtype=np.uint8)
    lib().orc_blokus_valid_corner_grid(C.c_int64(b.shape[0]), _p(b), _p(grid))
    return grid


def blokus_placement_tests(reset=False):
    """Reference-equivalent placement tests counted by the oracle since the last reset (see blokus_oracle.c)."""
    f = lib().orc_blokus_placement_tests
    f.restype = C.c_uint64
    return int(f(C.c_int(1 if reset else 0)))


def blokus_rollout(st, seed, first_env_id, T, n_threads=1):
    stats = _BlokusStats(*[_p(getattr(st, n)) for n, _ in _BlokusStats._fields_])
    f = lib().orc_blokus_rollout
    f.argtypes = [C.c_int64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                  _BlokusStats, C.c_int]
    f.restype = None
    f(st.B, seed, first_env_id, T, _p(st.occ), _p(st.inv), _p(st.score), _p(st.round), _p(st.to_move), stats, n_threads)
