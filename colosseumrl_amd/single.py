"""Single-state steppers on host-mapped memory: what the drop-in ``BaseEnvironment`` classes run.

The reference's API is one state per call (``match_server.py:193,201-203,218``, ``ClientEnvironment.py:327-328``).
A B = 1 call through device tensors costs a dozen blocking copies (round 2: ~165 us per ``next_state``); here every
env instance owns ONE block of page-locked host memory that the GPU maps (``crl_host_alloc``) and a private stream:
the call writes the reference-layout state into numpy views of that block, the HIP kernels read it and write their
results there over PCIe, and one ``crl_stream_synchronize`` ends the call -- no hipMemcpy, one blocking operation.
The kernels are the very ones of the batched steppers (same C ABI, B = 1).  No torch tensors are involved.

No CPU path: constructing a stepper without a visible MI355X raises.
"""
import ctypes as C
from typing import Dict, Sequence, Tuple

import numpy as np

from . import _native
from ._native import check


class HostBlob:
    """`crl_host_alloc` memory carved into named numpy views (``.v[name]``) with their device addresses (``.d[name]``)."""

    def __init__(self, lib, fields: Sequence[Tuple[str, type, int]]):
        self._lib = lib
        offs, off = {}, 0
        for name, dtype, count in fields:
            off = (off + 15) & ~15
            offs[name] = (off, np.dtype(dtype), int(count))
            off += np.dtype(dtype).itemsize * int(count)
        self.nbytes = max(off, 16)
        host, dev = C.c_void_p(), C.c_void_p()
        check(lib.crl_host_alloc(self.nbytes, C.byref(host), C.byref(dev)), "crl_host_alloc")
        self._host = host
        raw = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(host.value))
        self.v: Dict[str, np.ndarray] = {}
        self.d: Dict[str, C.c_void_p] = {}
        for name, (o, dt, n) in offs.items():
            self.v[name] = raw[o:o + dt.itemsize * n].view(dt)
            self.d[name] = C.c_void_p(dev.value + o)

    def __del__(self):
        try:
            if self._host:
                self.v.clear()
                self._lib.crl_host_free(self._host)
                self._host = None
        except Exception:  # interpreter shutdown
            pass


class _Single:
    """Context + stream + blob of one env instance."""

    def __init__(self):
        self._lib = _native.require_gpu()
        self._handle = C.c_void_p()
        self._stream = C.c_void_p()

    WAIT_TIMEOUT_S = 5.0     # after this long without the flag the wait falls back to hipStreamSynchronize (and its error)

    def _open_stream(self):
        check(self._lib.crl_stream_create(C.byref(self._stream)), "crl_stream_create")
        # the word the end-of-call wait spins on: a block of its own, so that re-binding the data block never moves it
        self._flag = HostBlob(self._lib, [("seq", np.uint32, 1)])
        self._flag_args = (self._stream, self._flag.d["seq"], C.c_void_p(self._flag.v["seq"].ctypes.data))
        self._seq = 0

    def sync(self):
        """End of a call: everything queued on this instance's stream has run and its results are in the mapped block.
        The wait is on mapped memory (``crl_stream_wait_mapped``: a one-thread kernel behind the chain publishes a sequence
        number, the host spins on it) -- ~3.5 us less than ``hipStreamSynchronize`` for these one-to-three-launch chains."""
        self._seq = seq = (self._seq + 1) & 0xFFFFFFFF or 1
        rc = self._lib.crl_stream_wait_mapped(*self._flag_args, seq, self.WAIT_TIMEOUT_S)
        if rc:
            check(rc, "crl_stream_wait_mapped")

    def __del__(self):
        try:
            if self._stream:
                self._lib.crl_stream_destroy(self._stream)
                self._stream = None
            if self._handle:
                self._lib.crl_destroy(self._handle)
                self._handle = None
        except Exception:  # interpreter shutdown
            pass


class SingleTron(_Single):
    """One Tron game on host-mapped memory, in two blocks: the reference's own layout (int64 board [N*N], heads / directions /
    deaths [P]: `s64`, stepped in place by ``crl_tron_next_state_inplace64_host`` -- what ``next_state`` and
    ``state_to_observation`` of the drop-in class run on) and the batched steppers' int8 / int16 layout at B = 1 (`v`: what
    ``new_state`` (``crl_tron_reset``) and ``compute_ranking`` (``crl_tron_ranking``) run on)."""

    def __init__(self, board_size: int, num_players: int, start_heads: Sequence[int], start_dirs: Sequence[int]):
        super().__init__()
        lib, N, P = self._lib, int(board_size), int(num_players)
        self.N, self.P, self.NN = N, P, N * N
        sh = (C.c_int16 * P)(*[int(h) for h in start_heads])
        sd = (C.c_int8 * P)(*[int(d) for d in start_dirs])
        check(lib.crl_tron_create(N, P, sh, sd, C.byref(self._handle)), "crl_tron_create")
        self._open_stream()
        NN = self.NN
        self.blob = HostBlob(lib, [("board", np.int8, NN), ("heads", np.int16, P), ("dirs", np.int8, P), ("deaths", np.int8, P),
                                   ("rank", np.int8, P)])
        v, d = self.blob.v, self.blob.d
        self.v = v
        h, s = self._handle, self._stream
        # argument tuples bound once (ctypes converts them per call; the pointers never change)
        self._a_rank = (h, 1, d["board"], d["deaths"], d["rank"], s)
        self._a_reset = (h, 1, None, d["board"], d["heads"], d["dirs"], d["deaths"], s)
        self._open64()

    def load(self, board, heads, dirs, deaths):
        v = self.v
        np.copyto(v["board"], np.asarray(board).reshape(-1), casting="unsafe")
        np.copyto(v["heads"], heads, casting="unsafe")
        np.copyto(v["dirs"], dirs, casting="unsafe")
        np.copyto(v["deaths"], deaths, casting="unsafe")

    def state64(self):
        """The blob's state as the reference's int64 arrays."""
        v = self.v
        return (v["board"].astype(np.int64).reshape(self.N, self.N), v["heads"].astype(np.int64),
                v["dirs"].astype(np.int64), v["deaths"].astype(np.int64))

    # ---- the reference's own layout: int64 arrays, stepped in place (crl_tron_next_state_inplace64) ----------------
    def _open64(self):
        """A second mapped block in the REFERENCE's layout: `state` = board [N*N] | heads [P] | directions [P] | deaths [P]
        as one run of int64 (so a state's value identity is one ``tobytes``), the actions, the outputs of the step and the
        observations of all P players of the new state, also one run: P boards | P x P heads | directions | deaths."""
        N, P, NN = self.N, self.P, self.NN
        self.b64 = b = HostBlob(self._lib, [("state", np.int64, NN + 3 * P), ("actions", np.int64, P), ("rewards", np.int64, P),
                                            ("obs", np.int64, P * NN + 3 * P * P), ("player", np.int64, 1),
                                            ("terminal", np.uint8, 1), ("winners", np.uint8, 1)])
        st, ob = b.v["state"], b.v["obs"]
        self.s64 = {"all": st, "board": st[:NN].reshape(N, N), "heads": st[NN:NN + P], "dirs": st[NN + P:NN + 2 * P],
                    "deaths": st[NN + 2 * P:], "actions": b.v["actions"], "rewards": b.v["rewards"], "obs": ob,
                    "terminal": b.v["terminal"], "player": b.v["player"]}
        base, obase = b.d["state"].value, b.d["obs"].value
        at = lambda off: C.c_void_p(base + 8 * off)                                                   # noqa: E731
        oat = lambda off: C.c_void_p(obase + 8 * off)                                                 # noqa: E731
        state = (at(0), at(NN), at(NN + P), at(NN + 2 * P), b.d["actions"], b.d["rewards"], b.d["terminal"], b.d["winners"])
        h, s = self._handle, self._stream
        self._a_next64 = (h, 1) + state + (None, None, None, None, s)
        self._a_next64_obs = (h, 1) + state + (oat(0), oat(P * NN), oat(P * NN + P * P), oat(P * NN + 2 * P * P), s)
        self._a_rel64 = (h, 1, oat(0), P, b.d["player"], s)
        # the one-call form (crl_tron_next_state_inplace64_host): only where the mapped blocks have ONE address for host and GPU
        self._unified = (b.d["state"].value == st.ctypes.data and self._flag.d["seq"].value == self._flag.v["seq"].ctypes.data)
        tail = (s, self._flag.d["seq"])
        self._a_host64 = (h,) + state + (None, None, None, None) + tail
        self._a_host64_obs = (h,) + state + (oat(0), oat(P * NN), oat(P * NN + P * P), oat(P * NN + 2 * P * P)) + tail

    def next_state64(self, board, heads, dirs, deaths, actions, with_obs: bool):
        """``CyTronGrid.next_state_inplace`` on the four int64 arrays (copied into the mapped block as the reference copies
        them, TronGridEnvironment.py:301-304) + rewards / terminal; with `with_obs` also the observations of all P players
        of the new state.  ONE launch, one wait; the results are the views in ``self.s64``."""
        v = self.s64
        v["board"][...] = board                                  # (slice assignment: numpy's unsafe cast, 0.6 us less than five copyto calls)
        v["heads"][:] = heads
        v["dirs"][:] = dirs
        v["deaths"][:] = deaths
        v["actions"][:] = actions
        if self._unified:                                        # launch + completion in one call, vectors by value
            self._seq = seq = (self._seq + 1) & 0xFFFFFFFF or 1
            rc = self._lib.crl_tron_next_state_inplace64_host(*(self._a_host64_obs if with_obs else self._a_host64), seq, self.WAIT_TIMEOUT_S)
            if rc:
                check(rc, "crl_tron_next_state_inplace64_host")
            return
        rc = self._lib.crl_tron_next_state_inplace64(*(self._a_next64_obs if with_obs else self._a_next64))
        if rc:
            check(rc, "crl_tron_next_state_inplace64")
        self.sync()

    def relative_board64(self, board, player_plus_1: int, num_players: int = None):
        """``relative_player_inplace(copy of board, num_players, player_plus_1)`` (CyTronGrid.pyx:65-71): the relabelled int64
        board (`num_players` defaults to this game's)."""
        NN = self.NN
        dst = self.s64["obs"][:NN].reshape(self.N, self.N)
        np.copyto(dst, board, casting="unsafe")
        self.s64["player"][0] = player_plus_1
        args = self._a_rel64 if num_players is None else self._a_rel64[:3] + (int(num_players),) + self._a_rel64[4:]
        rc = self._lib.crl_tron_relative_player_inplace64(*args)
        if rc:
            check(rc, "crl_tron_relative_player_inplace64")
        self.sync()
        return dst.copy()

    def ranking(self):
        rc = self._lib.crl_tron_ranking(*self._a_rank)
        if rc:
            check(rc, "crl_tron_ranking")
        self.sync()
        return self.v["rank"]

    def reset(self):
        rc = self._lib.crl_tron_reset(*self._a_reset)
        if rc:
            check(rc, "crl_tron_reset")
        self.sync()


class SingleTTT(_Single):
    """One TicTacToe game in the reference's layout (board int8 [cells] with -1 = empty, winner, to_move)."""

    def __init__(self, dims: Sequence[int], k: int, num_players: int, rel_mod: int):
        super().__init__()
        lib = self._lib
        d3 = (1,) * (3 - len(dims)) + tuple(int(x) for x in dims)
        self.n_cells = d3[0] * d3[1] * d3[2]
        self.P = int(num_players)
        check(lib.crl_ttt_create(d3[0], d3[1], d3[2], int(k), self.P, C.byref(self._handle)), "crl_ttt_create")
        self._open_stream()
        n = self.n_cells
        self.blob = HostBlob(lib, [("board", np.int8, n), ("obs_board", np.int8, n), ("winner", np.int8, 1),
                                   ("to_move", np.int8, 1), ("action", np.int8, 1), ("reward", np.int8, 1),
                                   ("terminal", np.uint8, 1), ("winners", np.int8, 1), ("valid", np.uint32, 1),
                                   ("player", np.int8, 1)])
        v, d = self.blob.v, self.blob.d
        self.v = v
        h, s = self._handle, self._stream
        self._a_step = (h, 1, d["board"], d["winner"], d["to_move"], d["action"], d["reward"], d["terminal"],
                        d["winners"], d["valid"], d["obs_board"], int(rel_mod), 0, s)
        # the one-call form (crl_ttt_step_board_host): only where the mapped blocks have ONE address for host and GPU
        self._unified = (d["board"].value == v["board"].ctypes.data and self._flag.d["seq"].value == self._flag.v["seq"].ctypes.data)
        self._a_step_host = (h, d["board"], d["winner"], d["to_move"], d["action"], d["reward"], d["terminal"],
                             d["winners"], d["valid"], d["obs_board"], int(rel_mod), 0, s, self._flag.d["seq"])
        self._a_valid = (h, 1, d["board"], None, int(rel_mod), None, d["valid"], s)
        self._a_obs = (h, 1, d["board"], d["player"], int(rel_mod), d["obs_board"], d["valid"], s)

    def load(self, board, winner, mover: int):
        v = self.v
        np.copyto(v["board"], np.asarray(board).reshape(-1), casting="unsafe")
        v["winner"][0] = -1 if winner is None else int(winner)
        v["to_move"][0] = int(mover)

    def step(self, cell: int):
        """next_state + empties mask and observation for the player to move next, one launch."""
        self.v["action"][0] = cell
        if self._unified:                                        # launch + completion in one call, the state by value
            self._seq = seq = (self._seq + 1) & 0xFFFFFFFF or 1
            rc = self._lib.crl_ttt_step_board_host(*self._a_step_host, seq, self.WAIT_TIMEOUT_S)
            if rc:
                check(rc, "crl_ttt_step_board_host")
            return
        rc = self._lib.crl_ttt_step_board(*self._a_step)
        if rc:
            check(rc, "crl_ttt_step_board")
        self.sync()

    def valid(self) -> int:
        rc = self._lib.crl_ttt_observe_board(*self._a_valid)
        if rc:
            check(rc, "crl_ttt_observe_board")
        self.sync()
        return int(self.v["valid"][0])

    def observe(self, player: int):
        self.v["player"][0] = player
        rc = self._lib.crl_ttt_observe_board(*self._a_obs)
        if rc:
            check(rc, "crl_ttt_observe_board")
        self.sync()


class SingleBlokus(_Single):
    """One Blokus game: ``Board.board_contents`` (int8 [20][20], 0 empty else colour), inventories as bit masks,
    scores, round, mover -- on host-mapped memory; the row bitboards the kernels work on are derived on the GPU
    (``crl_blokus_pack``) in the same stream."""
    CAP = 4096          # legal actions the list holds at first (observed maximum in reference games: 1,693); a state with more
                        # (hand-made boards: 12,952 seen) makes the list grow and the kernel run once more
    MAX_IDS = 21 * 400 * 8 * 5

    def __init__(self):
        super().__init__()
        lib = self._lib
        check(lib.crl_blokus_create(C.byref(self._handle)), "crl_blokus_create")
        self._open_stream()
        self.blob = HostBlob(lib, [
            ("board", np.int8, 400), ("obs_board", np.int8, 400), ("occ", np.uint32, 80), ("inv", np.uint32, 4),
            ("score", np.int32, 4), ("round", np.int32, 1), ("to_move", np.int32, 1), ("action", np.int32, 1),
            ("reward", np.int8, 1), ("terminal", np.uint8, 1), ("winners", np.uint8, 1), ("n_valid", np.int32, 1),
            ("obs_pieces", np.uint8, 84), ("obs_score", np.int32, 4), ("obs_player", np.int8, 1), ("player", np.int8, 1),
            ("count", np.int32, 1), ("ok", np.uint8, 1)])
        v, d = self.blob.v, self.blob.d
        self.v = v
        h, s = self._handle, self._stream
        st = (d["occ"], d["inv"], d["score"], d["round"], d["to_move"])
        self._st = st
        self._a_pack = (h, 1, d["board"], d["occ"], s)
        self._a_unpack = (h, 1, d["occ"], d["board"], s)
        self._a_step = (h, 1, 0, 0) + st + (d["action"], None, d["reward"], d["terminal"], d["winners"], d["n_valid"],
                                            d["obs_board"], d["obs_pieces"], d["obs_score"], d["obs_player"], 0, s)
        self._bind_list(self.CAP)
        self._a_is_valid = (h, 1) + st + (d["player"], d["action"], d["ok"], s)
        self._a_fits = (h, 1, d["occ"], d["player"], d["action"], d["ok"], s)
        self._a_observe = (h, 1, d["occ"], d["inv"], d["score"], d["player"], d["obs_board"], d["obs_pieces"],
                           d["obs_score"], s)
        self._listed = None                                      # whose list is on the block: None = the mover's

    def _bind_list(self, cap: int):
        """(Re)allocate the mapped id list with room for `cap` ids and bind the list calls' arguments to it."""
        self.cap = int(cap)
        self._ids_blob = HostBlob(self._lib, [("ids", np.int32, self.cap)])      # (the old block is freed with its last view)
        self._ids_view = self._ids_blob.v["ids"]
        d, ids = self.blob.d, self._ids_blob.d["ids"]
        self._a_list_mover = (self._handle, 1) + self._st + (None, ids, d["count"], self.cap, self._stream)
        self._a_list = (self._handle, 1) + self._st + (d["player"], ids, d["count"], self.cap, self._stream)

    def load(self, cells, inv_masks, scores, round_count: int, mover: int):
        v = self.v
        np.copyto(v["board"], np.asarray(cells).reshape(-1), casting="unsafe")
        v["inv"][:] = inv_masks
        v["score"][:] = scores
        v["round"][0] = round_count
        v["to_move"][0] = mover
        rc = self._lib.crl_blokus_pack(*self._a_pack)              # async: the next launch on this stream sees occ
        if rc:
            check(rc, "crl_blokus_pack")

    def step(self, action_id: int):
        """next_state, then -- for the NEW state and its mover -- the board, the ordered legal list and the observation:
        three launches, one synchronise."""
        lib = self._lib
        self.v["action"][0] = action_id
        rc = lib.crl_blokus_step_observe(*self._a_step) or lib.crl_blokus_board(*self._a_unpack) \
            or lib.crl_blokus_valid_list(*self._a_list_mover)
        if rc:
            check(rc, "crl_blokus_step_observe / board / valid_list")
        self.sync()
        self._listed = None                                      # the list on the block is the new mover's

    def _issue_list(self, player):
        if player is None:
            rc = self._lib.crl_blokus_valid_list(*self._a_list_mover)
        else:
            self.v["player"][0] = player
            rc = self._lib.crl_blokus_valid_list(*self._a_list)
        if rc:
            check(rc, "crl_blokus_valid_list")

    def legal_ids(self, player: int) -> np.ndarray:
        self._listed = player
        self._issue_list(player)
        self.sync()
        return self.ids()

    def ids(self) -> np.ndarray:
        """The list the last `step` (for the new mover) or `legal_ids` call left on the block.  A list longer than the
        block's capacity -- the kernel still counted all of it -- gets a larger block and one more launch."""
        n = int(self.v["count"][0])
        if n > self.cap:
            if n > self.MAX_IDS:
                raise _native.NativeError("%d legal Blokus actions: more than there are action ids" % n)
            grow = self.cap
            while grow < n:
                grow *= 2
            self._bind_list(min(grow, self.MAX_IDS))
            self._issue_list(self._listed)                       # the exceptional second launch of this call
            self.sync()
            n = int(self.v["count"][0])
        return self._ids_view[:n].copy()

    def is_valid(self, player: int, action_id: int) -> bool:
        self.v["player"][0] = player
        self.v["action"][0] = action_id
        rc = self._lib.crl_blokus_is_valid(*self._a_is_valid)
        if rc:
            check(rc, "crl_blokus_is_valid")
        self.sync()
        return bool(self.v["ok"][0])

    def fits(self, player: int, action_id: int) -> bool:
        """Every cell of the placement on the board, empty and not orthogonally next to `player`'s colour (anchor and
        inventory not asked: ``crl_blokus_fits``)."""
        self.v["player"][0] = player
        self.v["action"][0] = action_id
        rc = self._lib.crl_blokus_fits(*self._a_fits)
        if rc:
            check(rc, "crl_blokus_fits")
        self.sync()
        return bool(self.v["ok"][0])

    def observe(self, player: int):
        self.v["player"][0] = player
        rc = self._lib.crl_blokus_observe(*self._a_observe)
        if rc:
            check(rc, "crl_blokus_observe")
        self.sync()
