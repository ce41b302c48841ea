"""Batched (vectorised) environment steppers: B independent games resident in HBM.

This is the new surface the reference does not have (it steps one Python state
per call): ``reset / step / rollout / observe`` over struct-of-arrays state held
in torch-ROCm tensors and advanced by the HIP kernels behind the C ABI
(include/colosseum_hip.h).  The single-state ``BaseEnvironment`` classes in
``colosseumrl_amd.envs`` are thin B=1 clients of these steppers.

No CPU path exists here: constructing a stepper without a visible MI355X raises.
"""
from typing import Optional, Sequence

import ctypes as C

import torch

from . import _native
from ._native import CRL_STEP_AUTO_RESET, TronStats, TTTStats, check
from .envs.tron import layout as tron_layout


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """torch's CURRENT stream of the current device as a hipStream_t.  (Through the raw-stream query where this torch has
    it: `torch.cuda.current_stream()` builds a Stream object per call, 1-2 us next to a 20 us launch.)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_ROLLOUT_KERNEL_FLAGS = {"auto": 0, "bits": _native.CRL_ROLLOUT_BITS, "bytes": _native.CRL_ROLLOUT_BYTES,
                         "global": _native.CRL_ROLLOUT_NO_LDS, "quad": _native.CRL_ROLLOUT_QUAD,
                         "qbits": _native.CRL_ROLLOUT_QBITS, "gquad": _native.CRL_ROLLOUT_GQUAD, "pair": _native.CRL_ROLLOUT_PAIR}


_STEP_KERNEL_FLAGS = {"auto": 0, "bytes": _native.CRL_STEP_BYTES, "staged": _native.CRL_STEP_STAGED}


class _DevGuard:
    """``torch.cuda.device(dev)`` only when `dev` is not already current (the context manager costs several
    microseconds per call, which is visible next to a 20-step rollout launch)."""
    __slots__ = ("dev", "prev")

    def __init__(self, dev):
        self.dev = dev.index
        self.prev = -1

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.dev:
            self.prev = cur
            torch.cuda.set_device(self.dev)

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
            self.prev = -1
        return False


def _want(t: torch.Tensor, dtype, shape, device, name):
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous() or t.device != device:
        raise ValueError("%s must be a contiguous %s tensor of shape %s on %s (got %s %s on %s)"
                         % (name, dtype, tuple(shape), device, t.dtype, tuple(t.shape), t.device))
    return t


class _Ctx:
    """Owns one crl_ctx handle."""

    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            if self.handle:
                _native.lib().crl_destroy(self.handle)
                self.handle = None
        except Exception:  # interpreter shutdown
            pass


class _Waitable:
    """``wait()``: block the host until everything queued so far on the stream the stepper launches on (torch's CURRENT
    stream of its device) has run -- the stream-scoped end of a rollout.  The wait is on MAPPED MEMORY
    (``crl_stream_wait_mapped``, include/colosseum_hip.h: a one-thread kernel behind the queued work publishes a sequence
    number into page-locked host memory, the host spins on it): ~3.5 us less than ``hipStreamSynchronize`` /
    ``torch.cuda.synchronize()`` behind a short launch (tools/ubench/mailbox_rtt.hip), which is a tenth of a 20-step
    rollout region.  Work on OTHER streams is not waited for (``torch.cuda.synchronize()`` does that)."""
    WAIT_TIMEOUT_S = 30.0            # after this long without the flag the call falls back to hipStreamSynchronize (and its error)
    _wait_flag = None

    def _open_wait(self):
        """Construction-time set-up of the completion channel: the flag word in mapped memory (a hipHostMalloc: ~1.3 ms, and the
        launches right behind a fresh mapping are slow) and one first wait, so that no rollout region ever pays for either --
        a stepper whose first `wait()` allocated lazily ran its NEXT region 6 us (35 us under a process group) slower."""
        import numpy as np
        from .single import HostBlob
        with torch.cuda.device(self.device):                   # the mapping is made for the stepper's device
            blob = HostBlob(self._lib, [("seq", np.uint32, 1)])
        self._wait_flag = (blob, blob.d["seq"], C.c_void_p(blob.v["seq"].ctypes.data))
        self._wait_seq = 0
        if not torch.cuda.is_current_stream_capturing():        # (a stepper built under graph capture: nothing may block there)
            self.wait()

    def wait(self):
        if self._wait_flag is None:
            self._open_wait()
            return
        self._wait_seq = seq = (self._wait_seq + 1) & 0xFFFFFFFF or 1
        with _DevGuard(self.device):
            rc = self._lib.crl_stream_wait_mapped(_stream(), self._wait_flag[1], self._wait_flag[2], seq, self.WAIT_TIMEOUT_S)
        if rc:
            check(rc, "crl_stream_wait_mapped")


class TronBatch(_Waitable):
    """B games of N x N Tron with P players (reference: envs/tron/TronGridEnvironment.py).

    State tensors (device):
      board  int8 [B, N*N]; heads int16 [P, B]; dirs int8 [P, B]; deaths int8 [P, B]
    """

    def __init__(self, board_size: int = 19, num_players: int = 4, batch: int = 1, device="cuda",
                 ring_offset: int = 1, spawn_offset: int = 2,
                 start: Optional[Sequence[Sequence[int]]] = None, first_env_id: int = 0):
        lib = _native.require_gpu()
        self.N, self.P, self.B = int(board_size), int(num_players), int(batch)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _native.NativeError("TronBatch needs a ROCm device; there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if start is None:
            start = tron_layout.start_positions(self.N, self.P, ring_offset, [spawn_offset] * self.P)
        self.start_heads = [int(h) for h in start[0]]
        self.start_dirs = [int(d) for d in start[1]]
        sh = (C.c_int16 * self.P)(*self.start_heads)
        sd = (C.c_int8 * self.P)(*self.start_dirs)
        handle = C.c_void_p()
        check(lib.crl_tron_create(self.N, self.P, sh, sd, C.byref(handle)), "crl_tron_create")
        self._ctx = _Ctx(handle)
        self._lib = lib
        self.first_env_id = int(first_env_id)
        dev, B, P, NN = self.device, self.B, self.P, self.N * self.N
        with torch.cuda.device(dev):
            self.board = torch.zeros((B, NN), dtype=torch.int8, device=dev)
            self.heads = torch.zeros((P, B), dtype=torch.int16, device=dev)
            self.dirs = torch.zeros((P, B), dtype=torch.int8, device=dev)
            self.deaths = torch.zeros((P, B), dtype=torch.int8, device=dev)
            self.rewards = torch.zeros((P, B), dtype=torch.int8, device=dev)
            self.terminal = torch.zeros((B,), dtype=torch.uint8, device=dev)
            self.winners = torch.zeros((B,), dtype=torch.uint8, device=dev)
            # rollout bookkeeping
            self.tcount = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.tstep = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.n_episodes = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.win_count = torch.zeros((P, B), dtype=torch.int32, device=dev)
            self.len_sum = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.ret_sum = torch.zeros((P, B), dtype=torch.int32, device=dev)
            self.last_winners = torch.zeros((B,), dtype=torch.uint8, device=dev)
            self.last_len = torch.zeros((B,), dtype=torch.int16, device=dev)
            self._results = torch.zeros((B, 3 + 2 * P), dtype=torch.int32, device=dev)   # packed by the rollout kernels
            # the same row in 16-bit fields (include/colosseum_hip.h, crl_tron_stats.packed): 16 bytes per game at P = 4
            self._packed = torch.zeros((B, (4 + P + 1) & ~1), dtype=torch.int16, device=dev)
        self._stat_steps = 0                      # rollout steps the running totals span (since reset_stats)
        self._rollout_args = None
        self.reset()
        self._open_wait()

    # -- new_state for all (or masked) games
    def reset(self, mask: Optional[torch.Tensor] = None):
        if mask is not None:
            _want(mask, torch.uint8, (self.B,), self.device, "mask")
        with torch.cuda.device(self.device):
            check(self._lib.crl_tron_reset(self._ctx.handle, self.B, _ptr(mask), _ptr(self.board), _ptr(self.heads),
                                           _ptr(self.dirs), _ptr(self.deaths), _stream()), "crl_tron_reset")

    def reset_stats(self):
        for t in (self.tcount, self.tstep, self.n_episodes, self.win_count, self.len_sum, self.ret_sum,
                  self.last_winners, self.last_len, self._results, self._packed):
            t.zero_()
        self._stat_steps = 0

    # -- next_state for all games; actions int8 [P, B] in {0, +1, -1}
    def step(self, actions: torch.Tensor, auto_reset: bool = False, kernel: str = "auto"):
        """next_state of every game (``crl_tron_step``).  ``kernel``: "auto", or pin one of the two interchangeable kernels --
        "bytes" (byte probes in HBM) / "staged" (boards read once into LDS; ignored where the board shape does not allow it)."""
        _want(actions, torch.int8, (self.P, self.B), self.device, "actions")
        flags = (CRL_STEP_AUTO_RESET if auto_reset else 0) | _STEP_KERNEL_FLAGS[kernel]
        with torch.cuda.device(self.device):
            check(self._lib.crl_tron_step(self._ctx.handle, self.B, _ptr(self.board), _ptr(self.heads), _ptr(self.dirs),
                                          _ptr(self.deaths), _ptr(actions), _ptr(self.rewards), _ptr(self.terminal),
                                          _ptr(self.winners), flags, _stream()),
                  "crl_tron_step")
        return self.rewards, self.terminal, self.winners

    def _stats(self):
        return TronStats(*[t.data_ptr() for t in (self.tcount, self.tstep, self.n_episodes, self.win_count,
                                                   self.len_sum, self.ret_sum, self.last_winners, self.last_len,
                                                   self._results, self._packed)])

    # -- T fused random-agent steps with auto-reset
    def rollout(self, steps: int, seed: int = 0, use_lds: bool = True, kernel: str = "auto", events=None):
        """``kernel``: "auto" (library's choice), "quad" / "pair" / "qbits" / "bits" / "bytes" (pin one of the LDS kernels), "gquad" / "global"
        (boards in global memory: one lane per player / per game);
        ``use_lds=False`` is the older spelling of "global".  All kernels give identical results.
        ``events``: an optional pair (start, stop) of ``torch.cuda.Event(enable_timing=True)`` -- either may be None -- that
        have been recorded at least once (torch creates the HIP event at the first record): the launch carries them in
        its dispatch (``crl_tron_rollout_timed``), so ``start.elapsed_time(stop)`` after a synchronise is the rollout's
        kernel time without marker packets around it."""
        flags = _ROLLOUT_KERNEL_FLAGS[kernel]
        if not use_lds:
            flags = _native.CRL_ROLLOUT_NO_LDS
        if self._rollout_args is None:       # the state / statistics tensors are never reallocated: bind them once
            self._rollout_args = (_ptr(self.board), _ptr(self.heads), _ptr(self.dirs), _ptr(self.deaths), self._stats())
        with _DevGuard(self.device):
            if events is None:
                rc = self._lib.crl_tron_rollout(self._ctx.handle, self.B, seed & (2 ** 64 - 1), self.first_env_id, int(steps),
                                                *self._rollout_args, flags, _stream())
            else:
                handles = [C.c_void_p(e.cuda_event) if e is not None else None for e in events]
                if any(e is not None and not e.cuda_event for e in events):
                    raise ValueError("rollout(events=...): record() each event once before passing it (torch creates the "
                                     "HIP event lazily)")
                rc = self._lib.crl_tron_rollout_timed(self._ctx.handle, self.B, seed & (2 ** 64 - 1), self.first_env_id,
                                                      int(steps), *self._rollout_args, flags, _stream(), handles[0], handles[1])
        if rc:
            check(rc, "crl_tron_rollout")
        self._stat_steps += int(steps)

    def check_state(self) -> int:
        """Number of games whose state breaks the invariant of every reset / step / rollout product that the LDS
        rollout kernels rely on (heads on the board, ``board[heads[p]] == p + 1``); 0 unless states were hand-made.
        Synchronises."""
        bad = torch.zeros((1,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.crl_tron_check_state(self._ctx.handle, self.B, _ptr(self.board), _ptr(self.heads), _ptr(bad),
                                                 _stream()), "crl_tron_check_state")
        return int(bad.item())

    # -- the rollout's random agent for one step: int8 [P][B] actions in the step() encoding
    def sample(self, seed: int = 0, advance: bool = True):
        """Actions the fused rollout would take at each game's step counter (``tcount``); with ``advance`` the counter
        moves on, so ``step(sample(seed), auto_reset=True)`` T times == ``rollout(T, seed)``.  Overwrite the rows of
        the players you control."""
        act = torch.empty((self.P, self.B), dtype=torch.int8, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.crl_tron_sample(self._ctx.handle, self.B, seed & (2 ** 64 - 1), self.first_env_id,
                                            _ptr(self.tcount), int(advance), _ptr(act), _stream()), "crl_tron_sample")
        return act

    # -- state_to_observation for all games; player int8 [B]
    def observe(self, player: torch.Tensor):
        _want(player, torch.int8, (self.B,), self.device, "player")
        ob = torch.empty_like(self.board)
        oh = torch.empty_like(self.heads)
        od = torch.empty_like(self.dirs)
        ok = torch.empty_like(self.deaths)
        with torch.cuda.device(self.device):
            check(self._lib.crl_tron_observe(self._ctx.handle, self.B, _ptr(self.board), _ptr(self.heads), _ptr(self.dirs),
                                             _ptr(self.deaths), _ptr(player), _ptr(ob), _ptr(oh), _ptr(od), _ptr(ok),
                                             _stream()), "crl_tron_observe")
        return {"board": ob.view(self.B, self.N, self.N), "heads": oh, "directions": od, "deaths": ok}

    # -- state_to_observation of every game for every observer in one pass
    def observe_all(self, out: Optional[dict] = None):
        """{'board': int8 [P, B, N, N], 'heads': int16 [P, P, B], 'directions' / 'deaths': int8 [P, P, B]};
        slice [p] is what player p observes.  Pass a previous result as `out` to reuse its buffers."""
        P, B, N = self.P, self.B, self.N
        if out is None:
            out = self.observe_all_buffers()
        with _DevGuard(self.device):
            check(self._lib.crl_tron_observe_all(self._ctx.handle, B, _ptr(self.board), _ptr(self.heads), _ptr(self.dirs),
                                                 _ptr(self.deaths), _ptr(out["board"]), _ptr(out["heads"]),
                                                 _ptr(out["directions"]), _ptr(out["deaths"]), _stream()),
                  "crl_tron_observe_all")
        return out

    # -- [sample ->] next_state -> state_to_observation of all observers, one launch
    def step_observe(self, actions: Optional[torch.Tensor] = None, seed: int = 0, auto_reset: bool = True,
                     out: Optional[dict] = None):
        """What a self-play learner needs every step, fused: plays `actions` (int8 [P, B]; None = the rollout's random
        agent at each game's step counter, which then advances) and returns the observations of ALL P players of the
        resulting states together with the step outputs:
        {'board' [P, B, N, N], 'heads' [P, P, B], 'directions', 'deaths', 'rewards' [P, B], 'terminal' [B], 'winners' [B]}.
        Equals ``step(sample(seed) or actions, auto_reset); observe_all()``; one launch on every board size and player
        count.  Pass a previous result as `out` to reuse its observation buffers."""
        P, B, N = self.P, self.B, self.N
        if actions is not None:
            _want(actions, torch.int8, (P, B), self.device, "actions")
        if out is None:
            out = self.observe_all_buffers()
        with _DevGuard(self.device):
            check(self._lib.crl_tron_step_observe(self._ctx.handle, B, seed & (2 ** 64 - 1), self.first_env_id,
                                                  _ptr(self.board), _ptr(self.heads), _ptr(self.dirs), _ptr(self.deaths),
                                                  _ptr(actions), _ptr(self.tcount), _ptr(self.rewards), _ptr(self.terminal),
                                                  _ptr(self.winners), _ptr(out["board"]), _ptr(out["heads"]),
                                                  _ptr(out["directions"]), _ptr(out["deaths"]),
                                                  CRL_STEP_AUTO_RESET if auto_reset else 0, _stream()), "crl_tron_step_observe")
        out["rewards"], out["terminal"], out["winners"] = self.rewards, self.terminal, self.winners
        return out

    def observe_all_buffers(self):
        P, B, N = self.P, self.B, self.N
        return {"board": torch.empty((P, B, N, N), dtype=torch.int8, device=self.device),
                "heads": torch.empty((P, P, B), dtype=torch.int16, device=self.device),
                "directions": torch.empty((P, P, B), dtype=torch.int8, device=self.device),
                "deaths": torch.empty((P, P, B), dtype=torch.int8, device=self.device)}

    # -- compute_ranking for all games: int8 [P, B], 0 = best
    def ranking(self):
        out = torch.empty((self.P, self.B), dtype=torch.int8, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.crl_tron_ranking(self._ctx.handle, self.B, _ptr(self.board), _ptr(self.deaths), _ptr(out),
                                             _stream()), "crl_tron_ranking")
        return out

    def results(self, copy: bool = True):
        """Per-game episode results packed for the end-of-rollout gather: int32 [B, 3+2P] =
        n_episodes, len_sum, last_winners, win_count[P], ret_sum[P].  The rows are written by the rollout kernel
        itself at the end of every launch (``crl_tron_stats.results``): no packing pass.  By default a snapshot;
        ``copy=False`` hands out the live buffer, which the next rollout rewrites in place."""
        return self._results.clone() if copy else self._results

    rollout_takes_events = True    # rollout(events=(start, stop)): HIP events attached to the dispatches

    PACKED_EXACT_STEPS = 3276      # |ret_sum| <= 10 per step: the int16 fields hold the totals of this many steps

    def packed_rows_exact(self) -> bool:
        """True while the 16-bit row of ``results_packed`` cannot have wrapped: the running totals span at most
        ``PACKED_EXACT_STEPS`` rollout steps since ``reset_stats`` (known on the host: it issues the launches)."""
        return self._stat_steps <= self.PACKED_EXACT_STEPS

    def results_packed(self, copy: bool = True):
        """The gather row in 16-bit fields: int16 [B, (4+P+1)&~1] = n_episodes, len_sum, last_winners, tstep (steps into
        the unfinished episode), ret_sum[P] -- the low 16 bits of the running totals, written by the rollout kernel
        (``crl_tron_stats.packed``); 16 bytes per game at P = 4 instead of 44.  Exact while ``packed_rows_exact()``."""
        return self._packed.clone() if copy else self._packed

    def results_from_columns(self):
        """The same rows assembled from the per-column statistics (what ``results()`` must equal; used by the tests)."""
        cols = [self.n_episodes, self.len_sum, self.last_winners.to(torch.int32)]
        cols += [self.win_count[p] for p in range(self.P)] + [self.ret_sum[p] for p in range(self.P)]
        return torch.stack(cols, dim=1).contiguous()

    def results_packed_from_columns(self):
        """The 16-bit rows assembled from the per-column statistics (what ``results_packed()`` must equal; tests)."""
        cols = [self.n_episodes, self.len_sum, self.last_winners.to(torch.int32), self.tstep]
        cols += [self.ret_sum[p] for p in range(self.P)]
        cols += [torch.zeros_like(self.tstep)] * (self._packed.shape[1] - len(cols))
        return torch.stack(cols, dim=1).to(torch.int16).contiguous()


class TTTBatch(_Waitable):
    """B games of n-player TicTacToe on a dims board, K in a row (reference: envs/tictactoe/*).

    State tensors (device): occ int32 [P, B] bit masks; winner int8 [B] (-1 none); to_move int8 [B].
    """

    def __init__(self, dims: Sequence[int] = (3, 3), k: int = 3, num_players: int = 2, batch: int = 1,
                 device="cuda", first_env_id: int = 0):
        lib = _native.require_gpu()
        self.dims = tuple(int(d) for d in dims)
        d3 = (1,) * (3 - len(self.dims)) + self.dims
        self.K, self.P, self.B = int(k), int(num_players), int(batch)
        self.n_cells = d3[0] * d3[1] * d3[2]
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _native.NativeError("TTTBatch needs a ROCm device; there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        handle = C.c_void_p()
        with torch.cuda.device(self.device):     # (boards of <= 16 cells: the context keeps its win-mask table on THIS device)
            check(lib.crl_ttt_create(d3[0], d3[1], d3[2], self.K, self.P, C.byref(handle)), "crl_ttt_create")
        self._ctx = _Ctx(handle)
        self._lib = lib
        self.first_env_id = int(first_env_id)
        dev, B, P = self.device, self.B, self.P
        with torch.cuda.device(dev):
            self.occ = torch.zeros((P, B), dtype=torch.int32, device=dev)
            self.winner = torch.full((B,), -1, dtype=torch.int8, device=dev)
            self.to_move = torch.zeros((B,), dtype=torch.int8, device=dev)
            self.reward = torch.zeros((B,), dtype=torch.int8, device=dev)
            self.terminal = torch.zeros((B,), dtype=torch.uint8, device=dev)
            self.winners = torch.zeros((B,), dtype=torch.int8, device=dev)
            self.tcount = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.tstep = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.n_episodes = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.win_count = torch.zeros((P, B), dtype=torch.int32, device=dev)
            self.draw_count = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.len_sum = torch.zeros((B,), dtype=torch.int32, device=dev)
            self._results = torch.zeros((B, 3 + P), dtype=torch.int32, device=dev)
        self._open_wait()

    def reset_stats(self):
        for t in (self.tcount, self.tstep, self.n_episodes, self.win_count, self.draw_count, self.len_sum, self._results):
            t.zero_()

    def lines(self):
        buf = (C.c_uint32 * 256)()
        n = self._lib.crl_ttt_lines(self._ctx.handle, buf, 256)
        return [int(buf[i]) for i in range(n)]

    def reset(self, mask: Optional[torch.Tensor] = None):
        if mask is not None:
            _want(mask, torch.uint8, (self.B,), self.device, "mask")
        with torch.cuda.device(self.device):
            check(self._lib.crl_ttt_reset(self._ctx.handle, self.B, _ptr(mask), _ptr(self.occ), _ptr(self.winner),
                                          _ptr(self.to_move), _stream()), "crl_ttt_reset")

    def step(self, action: torch.Tensor, auto_reset: bool = False):
        _want(action, torch.int8, (self.B,), self.device, "action")
        with torch.cuda.device(self.device):
            check(self._lib.crl_ttt_step(self._ctx.handle, self.B, _ptr(self.occ), _ptr(self.winner), _ptr(self.to_move),
                                         _ptr(action), _ptr(self.reward), _ptr(self.terminal), _ptr(self.winners),
                                         CRL_STEP_AUTO_RESET if auto_reset else 0, _stream()), "crl_ttt_step")
        return self.reward, self.terminal, self.winners

    def valid_mask(self):
        out = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.crl_ttt_valid(self._ctx.handle, self.B, _ptr(self.occ), _ptr(out), _stream()), "crl_ttt_valid")
        return out

    def sample(self, seed: int = 0, advance: bool = True):
        """The rollout's random agent for one step: int8 [B] flat cell (uniform over the empty cells, -1 on a full
        board) at each game's step counter; ``step(sample(seed), auto_reset=True)`` T times == ``rollout(T, seed)``."""
        act = torch.empty((self.B,), dtype=torch.int8, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.crl_ttt_sample(self._ctx.handle, self.B, seed & (2 ** 64 - 1), self.first_env_id, _ptr(self.occ),
                                           _ptr(self.tcount), int(advance), _ptr(act), _stream()), "crl_ttt_sample")
        return act

    def board(self, player: Optional[torch.Tensor] = None, rel_mod: Optional[int] = None):
        out = torch.empty((self.B, self.n_cells), dtype=torch.int8, device=self.device)
        if player is not None:
            _want(player, torch.int8, (self.B,), self.device, "player")
        with torch.cuda.device(self.device):
            check(self._lib.crl_ttt_board(self._ctx.handle, self.B, _ptr(self.occ), _ptr(player),
                                          int(rel_mod if rel_mod else self.P), _ptr(out), _stream()), "crl_ttt_board")
        return out

    def observe(self, player: torch.Tensor, rel_mod: Optional[int] = None):
        """state_to_observation for all games: board with ids relative to player[b] (reference 2p:382-407)."""
        return {"board": self.board(player, rel_mod)}

    def step_observe(self, action: Optional[torch.Tensor] = None, seed: int = 0, auto_reset: bool = True,
                     rel_mod: Optional[int] = None, out: Optional[dict] = None):
        """One ply of every game in ONE launch: plays `action` (int8 [B]; None = the rollout's random agent at each
        game's step counter, which then advances) and returns what the next mover needs:
        {'board' int8 [B, cells] relative to the player to move next, 'valid' int32 [B] empties mask, 'mover' int8 [B],
        'reward', 'terminal', 'winners'}.  Equals ``step(...); valid_mask(); board(to_move, rel_mod)``."""
        if action is not None:
            _want(action, torch.int8, (self.B,), self.device, "action")
        if out is None:
            out = {"board": torch.empty((self.B, self.n_cells), dtype=torch.int8, device=self.device),
                   "valid": torch.empty((self.B,), dtype=torch.int32, device=self.device)}
        with _DevGuard(self.device):
            check(self._lib.crl_ttt_step_observe(self._ctx.handle, self.B, seed & (2 ** 64 - 1), self.first_env_id,
                                                 _ptr(self.occ), _ptr(self.winner), _ptr(self.to_move), _ptr(action),
                                                 _ptr(self.tcount), _ptr(self.reward), _ptr(self.terminal), _ptr(self.winners),
                                                 _ptr(out["board"]), _ptr(out["valid"]), int(rel_mod if rel_mod else self.P),
                                                 CRL_STEP_AUTO_RESET if auto_reset else 0, _stream()), "crl_ttt_step_observe")
        out["mover"], out["reward"], out["terminal"], out["winners"] = self.to_move, self.reward, self.terminal, self.winners
        return out

    def _stats(self):
        return TTTStats(*[t.data_ptr() for t in (self.tcount, self.tstep, self.n_episodes, self.win_count,
                                                  self.draw_count, self.len_sum, self._results)])

    def rollout(self, steps: int, seed: int = 0):
        with torch.cuda.device(self.device):
            check(self._lib.crl_ttt_rollout(self._ctx.handle, self.B, seed & (2 ** 64 - 1), self.first_env_id, int(steps),
                                            _ptr(self.occ), _ptr(self.winner), _ptr(self.to_move), self._stats(),
                                            _stream()), "crl_ttt_rollout")

    def results(self, copy: bool = True):
        """int32 [B, 3+P] = n_episodes, len_sum, draw_count, win_count[P]; written by the rollout kernel at the end of
        every launch.  A snapshot by default; ``copy=False`` hands out the live buffer the next rollout rewrites."""
        return self._results.clone() if copy else self._results

    def results_from_columns(self):
        cols = [self.n_episodes, self.len_sum, self.draw_count] + [self.win_count[p] for p in range(self.P)]
        return torch.stack(cols, dim=1).contiguous()


class TTTBoards:
    """B TicTacToe games held in the REFERENCE's layout (``board int8 [B, cells]`` with -1 = empty, ``winner``,
    ``to_move``) and stepped there by ``crl_ttt_step_board`` / ``crl_ttt_observe_board`` -- what the single-state
    drop-in classes run at B = 1; `TTTBatch` (bit masks) is the layout for throughput."""

    def __init__(self, dims: Sequence[int] = (3, 3), k: int = 3, num_players: int = 2, batch: int = 1, device="cuda"):
        lib = _native.require_gpu()
        self.dims = tuple(int(d) for d in dims)
        d3 = (1,) * (3 - len(self.dims)) + self.dims
        self.K, self.P, self.B = int(k), int(num_players), int(batch)
        self.n_cells = d3[0] * d3[1] * d3[2]
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        handle = C.c_void_p()
        check(lib.crl_ttt_create(d3[0], d3[1], d3[2], self.K, self.P, C.byref(handle)), "crl_ttt_create")
        self._ctx = _Ctx(handle)
        self._lib = lib
        dev, B = self.device, self.B
        self.board = torch.full((B, self.n_cells), -1, dtype=torch.int8, device=dev)
        self.winner = torch.full((B,), -1, dtype=torch.int8, device=dev)
        self.to_move = torch.zeros((B,), dtype=torch.int8, device=dev)
        self.reward = torch.zeros((B,), dtype=torch.int8, device=dev)
        self.terminal = torch.zeros((B,), dtype=torch.uint8, device=dev)
        self.winners = torch.zeros((B,), dtype=torch.int8, device=dev)
        self.valid = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.obs = torch.zeros((B, self.n_cells), dtype=torch.int8, device=dev)

    def step(self, action: torch.Tensor, auto_reset: bool = False, rel_mod: Optional[int] = None):
        _want(action, torch.int8, (self.B,), self.device, "action")
        with _DevGuard(self.device):
            check(self._lib.crl_ttt_step_board(self._ctx.handle, self.B, _ptr(self.board), _ptr(self.winner), _ptr(self.to_move),
                                               _ptr(action), _ptr(self.reward), _ptr(self.terminal), _ptr(self.winners),
                                               _ptr(self.valid), _ptr(self.obs), int(rel_mod if rel_mod else self.P),
                                               CRL_STEP_AUTO_RESET if auto_reset else 0, _stream()), "crl_ttt_step_board")
        return self.reward, self.terminal, self.winners

    def observe(self, player: Optional[torch.Tensor] = None, rel_mod: Optional[int] = None):
        """(obs int8 [B, cells] relative to player[b] -- absolute when player is None --, empties mask int32 [B])"""
        if player is not None:
            _want(player, torch.int8, (self.B,), self.device, "player")
        obs = torch.empty_like(self.board)
        valid = torch.empty_like(self.valid)
        with _DevGuard(self.device):
            check(self._lib.crl_ttt_observe_board(self._ctx.handle, self.B, _ptr(self.board), _ptr(player),
                                                  int(rel_mod if rel_mod else self.P), _ptr(obs), _ptr(valid), _stream()),
                  "crl_ttt_observe_board")
        return obs, valid


class BlokusBatch(_Waitable):
    """B games of 4-player 20x20 Blokus (reference: envs/blokus/*).

    State tensors (device): occ int32 [B, 4, 20] row bitboards per colour; inv int32 [B, 4] piece masks;
    score int32 [B, 4]; round int32 [B]; to_move int32 [B].  Actions are dense ids
    ``((piece*400 + y*20 + x)*8 + orientation)*5 + shift`` (-1 = pass), see ``envs.blokus.actions``.
    """
    MASK_WORDS = 10500

    def __init__(self, batch: int = 1, device="cuda", first_env_id: int = 0):
        lib = _native.require_gpu()
        self.B = int(batch)
        self.P = 4
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _native.NativeError("BlokusBatch needs a ROCm device; there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.crl_blokus_create(C.byref(handle)), "crl_blokus_create")
        self._ctx = _Ctx(handle)
        self._lib = lib
        self.first_env_id = int(first_env_id)
        dev, B = self.device, self.B
        with torch.cuda.device(dev):
            self.occ = torch.zeros((B, 4, 20), dtype=torch.int32, device=dev)
            self.inv = torch.zeros((B, 4), dtype=torch.int32, device=dev)
            self.score = torch.zeros((B, 4), dtype=torch.int32, device=dev)
            self.round = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.to_move = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.reward = torch.zeros((B,), dtype=torch.int8, device=dev)
            self.terminal = torch.zeros((B,), dtype=torch.uint8, device=dev)
            self.winners = torch.zeros((B,), dtype=torch.uint8, device=dev)
            self.tcount = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.tstep = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.n_episodes = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.win_count = torch.zeros((4, B), dtype=torch.int32, device=dev)
            self.len_sum = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.score_sum = torch.zeros((4, B), dtype=torch.int32, device=dev)
            self._results = torch.zeros((B, 10), dtype=torch.int32, device=dev)
        self.reset()
        self._open_wait()

    def _state(self):
        return (_ptr(self.occ), _ptr(self.inv), _ptr(self.score), _ptr(self.round), _ptr(self.to_move))

    def reset(self, mask: Optional[torch.Tensor] = None):
        if mask is not None:
            _want(mask, torch.uint8, (self.B,), self.device, "mask")
        with torch.cuda.device(self.device):
            check(self._lib.crl_blokus_reset(self._ctx.handle, self.B, _ptr(mask), *self._state(), _stream()), "crl_blokus_reset")

    def step(self, action: torch.Tensor, auto_reset: bool = False):
        _want(action, torch.int32, (self.B,), self.device, "action")
        with torch.cuda.device(self.device):
            check(self._lib.crl_blokus_step(self._ctx.handle, self.B, *self._state(), _ptr(action), _ptr(self.reward),
                                            _ptr(self.terminal), _ptr(self.winners),
                                            CRL_STEP_AUTO_RESET if auto_reset else 0, _stream()), "crl_blokus_step")
        return self.reward, self.terminal, self.winners

    def valid(self, player: Optional[torch.Tensor] = None, want_mask: bool = False):
        """Legal-action count int32 [B] of `player` (default: the player to move) and, on request, the dense
        id bitmap int32 [B, 10500] (bit id set = legal; ascending ids = the reference's valid_actions order)."""
        if player is not None:
            _want(player, torch.int8, (self.B,), self.device, "player")
        count = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        mask = torch.empty((self.B, self.MASK_WORDS), dtype=torch.int32, device=self.device) if want_mask else None
        with torch.cuda.device(self.device):
            check(self._lib.crl_blokus_valid(self._ctx.handle, self.B, *self._state(), _ptr(player), _ptr(count),
                                             _ptr(mask), _stream()), "crl_blokus_valid")
        return (count, mask) if want_mask else count

    def valid_list(self, cap: int = 2048, player: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
        """The ordered legal-action LIST of `player` (default: the player to move), compacted: (count int32 [B],
        ids int32 [B, cap]) with ``ids[b, :min(count[b], cap)]`` = the dense ids in ascending order = the reference's
        ``valid_actions`` order (BlokusEnvironment.py:453-500); entries beyond are -1 (or whatever `out` held)."""
        if player is not None:
            _want(player, torch.int8, (self.B,), self.device, "player")
        count = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        ids = out if out is not None else torch.full((self.B, int(cap)), -1, dtype=torch.int32, device=self.device)
        _want(ids, torch.int32, (self.B, int(cap)), self.device, "out")
        with _DevGuard(self.device):
            check(self._lib.crl_blokus_valid_list(self._ctx.handle, self.B, *self._state(), _ptr(player), _ptr(ids),
                                                  _ptr(count), int(cap), _stream()), "crl_blokus_valid_list")
        return count, ids

    def select(self, rank: torch.Tensor, player: Optional[torch.Tensor] = None):
        """Dense id of the rank[b]-th legal action (reference order) of `player` (default: the player to move) without
        materialising the list; -1 where rank is outside [0, count).  -> (action int32 [B], count int32 [B])."""
        _want(rank, torch.int32, (self.B,), self.device, "rank")
        if player is not None:
            _want(player, torch.int8, (self.B,), self.device, "player")
        act = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        count = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        with _DevGuard(self.device):
            check(self._lib.crl_blokus_select(self._ctx.handle, self.B, *self._state(), _ptr(player), _ptr(rank), _ptr(act),
                                              _ptr(count), _stream()), "crl_blokus_select")
        return act, count

    def is_valid(self, action: torch.Tensor, player: Optional[torch.Tensor] = None):
        """``is_valid_action`` for all games: uint8 [B], 1 iff the dense id action[b] is a legal action of `player`."""
        _want(action, torch.int32, (self.B,), self.device, "action")
        if player is not None:
            _want(player, torch.int8, (self.B,), self.device, "player")
        ok = torch.empty((self.B,), dtype=torch.uint8, device=self.device)
        with _DevGuard(self.device):
            check(self._lib.crl_blokus_is_valid(self._ctx.handle, self.B, *self._state(), _ptr(player), _ptr(action), _ptr(ok),
                                                _stream()), "crl_blokus_is_valid")
        return ok

    def fits(self, action: torch.Tensor, player: torch.Tensor):
        """uint8 [B]: 1 iff every cell of the placement action[b] lies on the board, is empty and has no orthogonal neighbour
        of player[b]'s colour -- ``is_valid`` without the anchor and inventory conditions (one shift of the reference's
        ``Board.check_orientation_shifts``)."""
        _want(action, torch.int32, (self.B,), self.device, "action")
        _want(player, torch.int8, (self.B,), self.device, "player")
        ok = torch.empty((self.B,), dtype=torch.uint8, device=self.device)
        with _DevGuard(self.device):
            check(self._lib.crl_blokus_fits(self._ctx.handle, self.B, _ptr(self.occ), _ptr(player), _ptr(action), _ptr(ok),
                                            _stream()), "crl_blokus_fits")
        return ok

    def set_board(self, board: torch.Tensor):
        """Loads ``Board.board_contents`` (int8 [B, 20, 20], 0 empty else colour) into the row bitboards."""
        _want(board, torch.int8, (self.B, 20, 20), self.device, "board")
        with _DevGuard(self.device):
            check(self._lib.crl_blokus_pack(self._ctx.handle, self.B, _ptr(board), _ptr(self.occ), _stream()), "crl_blokus_pack")

    def sample(self, seed: int = 0, advance: bool = True):
        """The rollout's random agent for one step: int32 [B] dense action id of a uniformly drawn legal action of the
        player to move (-1 = pass); ``step(sample(seed), auto_reset=True)`` T times == ``rollout(T, seed)``."""
        act = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.crl_blokus_sample(self._ctx.handle, self.B, seed & (2 ** 64 - 1), self.first_env_id,
                                              *self._state(), _ptr(self.tcount), int(advance), _ptr(act), _stream()),
                  "crl_blokus_sample")
        return act

    def observe(self, player: torch.Tensor):
        """state_to_observation for all games; player int8 [B] is the observer of each game."""
        _want(player, torch.int8, (self.B,), self.device, "player")
        ob = torch.empty((self.B, 20, 20), dtype=torch.int8, device=self.device)
        op = torch.empty((self.B, 4, 21), dtype=torch.uint8, device=self.device)
        osc = torch.empty((self.B, 4), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.crl_blokus_observe(self._ctx.handle, self.B, _ptr(self.occ), _ptr(self.inv), _ptr(self.score),
                                               _ptr(player), _ptr(ob), _ptr(op), _ptr(osc), _stream()), "crl_blokus_observe")
        return {"board": ob, "pieces": op, "score": osc, "player": player.view(self.B, 1)}

    def step_observe(self, action: Optional[torch.Tensor] = None, seed: int = 0, auto_reset: bool = True,
                     out: Optional[dict] = None, list_cap: int = 0):
        """One ply of every game in ONE launch: plays `action` (int32 [B] dense ids; None = the rollout's random agent
        at each game's step counter, which then advances) and returns what the next mover needs:
        {'board' int8 [B, 20, 20], 'pieces' uint8 [B, 4, 21], 'score' int32 [B, 4], 'player' int8 [B, 1] (its observation),
        'n_valid' int32 [B] (its number of legal actions), 'reward', 'terminal', 'winners'}.
        Equals ``step(...); valid(); observe(to_move)``.
        ``list_cap`` > 0 queues ``valid_list(list_cap)`` for the next mover right behind it (same stream, no synchronise in
        between) and adds 'ids' int32 [B, list_cap]: the ordered legal ids a policy picks from (-1 beyond 'n_valid')."""
        if action is not None:
            _want(action, torch.int32, (self.B,), self.device, "action")
        if out is None:
            out = {"board": torch.empty((self.B, 20, 20), dtype=torch.int8, device=self.device),
                   "pieces": torch.empty((self.B, 4, 21), dtype=torch.uint8, device=self.device),
                   "score": torch.empty((self.B, 4), dtype=torch.int32, device=self.device),
                   "player": torch.empty((self.B, 1), dtype=torch.int8, device=self.device),
                   "n_valid": torch.empty((self.B,), dtype=torch.int32, device=self.device)}
        with _DevGuard(self.device):
            check(self._lib.crl_blokus_step_observe(self._ctx.handle, self.B, seed & (2 ** 64 - 1), self.first_env_id,
                                                    *self._state(), _ptr(action), _ptr(self.tcount), _ptr(self.reward),
                                                    _ptr(self.terminal), _ptr(self.winners), _ptr(out["n_valid"]),
                                                    _ptr(out["board"]), _ptr(out["pieces"]), _ptr(out["score"]),
                                                    _ptr(out["player"]), CRL_STEP_AUTO_RESET if auto_reset else 0, _stream()),
                  "crl_blokus_step_observe")
        out["reward"], out["terminal"], out["winners"] = self.reward, self.terminal, self.winners
        if list_cap > 0:
            ids = out.get("ids")
            if ids is None or tuple(ids.shape) != (self.B, int(list_cap)):
                ids = torch.empty((self.B, int(list_cap)), dtype=torch.int32, device=self.device)
            ids.fill_(-1)
            out["ids"] = self.valid_list(int(list_cap), out=ids)[1]
        return out

    def board(self):
        out = torch.empty((self.B, 20, 20), dtype=torch.int8, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.crl_blokus_board(self._ctx.handle, self.B, _ptr(self.occ), _ptr(out), _stream()), "crl_blokus_board")
        return out

    def _stats(self):
        return _native.BlokusStats(*[t.data_ptr() for t in (self.tcount, self.tstep, self.n_episodes, self.win_count,
                                                            self.len_sum, self.score_sum, self._results)])

    def reset_stats(self):
        for t in (self.tcount, self.tstep, self.n_episodes, self.win_count, self.len_sum, self.score_sum, self._results):
            t.zero_()

    def rollout(self, steps: int, seed: int = 0):
        with torch.cuda.device(self.device):
            check(self._lib.crl_blokus_rollout(self._ctx.handle, self.B, seed & (2 ** 64 - 1), self.first_env_id, int(steps),
                                               *self._state(), self._stats(), _stream()), "crl_blokus_rollout")

    def results(self, copy: bool = True):
        """int32 [B, 10] = n_episodes, len_sum, win_count[4], score_sum[4]; written by the rollout kernel at the end of
        every launch.  A snapshot by default; ``copy=False`` hands out the live buffer the next rollout rewrites."""
        return self._results.clone() if copy else self._results

    def results_from_columns(self):
        cols = [self.n_episodes, self.len_sum] + [self.win_count[p] for p in range(4)] + [self.score_sum[p] for p in range(4)]
        return torch.stack(cols, dim=1).contiguous()
