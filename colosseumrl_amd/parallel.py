"""Multi-GPU sharding of the batched steppers: one process per GPU, games split by global id.

Every game is independent, so the only parallelism is over the batch axis: rank r of W
owns the contiguous block of global game ids ``shard_bounds(G, r, W)`` and runs the same kernels on
it; the random agent's RNG is keyed by the GLOBAL game id, so results do not depend on W.  There is
no data-path collective while stepping.  The single exchange is at the end of a rollout: the per-game
result rows are concatenated across ranks with ONE collective -- a ``gather`` to one rank or an ``all_gather``
(RCCL over xGMI with the ``nccl`` backend).  For Tron the row has two encodings, both written by the rollout kernel
itself: int32 ``[3+2P]`` (44 bytes per game at P = 4) and 16-bit fields (16 bytes per game: what SURVEY 8e specifies --
winners mask, episode length, P int16 returns, plus the counts); the narrow one is shipped whenever it is exact
(the running totals span at most 3,276 rollout steps -- every rank knows, it issued the launches), because a gather of
65,536 rows per GPU is latency-bound and seven 2.9 MB shards into one GPU take ~3x as long as seven 1 MB ones.
The reference has no counterpart (it has no device code at all).

The collective layer is backend-agnostic (``gloo`` on CPU tensors in the tests).
"""
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the games owned by `rank`: contiguous blocks, the first total % world ranks get one extra."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _global_rank(group: Optional[dist.ProcessGroup], group_rank: int) -> int:
    """`dist.gather` names its destination by GLOBAL rank; callers of this module name ranks inside `group`."""
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def gather_results(local: torch.Tensor, total: int, group: Optional[dist.ProcessGroup] = None,
                   out: Optional[torch.Tensor] = None, dst: Optional[int] = None) -> Optional[torch.Tensor]:
    """Concatenate per-game result rows [n_local, k] of all ranks in global game order: on every rank (``dst=None``,
    one ``all_gather_into_tensor``) or on rank `dst` OF `group` only (one ``gather``; the other ranks get ``None``).

    ONE collective either way; shards are padded to the largest shard so a single call moves everything.  The gather
    to one rank is what an episode-end statistics sink needs and what bench.py times: over RCCL it is a set of
    point-to-point transfers, so on the fully connected xGMI mesh of an 8-GPU node the 7 shards travel over 7 different
    links at once, where a ring all_gather needs 7 serial hops.  The collective runs whenever a process group is
    initialised -- also for a world of one rank (``torchrun --nproc-per-node 1``), so the RCCL path is the same code at
    every world size.  `out` (``[world * ceil(total / world), k]``, same dtype / device) is an optional preallocated
    receive buffer; when given, the result aliases it."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(total, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError("rank %d holds %d rows, expected %d" % (rank, local.shape[0], hi - lo))
    widest = -(-total // world)
    padded = local
    if local.shape[0] != widest:
        padded = torch.zeros((widest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    padded = padded.contiguous()
    shape = (world * widest,) + tuple(local.shape[1:])
    receiver = dst is None or rank == dst
    if receiver and (out is None or tuple(out.shape) != shape or out.dtype != local.dtype or out.device != local.device):
        out = torch.empty(shape, dtype=local.dtype, device=local.device)
    if dst is None:
        dist.all_gather_into_tensor(out, padded, group=group)
    else:
        pieces = [out[r * widest: (r + 1) * widest] for r in range(world)] if receiver else None
        dist.gather(padded, pieces, dst=_global_rank(group, dst), group=group)
        if not receiver:
            return None
    if total == world * widest:
        return out
    pieces = []
    for r in range(world):
        a, b = shard_bounds(total, r, world)
        pieces.append(out[r * widest: r * widest + (b - a)])
    return torch.cat(pieces, dim=0)


class ShardedRollout:
    """Random-agent rollouts of `global_batch` games spread over the ranks of `group` (default: the world).

    make_stepper(batch=n, first_env_id=lo) must return an object with ``rollout(steps, seed)`` and
    ``results(copy=False) -> tensor [n, k]`` (``TronBatch`` / ``TTTBatch`` / ``BlokusBatch`` partials); a stepper that
    also has ``results_packed`` / ``packed_rows_exact`` (``TronBatch``) gets its 16-byte rows shipped while they are exact.
    """

    def __init__(self, make_stepper: Callable, global_batch: int, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.total = int(global_batch)
        self.dist = dist.is_available() and dist.is_initialized()
        if self.dist:
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        self.lo, self.hi = shard_bounds(self.total, self.rank, self.world)
        self.stepper = make_stepper(batch=self.hi - self.lo, first_env_id=self.lo)
        self._equal = self.total % self.world == 0          # equal shards: the local rows go into the collective as they are
        self._recv = {}                                     # (narrow rows?, all ranks?) -> (receive buffer, its per-rank pieces, its int16 view)
        self._wire_view = None                              # (the stepper's packed rows, their int32 view)

    def rollout(self, steps: int, seed: int = 0, chunk: int = 512, events=None) -> int:
        """Exactly `steps` env-steps for every local game, as fused launches of <= chunk steps.  ``events`` = (start, stop)
        torch events a stepper that supports it (``TronBatch.rollout(events=...)``) carries in the first / last launch."""
        launches, left = 0, int(steps)
        while left > 0:
            t = min(int(chunk), left)
            if events is None:
                self.stepper.rollout(t, seed)
            else:
                self.stepper.rollout(t, seed, events=(events[0] if launches == 0 else None, events[1] if left == t else None))
            left -= t
            launches += 1
        return launches

    def wait(self):
        """Block until this rank's queued rollout launches (and a collective issued behind them on the same stream) have
        run: the stepper's stream-scoped wait on mapped memory (``TronBatch.wait`` & co., ``crl_stream_wait_mapped``) where
        it has one, else a device synchronise."""
        w = getattr(self.stepper, "wait", None)
        if w is not None:
            w()
        elif torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()

    def warm_collective(self, dst: Optional[int] = None, times: int = 3):
        """SET-UP, not stepping: bring the communicator, its channels and the receive buffer up by running the rollout's one
        collective `times` times on the rows as they are (RCCL creates the communicator lazily at the first collective, and
        the second and third calls still pay first-use costs); a no-op without a process group."""
        if self.dist:
            for _ in range(int(times)):
                self.gather(dst=dst, copy=False)

    def _local_rows(self, packed):
        st = self.stepper
        if packed is not False and hasattr(st, "results_packed") and (packed is True or st.packed_rows_exact()):
            rows = st.results_packed(copy=False)            # int16 [n, 2m]; shipped as int32 [n, m] (NCCL has no int16)
            cached = self._wire_view
            if cached is None or cached[0] is not rows:     # (the stepper's live buffer: one view for all gathers, not one per call)
                cached = self._wire_view = (rows, rows.view(torch.int32))
            return cached[1], True
        return st.results(copy=False), False

    def gather(self, dst: Optional[int] = None, packed="auto", copy: bool = True) -> Optional[torch.Tensor]:
        """Per-game results of ALL games (the one collective of a rollout): on every rank, or with `dst` (a rank of
        this object's group) on that rank only (``None`` elsewhere).

        ``packed``: "auto" ships the stepper's 16-bit rows when it has them and they are exact (every rank decides
        alike: they issued the same launches), True / False force one encoding.  The result is int32 ``[total, k]`` for
        the wide rows and int16 ``[total, 2m]`` for the 16-bit ones; column 0 is n_episodes and column 1 len_sum in both.
        ``copy=False`` returns the reused receive buffer (or, without a process group, the stepper's live rows), which
        the next gather / rollout overwrites -- the fast path bench.py times."""
        if not self.dist:                                   # no process group: the stepper's own rows, no wire format in between
            st = self.stepper
            if packed is not False and hasattr(st, "results_packed") and (packed is True or st.packed_rows_exact()):
                return st.results_packed(copy=copy)
            return st.results(copy=copy)
        local, narrow = self._local_rows(packed)
        if not self._equal:
            got = gather_results(local, self.total, self.group, dst=dst)
            if got is None:
                return None
        else:
            receiver = dst is None or self.rank == dst
            key = (narrow, dst is None)
            buf = self._recv.get(key)
            if buf is None and receiver:
                n = local.shape[0]
                out = torch.empty((self.world * n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
                buf = self._recv[key] = (out, [out[r * n: (r + 1) * n] for r in range(self.world)], out.view(torch.int16) if narrow else out)
            if dst is None:
                dist.all_gather_into_tensor(buf[0], local, group=self.group)
            else:
                dist.gather(local, buf[1] if receiver else None, dst=_global_rank(self.group, dst), group=self.group)
                if not receiver:
                    return None
            got = buf[2]
            return got.clone() if copy else got
        if narrow:
            got = got.view(torch.int16)
        return got.clone() if copy else got
