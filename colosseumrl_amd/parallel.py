"""Multi-GPU sharding of the batched steppers: one process per GPU, games split by global id.

Every game is independent, so the only parallelism is over the batch axis: rank r of W
owns the contiguous block of global game ids ``shard_bounds(G, r, W)`` and runs the same kernels on
it; the random agent's RNG is keyed by the GLOBAL game id, so results do not depend on W.  There is
no data-path collective while stepping.  The single exchange is at the end of a rollout: the per-game
result records are concatenated across ranks with ONE collective -- a ``gather`` to one rank or an ``all_gather``
(RCCL over xGMI with the ``nccl`` backend; 44 bytes per game for 4-player Tron, latency-bound).  The reference has no counterpart (it has no
device code at all).

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


def gather_results(local: torch.Tensor, total: int, group: Optional[dist.ProcessGroup] = None,
                   out: Optional[torch.Tensor] = None, dst: Optional[int] = None) -> Optional[torch.Tensor]:
    """Concatenate per-game result rows [n_local, k] of all ranks in global game order: on every rank (``dst=None``,
    one ``all_gather_into_tensor``) or on rank `dst` only (one ``gather``; the other ranks get ``None``).

    ONE collective either way; shards are padded to the largest shard so a single call moves everything.  The gather
    to one rank is what an episode-end statistics sink needs and what bench.py times: over RCCL it is a set of
    point-to-point transfers, so on the fully connected xGMI mesh of an 8-GPU node the 7 shards travel over 7 different
    links at once (~40 us for 2.9 MB each), where a ring all_gather needs 7 serial hops.  The collective runs whenever
    a process group is initialised -- also for a world of one rank (``torchrun --nproc-per-node 1``), so the RCCL path
    is the same code at every world size.  `out` (``[world * ceil(total / world), k]``, same dtype / device) is an
    optional preallocated receive buffer."""
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
        dist.gather(padded, pieces, dst=dst, group=group)
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
    """Random-agent rollouts of `global_batch` games spread over the ranks of the default process group.

    make_stepper(batch=n, first_env_id=lo) must return an object with ``rollout(steps, seed)`` and
    ``results() -> tensor [n, k]`` (``TronBatch`` / ``TTTBatch`` / ``BlokusBatch`` partials).
    """

    def __init__(self, make_stepper: Callable, global_batch: int, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.total = int(global_batch)
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        self.lo, self.hi = shard_bounds(self.total, self.rank, self.world)
        self.stepper = make_stepper(batch=self.hi - self.lo, first_env_id=self.lo)
        self._recv = None                      # receive buffer of the gather, allocated once

    def rollout(self, steps: int, seed: int = 0, chunk: int = 512) -> int:
        """Exactly `steps` env-steps for every local game, as fused launches of <= chunk steps."""
        launches, left = 0, int(steps)
        while left > 0:
            t = min(int(chunk), left)
            self.stepper.rollout(t, seed)
            left -= t
            launches += 1
        return launches

    def gather(self, dst: Optional[int] = None) -> Optional[torch.Tensor]:
        """Per-game results of ALL games (the one collective of a rollout): on every rank, or with `dst` on that rank
        only (``None`` elsewhere)."""
        local = self.stepper.results()
        receiver = dst is None or self.rank == dst
        if self._recv is None and receiver and dist.is_available() and dist.is_initialized():
            widest = -(-self.total // self.world)
            self._recv = torch.empty((self.world * widest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        return gather_results(local, self.total, self.group, out=self._recv, dst=dst)
