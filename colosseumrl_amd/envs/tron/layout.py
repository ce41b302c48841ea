"""Tron spawn layout (host side, computed once per environment instance).

Restates ``TronGridEnvironment.generate_start_positions``
(reference colosseumrl/envs/tron/TronGridEnvironment.py:183-226): a square ring
``ring_offset`` cells in from the wall is walked as top row, strided "right",
reversed bottom row, reversed strided "left"; the walk is split into P sections
and every player spawns at the section centre plus its spawn offset, facing
``(side_index + 2) % 4``.  The strided slices of the reference do not select the
columns their names suggest and ``4 * side`` differs from the ring length; both
are part of the observable behaviour and are kept.
"""
from functools import lru_cache
from typing import List, Sequence, Tuple


def _split_sizes(length: int, parts: int) -> List[int]:
    # numpy.array_split: the first length % parts chunks get one extra element
    base, extra = divmod(length, parts)
    return [base + (1 if i < extra else 0) for i in range(parts)]


@lru_cache(maxsize=256)
def ring_walk(board_size: int, ring_offset: int) -> Tuple[List[int], int]:
    """Ordered ring cells (flat indices) and the nominal side length.  (A pure function of two integers, cached: the N*N
    scan is most of what a drop-in ``new_state`` costs -- the reference rebuilds its ogrid masks on every call.)"""
    n = board_size
    half, odd = divmod(n, 2)
    inner, outer = half - ring_offset - 1, half - ring_offset
    side = 2 * (inner + 1)
    if side <= 0:
        raise ValueError("ring_offset %d leaves no spawn ring on a %dx%d board" % (ring_offset, n, n))
    # doubled coordinates keep the even-N half-integers integral
    coord2 = [2 * (k - half) + (0 if odd else 1) for k in range(n)]
    ring = [y * n + x
            for y in range(n) for x in range(n)
            if (max(abs(coord2[x]), abs(coord2[y])) <= 2 * outer) != (max(abs(coord2[x]), abs(coord2[y])) <= 2 * inner)]
    top = ring[:side]
    right = ring[side:3 * side:2]
    bottom = ring[3 * side:]
    left = ring[side + 1:3 * side + 1:2]
    return top + right + bottom[::-1] + left[::-1], side


def start_positions(board_size: int, num_players: int, ring_offset: int = 1,
                    offsets: Sequence[int] = None) -> Tuple[List[int], List[int]]:
    """(heads, directions) for every player; ``offsets[p]`` is the player's spawn offset."""
    if offsets is None:
        offsets = [2] * num_players
    walk, side = ring_walk(board_size, ring_offset)
    facing = [((i // side) + 2) % 4 for i in range(4 * side)]
    heads, dirs = [], []
    pos_w = pos_f = 0
    for p, (len_w, len_f) in enumerate(zip(_split_sizes(len(walk), num_players),
                                           _split_sizes(len(facing), num_players))):
        if len_w == 0 or len_f == 0:
            raise ValueError("board %d too small for %d players" % (board_size, num_players))
        iw = min(max(len_w // 2 + offsets[p], 0), len_w - 1)
        jf = min(max(len_f // 2 + offsets[p], 0), len_f - 1)
        heads.append(walk[pos_w + iw])
        dirs.append(facing[pos_f + jf])
        pos_w += len_w
        pos_f += len_f
    return heads, dirs
