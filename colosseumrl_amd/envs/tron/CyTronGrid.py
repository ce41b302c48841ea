"""Stand-in for the reference's ONLY native module, ``colosseumrl/envs/tron/CyTronGrid`` (CyTronGrid.pyx:3-7, :65).

The same two functions with the same signatures -- C-contiguous int64 numpy arrays, mutated in place, nothing returned --
evaluated on the GPU through the C ABI's entries of the same shape (``crl_tron_next_state_inplace64_host``,
``crl_tron_relative_player_inplace64``).  A reference checkout that wants its Tron rules from this library changes one import
(``TronGridEnvironment.py:9``)::

    from colosseumrl_amd.envs.tron.CyTronGrid import next_state_inplace, relative_player_inplace

and nothing else; ``colosseumrl_amd``'s own ``TronGridEnvironment`` calls the same entries.  The arrays may live anywhere: each
call copies them through a block of host memory the GPU maps (one per thread and board shape, created on first use) and
back.  Values are the Cython function's, also where its C arithmetic shows (``cdivision=True``: negative directions for
action sums below -4, observer ids of any size) -- ``tests/golden/tron_edge.npz``, ``tron_wild64.npz``, ``tron_observe_*.npz``
are calls of the reference's functions, replayed through this module in ``tests/test_gpu_tron.py``.

No CPU path: without an MI355X the first call raises.
"""
import threading

import numpy as np

_local = threading.local()          # the mapped block is not re-entrant (ctypes releases the GIL): one stepper per thread


def _stepper(n: int, p: int):
    cache = getattr(_local, "steppers", None)
    if cache is None:
        cache = _local.steppers = {}
    st = cache.get((n, p))
    if st is None:
        from ...single import SingleTron
        st = cache[(n, p)] = SingleTron(n, p, list(range(p)), [0] * p)      # (the spawn layout matters to resets only)
    return st


def next_state_inplace(board: np.ndarray, heads: np.ndarray, directions: np.ndarray, deaths: np.ndarray,
                       actions: np.ndarray) -> None:
    """One simultaneous move of every living player, in the reference's sequential order (CyTronGrid.pyx:15-62).
    ``board`` int64 [N, N], the others int64 [P]; ``actions`` in {0 forward, +1 right, -1 left}.  All four state arrays are
    updated in place."""
    st = _stepper(board.shape[0], heads.shape[0])
    st.next_state64(board, heads, directions, deaths, actions, False)
    v = st.s64
    board[...] = v["board"]
    heads[...] = v["heads"]
    directions[...] = v["dirs"]
    deaths[...] = v["deaths"]


def relative_player_inplace(board: np.ndarray, num_players: int, player: int) -> None:
    """``board[i, j] > 0  ->  ((board[i, j] - player + num_players) % num_players) + 1`` with C's remainder, in place
    (CyTronGrid.pyx:65-71; ``player`` as the reference passes it: the observer's id + 1)."""
    st = _stepper(board.shape[0], min(max(int(num_players), 1), 8))
    board[...] = st.relative_board64(board, int(player), int(num_players))
