"""Host-side board record of a Blokus state (shape of reference colosseumrl/envs/blokus/board.py:75-103).

Only the data members agents and renderers read are kept (``board_contents``); move generation lives in
the HIP kernels.
"""
import numpy as np

from .actions import ORIENTATIONS, PIECE_NAMES  # noqa: F401  (re-exported like the reference module)

PLAYER_DEFAULT_CORNERS = [(0, 0), (19, 0), (0, 19), (19, 19)]      # (x, y) per player, reference board.py:50

# rotation of an (x, y) index about the board centre into / out of a player's viewpoint (reference :52-73)
BOARD_TO_PLAYER_OBSERVATION_ROTATION_MATRICES = np.array(
    [[[1, 0], [0, 1]], [[0, -1], [1, 0]], [[-1, 0], [0, -1]], [[0, 1], [-1, 0]]], dtype=np.int32)
PLAYER_OBSERVATION_TO_BOARD_ROTATION_MATRICES = np.array(
    [[[1, 0], [0, 1]], [[0, 1], [-1, 0]], [[-1, 0], [0, -1]], [[0, -1], [1, 0]]], dtype=np.int32)


class Board:
    """``board_contents[y][x]``: 0 empty, colour 1..4 = player + 1 (int64, 20 x 20)."""

    def __init__(self, copy_from_board=None):
        if copy_from_board is not None:
            self.board_contents = np.array(copy_from_board.board_contents, dtype=np.int64, copy=True)
        else:
            self.board_contents = np.zeros((20, 20), dtype=np.int64)
