"""Host-side player record of a Blokus state (shape of reference colosseumrl/envs/blokus/ai.py:25-54)."""
from .actions import PIECE_NAMES, PIECE_VALUES

GAME_PIECE_VALUES = dict(zip(PIECE_NAMES, PIECE_VALUES))


class AI:
    """``player_score``, ``player_color`` (1..4) and ``current_pieces`` (ordered list of piece names)."""

    def __init__(self, board_state=None, color=1):
        self.player_score = 0
        self.player_color = color
        self.current_pieces = list(PIECE_NAMES)

    def inventory_mask(self) -> int:
        return sum(1 << PIECE_NAMES.index(p) for p in self.current_pieces)
