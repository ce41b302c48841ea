"""Host-side player record of a Blokus state (reference colosseumrl/envs/blokus/ai.py:25-54): score, colour, inventory,
and the three methods of the reference's record -- the move lists come from the board record, i.e. from the GPU."""
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

    def collect_moves(self, board, round_count):
        return board.get_all_valid_moves(round_count, self.player_color, self.current_pieces)

    def check_moves(self, board, round_count):
        """True iff this player has a legal move; the moves stay in ``all_valid_moves`` as in the reference (:36-43)."""
        self.all_valid_moves = self.collect_moves(board, round_count)
        return len(self.all_valid_moves) > 0

    def update_player(self, piece_type):
        """Inventory and score after playing `piece_type`: its cells, +20 when the last piece is the monomino, +15 for any
        other last piece (:45-54)."""
        self.current_pieces.remove(piece_type)
        if not self.current_pieces:
            self.player_score += 20 if piece_type == "monomino1" else 15
        self.player_score += GAME_PIECE_VALUES[piece_type]
