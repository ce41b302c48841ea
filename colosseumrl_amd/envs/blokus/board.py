"""Host-side board record of a Blokus state (reference colosseumrl/envs/blokus/board.py:75-227).

``board_contents`` is what agents and renderers read.  The methods a reference agent may call on the record it is handed
-- ``get_all_valid_moves``, ``gather_empty_corner_indexes``, ``check_valid_corner``, ``update_board``, ``calculate_winner``
... -- are kept with the reference's signatures and return formats; move generation itself runs in the HIP kernels
(``crl_blokus_valid_list`` / ``crl_blokus_fits`` on host-mapped memory, one launch per call), never on the CPU.
"""
import threading
from collections import defaultdict

import numpy as np

from .actions import ORIENTATIONS, PIECE_INDEX, PIECE_NAMES, PIECE_VALUES, decode, encode  # noqa: F401  (re-exported like the reference module)

PLAYER_DEFAULT_CORNERS = [(0, 0), (19, 0), (0, 19), (19, 19)]      # (x, y) per player, reference board.py:50

# rotation of an (x, y) index about the board centre into / out of a player's viewpoint (reference :52-73)
BOARD_TO_PLAYER_OBSERVATION_ROTATION_MATRICES = np.array(
    [[[1, 0], [0, 1]], [[0, -1], [1, 0]], [[-1, 0], [0, -1]], [[0, 1], [-1, 0]]], dtype=np.int32)
PLAYER_OBSERVATION_TO_BOARD_ROTATION_MATRICES = np.array(
    [[[1, 0], [0, 1]], [[0, 1], [-1, 0]], [[-1, 0], [0, -1]], [[0, -1], [1, 0]]], dtype=np.int32)

_local = threading.local()


def _stager():
    """This thread's single-state Blokus stepper (host-mapped block + stream), created at first use.  Raises without a GPU."""
    st = getattr(_local, "stager", None)
    if st is None:
        from ...single import SingleBlokus
        st = _local.stager = SingleBlokus()
    return st


def _legal_ids(board_contents, round_count: int, player_color: int, piece_names) -> np.ndarray:
    """Dense ids of every legal action of colour `player_color` holding `piece_names`, ascending = reference order."""
    q = int(player_color) - 1
    inv = [0, 0, 0, 0]
    inv[q] = sum({1 << PIECE_INDEX[p] for p in piece_names})
    st = _stager()
    st.load(board_contents, inv, [0, 0, 0, 0], int(round_count), q)
    return st.legal_ids(q)


def placement_cells(piece_type: str, orientation: str):
    """(dx, dy) of every cell of `piece_type` in orientation+shift `orientation` (e.g. ``'northwest3'``) relative to the
    index cell -- what the reference derives with shift_offsets + rotate_piece (board.py:93-98).  Host-side table lookup
    (``crl_blokus_placement``)."""
    import ctypes as C
    from ... import _native
    cells = (C.c_int8 * 10)()
    piece = PIECE_INDEX[piece_type]                                  # KeyError first, as PIECE_TYPES[piece_type] in board.py:93
    shift = int(orientation[-1])
    name = orientation[:-1]
    o = ORIENTATIONS.index(name) if name in ORIENTATIONS else 2      # any other name: rotate_piece's default branch = east
    if shift >= PIECE_VALUES[piece]:
        raise IndexError("index %d is out of bounds for axis 0 with size %d" % (shift, PIECE_VALUES[piece]))
    n = _native.lib().crl_blokus_placement(piece, o, shift, cells)
    if n < 0:
        _native.check(n, "crl_blokus_placement")
    return [(int(cells[2 * j]), int(cells[2 * j + 1])) for j in range(n)]


class Board:
    """``board_contents[y][x]``: 0 empty, colour 1..4 = player + 1 (int64, 20 x 20)."""

    def __init__(self, copy_from_board=None):
        self.reset_board(copy_from_board)

    def reset_board(self, copy_from_board=None):
        if copy_from_board is not None:
            self.board_contents = np.array(copy_from_board.board_contents, dtype=np.int64, copy=True)
        else:
            self.board_contents = np.zeros((20, 20), dtype=np.int64)

    # ---- placing (reference :87-103): no legality test here either, numpy's own index rules apply
    def update_board(self, player_color, piece_type, index, piece_orientation, round_count, ai_game):
        self.player_color = player_color
        for dx, dy in placement_cells(piece_type, piece_orientation):
            self.place_piece(index[0] + dx, index[1] + dy)

    def place_piece(self, x, y):
        self.board_contents[y][x] = self.player_color

    # ---- anchors (reference :114-154)
    def gather_empty_corner_indexes(self, player_color):
        """``[(x, y), ...]`` row-major: empty cells diagonal to a cell of `player_color` and not orthogonally next to one.
        They are the cells where the monomino is legal from round 1 on -- which is how the GPU is asked."""
        ids = _legal_ids(self.board_contents, 1, player_color, ["monomino1"])
        return [decode(int(i))[1:3] for i in ids[::8]]            # eight orientations of the one-cell piece per anchor

    def check_valid_corner(self, board_contents, player_color, row_num, col_num):
        """No orthogonal neighbour of `player_color` and at least one diagonal one (reference :127-154).  As there, the cell
        ITSELF is not looked at -- gather_empty_corner_indexes tests emptiness before it calls this (:121) -- so an occupied
        cell can answer True.  Eight array reads on the record the caller holds: a host-side record method, no launch."""
        b = board_contents
        r, c = row_num, col_num
        if (r != 0 and b[r - 1][c] == player_color) or (c != 0 and b[r][c - 1] == player_color) \
                or (r != 19 and b[r + 1][c] == player_color) or (c != 19 and b[r][c + 1] == player_color):
            return False
        return bool((r != 0 and c != 19 and b[r - 1][c + 1] == player_color) or (r != 0 and c != 0 and b[r - 1][c - 1] == player_color)
                    or (r != 19 and c != 19 and b[r + 1][c + 1] == player_color) or (r != 19 and c != 0 and b[r + 1][c - 1] == player_color))

    def check_orientation_shifts(self, player_color, piece_type, index, orientation):
        """Shift ids k (ascending, int64 array) for which `piece_type` in `orientation` fits with its cell k on `index`:
        every cell on the board, empty, no orthogonal neighbour of the player's colour -- whether or not `index` is an anchor
        (reference :156-168, computation.py:144-180).  One ``crl_blokus_fits`` launch per shift."""
        x, y = int(index[0]), int(index[1])
        if not (0 <= x < 20 and 0 <= y < 20):
            return np.zeros(0, dtype=np.int64)
        q, piece, o = int(player_color) - 1, PIECE_INDEX[piece_type], ORIENTATIONS.index(orientation)
        st = _stager()
        st.load(self.board_contents, [0, 0, 0, 0], [0, 0, 0, 0], 1, q)
        return np.array([k for k in range(PIECE_VALUES[piece]) if st.fits(q, encode(piece, x, y, o, k))], dtype=np.int64)

    # ---- move generation (reference :170-193)
    def get_all_valid_moves(self, round_count, player_color, player_pieces):
        """``{piece: {(x, y): [orientation+shift, ...]}}``: pieces in the order of `player_pieces`, anchors row-major,
        orientations in ``ORIENTATIONS`` order, shifts ascending; pieces without a legal placement are left out."""
        found = {}
        for i in _legal_ids(self.board_contents, round_count, player_color, player_pieces):
            piece, x, y, o, k = decode(int(i))
            found.setdefault(PIECE_NAMES[piece], defaultdict(list))[(x, y)].append(ORIENTATIONS[o] + str(k))
        return {p: found[p] for p in player_pieces if p in found}

    # ---- scoring (reference :195-227)
    def decode_color(self, player_color):
        return {1: "R", 2: "B", 3: "G", 4: "Y"}[player_color]

    def calculate_winner(self, players, round_count):
        """Colour letter of the best score; with a tie, of the tied player the reference's stable sort by score visits
        last (so with nobody on the score sheet yet: the last player's, never the initial ``"NONE"``; reference :195-214)."""
        best = max([p.player_score for p in players] + [0])
        winner = "NONE"
        for p in sorted(players, key=lambda p: p.player_score):
            if p.player_score == best:
                winner = self.decode_color(p.player_color)
        return winner
