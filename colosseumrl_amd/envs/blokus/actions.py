"""Blokus action strings <-> dense action ids.

Reference format (colosseumrl/envs/blokus/BlokusEnvironment.py:55-106):
``"{piece};({x}, {y});{orientation}{shift}"``, e.g. ``'monomino1;(0, 0);north0'``; ``''`` is a pass.
Dense id: ``((piece*400 + y*20 + x)*8 + orientation)*5 + shift`` -- ascending ids enumerate actions in
exactly the order of the reference's ``valid_actions`` (piece in inventory order -> anchor row-major ->
orientation in ``ORIENTATIONS`` order -> shift ascending).
"""
from typing import Tuple, Union

# inventory order (reference ai.py:12-22 / board.py:24-44) and the cell count = score value of each piece
PIECE_NAMES = ["monomino1", "domino1", "trominoe1", "trominoe2", "tetrominoes1", "tetrominoes2", "tetrominoes3",
               "tetrominoes4", "tetrominoes5", "pentominoe1", "pentominoe2", "pentominoe3", "pentominoe4",
               "pentominoe5", "pentominoe6", "pentominoe7", "pentominoe8", "pentominoe9", "pentominoe10",
               "pentominoe11", "pentominoe12"]
PIECE_VALUES = [1, 2, 3, 3, 4, 4, 4, 4, 4] + [5] * 12
PIECE_INDEX = {name: i for i, name in enumerate(PIECE_NAMES)}
# clockwise, reference board.py:47
ORIENTATIONS = ["north", "northeast", "east", "southeast", "south", "southwest", "west", "northwest"]
ORIENTATION_INDEX = {name: i for i, name in enumerate(ORIENTATIONS)}
PASS = -1


def action_to_string(piece_type: str, index: Tuple[int, int], orientation: str) -> str:
    """``(piece, (x, y), orientation+shift)`` -> action string, with plain Python ints in the index."""
    return "{};({}, {});{}".format(piece_type, int(index[0]), int(index[1]), orientation)


def string_to_action(action_str: str) -> Union[Tuple[str, Tuple[int, int], str], None]:
    """Action string -> ``(piece, (x, y), orientation+shift)``; ``''`` -> None (reference :83-106)."""
    if action_str == "":
        return None
    piece_type, index, orientation = action_str.split(";")
    index = tuple(map(int, index.replace("(", "").replace(")", "").split(",")))
    return piece_type, index, orientation


def separate_offset_from_orientation(orientation_string: str) -> Tuple[str, str]:
    """``'northwest3'`` -> ``('northwest', '3')`` (reference :33-43)."""
    letters = "".join(c for c in orientation_string if not c.isdigit())
    digits = "".join(c for c in orientation_string if c.isdigit())
    return letters, digits


def encode(piece: int, x: int, y: int, orientation: int, shift: int) -> int:
    return ((piece * 400 + y * 20 + x) * 8 + orientation) * 5 + shift


def decode(action_id: int) -> Tuple[int, int, int, int, int]:
    """id -> (piece, x, y, orientation, shift)"""
    shift = action_id % 5
    orientation = (action_id // 5) % 8
    cell = (action_id // 40) % 400
    return action_id // 16000, cell % 20, cell // 20, orientation, shift


_STRINGS = {}          # id -> string, filled as ids are seen (a valid_actions list is ~100-1,700 strings per call)


def id_to_string(action_id: int) -> str:
    action_id = int(action_id)
    s = _STRINGS.get(action_id)
    if s is None:
        if action_id < 0:
            return ""
        piece, x, y, orientation, shift = decode(action_id)
        s = _STRINGS[action_id] = "{};({}, {});{}{}".format(PIECE_NAMES[piece], x, y, ORIENTATIONS[orientation], shift)
    return s


def ids_to_strings(ids) -> list:
    """Strings of a whole id list (ascending ids = the reference's valid_actions order)."""
    get = _STRINGS.get
    out = []
    for i in ids.tolist() if hasattr(ids, "tolist") else ids:
        s = get(i)
        out.append(s if s is not None else id_to_string(i))
    return out


EXT_BASE = 336000                     # CRL_BLOKUS_EXT_BASE: ids whose index lies anywhere in [-20, 20) x [-20, 20)
INDEX_ERROR, VALUE_ERROR, BAD_ACTION = -1, -2, -3      # CRL_BLOKUS_*: what the step puts into the reward slot


def string_to_step_id(action_str: str) -> int:
    """The id ``crl_blokus_step`` takes for an action string ``next_state`` was handed -- ANY string, listed by
    ``valid_actions`` or not -- raising what the reference raises before it touches the board, in its order: ``split`` /
    ``int`` of string_to_action (ValueError, BlokusEnvironment.py:83-106), the piece name (KeyError, board.py:93), the shift
    digit ``int(orientation[-1])`` (IndexError on an empty field, ValueError on a non-digit), a shift that names no cell of
    the piece (IndexError, computation.py:218).  An orientation NAME that is none of the eight places the piece as 'east',
    rotate_piece's default branch (computation.py:85-86).  An index outside [-20, 20) raises the IndexError numpy would
    (board.py:103: the index cell itself is always written); cells of the piece that leave that range, and a piece the
    mover does not hold, are found by the kernel (codes INDEX_ERROR / VALUE_ERROR in the reward slot)."""
    parsed = string_to_action(action_str)
    if parsed is None:
        return PASS
    piece_type, index, orientation = parsed
    piece = PIECE_INDEX[piece_type]
    shift = int(orientation[-1])
    o = ORIENTATION_INDEX.get(orientation[:-1], 2)
    if shift >= PIECE_VALUES[piece]:
        raise IndexError("index %d is out of bounds for axis 0 with size %d" % (shift, PIECE_VALUES[piece]))
    x, y = index[0], index[1]
    for v in (y, x):                                  # board_contents[y] is indexed first
        if not -20 <= v < 20:
            raise IndexError("index %d is out of bounds for axis 0 with size 20" % v)
    if 0 <= x < 20 and 0 <= y < 20:
        return encode(piece, x, y, o, shift)
    return EXT_BASE + ((piece * 1600 + (y + 20) * 40 + (x + 20)) * 8 + o) * 5 + shift


def string_to_id(action_str: str) -> int:
    """Action string -> dense id.  The shift is the LAST character of the orientation field, as in the
    reference's update_board (board.py:93); unknown pieces / orientations raise KeyError / ValueError."""
    parsed = string_to_action(action_str)
    if parsed is None:
        return PASS
    piece_type, (x, y), orientation = parsed
    return encode(PIECE_INDEX[piece_type], x, y, ORIENTATION_INDEX[orientation[:-1]], int(orientation[-1]))
