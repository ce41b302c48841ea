"""Blokus state pickles under the reference's class paths (opt-in, host-only)."""
import pickle
import pickletools

import numpy as np
import pytest

from colosseumrl_amd import compat
from colosseumrl_amd.envs.blokus.ai import AI
from colosseumrl_amd.envs.blokus.board import Board
from colosseumrl_amd.envs.blokus.BlokusEnvironment import BlokusEnvironment


def _state():
    board = Board()
    board.board_contents[3, 4] = 2
    players = [AI(board, c) for c in (1, 2, 3, 4)]
    players[1].current_pieces.remove("domino1")
    players[1].player_score = 2
    return board, 5, players


def _globals(data):
    """(module, name) pairs a pickle refers to."""
    out, strings = set(), []
    for op, arg, _ in pickletools.genops(data):
        if op.name == "GLOBAL":
            out.add(tuple(arg.split(" ")))
        elif op.name in ("SHORT_BINUNICODE", "BINUNICODE", "UNICODE"):
            strings.append(arg)
        elif op.name == "STACK_GLOBAL":
            out.add((strings[-2], strings[-1]))
    return out


def test_default_pickles_use_this_package():
    assert not compat.enabled()
    data = BlokusEnvironment.serialize_state(_state())
    assert ("colosseumrl_amd.envs.blokus.board", "Board") in _globals(data)


_ALIAS_SCRIPT = r"""
import pickle, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
from test_wire_format import _state, _globals
from colosseumrl_amd import compat
from colosseumrl_amd.envs.blokus.ai import AI
from colosseumrl_amd.envs.blokus.board import Board
from colosseumrl_amd.envs.blokus.BlokusEnvironment import BlokusEnvironment
assert compat.reference_wire_format(True) is False          # stand-in mode: no real package in this interpreter
data = BlokusEnvironment.serialize_state(_state())
g = _globals(data)
assert ("colosseumrl.envs.blokus.board", "Board") in g and ("colosseumrl.envs.blokus.ai", "AI") in g, g
assert not any(m.startswith("colosseumrl_amd") for m, _ in g), g
board, rnd, players = BlokusEnvironment.deserialize_state(data)
assert isinstance(board, Board) and rnd == 5 and board.board_contents[3, 4] == 2
assert players[1].player_score == 2 and "domino1" not in players[1].current_pieces
assert [p.player_color for p in players] == [1, 2, 3, 4]
# a pickle written by "the other side" (plain pickle, reference paths) loads as this package's records
b2, r2, p2 = BlokusEnvironment.deserialize_state(pickle.dumps((board, 7, players)))
assert r2 == 7 and isinstance(p2[0], AI) and np.array_equal(b2.board_contents, board.board_contents)
compat.reference_wire_format(False)
assert Board.__module__ == "colosseumrl_amd.envs.blokus.board" and not compat.enabled()
assert ("colosseumrl_amd.envs.blokus.board", "Board") in _globals(BlokusEnvironment.serialize_state(_state()))
print("ok")
"""


def test_reference_class_paths_round_trip():
    """Stand-in mode (no reference package importable): run in a fresh interpreter so that neither a reference tree
    loaded by other tests nor the aliases registered here leak between tests."""
    import os
    import subprocess
    import sys
    tests = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, "-c", _ALIAS_SCRIPT % {"root": os.path.dirname(tests), "tests": tests}],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
