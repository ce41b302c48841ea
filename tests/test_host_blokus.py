"""CPU-only checks of the Blokus host layer against fixtures captured from the reference."""
import numpy as np
import pytest

from colosseumrl_amd.envs.blokus import actions as A
from colosseumrl_amd.envs.blokus.BlokusEnvironment import BlokusEnvironment
from colosseumrl_amd.envs.blokus.ai import AI
from colosseumrl_amd.envs.blokus.board import Board


def _state(board, inv, score, rnd):
    b = Board()
    b.board_contents = np.asarray(board, dtype=np.int64)
    players = []
    for c in range(4):
        ai = AI(b, c + 1)
        ai.player_score = int(score[c])
        ai.current_pieces = [n for i, n in enumerate(A.PIECE_NAMES) if (int(inv[c]) >> i) & 1]
        players.append(ai)
    return b, int(rnd), players


def test_action_codec_roundtrip():
    assert A.id_to_string(0) == "monomino1;(0, 0);north0"
    assert A.string_to_id("pentominoe12;(2, 2);northwest4") == A.encode(20, 2, 2, 7, 4)
    for aid in (0, 1, 39, 40, 15999, 16000, 335999, 123457):
        assert A.string_to_id(A.id_to_string(aid)) == aid
    assert A.string_to_id("") == -1 and A.id_to_string(-1) == ""
    assert A.string_to_action("domino1;(19, 0);south0") == ("domino1", (19, 0), "south0")
    assert A.action_to_string("domino1", (np.int32(19), np.int64(0)), "south0") == "domino1;(19, 0);south0"
    assert sum(A.PIECE_VALUES) == 89 and len(A.PIECE_NAMES) == 21


def test_perspective_conversion_matches_reference(golden):
    g = golden("blokus_observe")
    env = BlokusEnvironment()
    for aid, pl, out, back in zip(g["conv_in"], g["conv_player"], g["conv_out"], g["conv_back"]):
        s = env.convert_real_action_to_player_perspective_action(A.id_to_string(int(aid)), int(pl))
        assert A.string_to_id(s) == int(out)
        assert A.string_to_id(env.convert_player_perspective_action_to_real_action(s, int(pl))) == int(back) == int(aid)
    assert env.convert_real_action_to_player_perspective_action("", 2) == ""


def test_new_state_shape():
    env = BlokusEnvironment()
    (board, rnd, players), movers = env.new_state()
    assert board.board_contents.shape == (20, 20) and board.board_contents.dtype == np.int64 and rnd == 0 and movers == [0]
    assert [p.player_color for p in players] == [1, 2, 3, 4] and players[0].current_pieces == A.PIECE_NAMES
    assert env.current_rewards((board, rnd, players)) == [0, 0, 0, 0]
    blob = env.serialize_state((board, rnd, players))
    b2, r2, p2 = env.deserialize_state(blob)
    assert np.array_equal(b2.board_contents, board.board_contents) and p2[3].current_pieces == A.PIECE_NAMES


def test_record_methods_without_gpu_work_match_reference(golden):
    """Board.update_board / place_piece / decode_color / calculate_winner and AI.update_player against the reference's own
    records (tests/golden/blokus_records.npz): pure host bookkeeping, no GPU."""
    from colosseumrl_amd.envs.blokus.actions import ORIENTATIONS, PIECE_NAMES
    from colosseumrl_amd.envs.blokus.ai import AI
    from colosseumrl_amd.envs.blokus.board import Board
    g = golden("blokus_records")
    for spec, want in zip(g["placed_spec"], g["placed"]):
        piece, x, y, o, k, color = [int(v) for v in spec]
        b = Board()
        b.update_board(color, PIECE_NAMES[piece], (x, y), ORIENTATIONS[o] + str(k), 3, True)
        assert np.array_equal(b.board_contents, want), spec
        assert b.board_contents.dtype == np.int64 and b.player_color == color
    for order, want in zip((PIECE_NAMES[::-1], PIECE_NAMES), g["update_scores"]):
        a = AI(None, 1)
        got = []
        for p in order:
            a.update_player(p)
            got.append(a.player_score)
        assert got == want.tolist() and a.current_pieces == []
    assert g["update_scores"][0][-1] == 109 and g["update_scores"][1][-1] == 104           # +20 / +15
    for case in g["winner_cases"]:
        ps = [AI(None, c) for c in (1, 2, 3, 4)]
        for p, v in zip(ps, case[:4]):
            p.player_score = int(v)
        assert Board().calculate_winner(ps, 20) == ["R", "B", "G", "Y", "NONE"][int(case[4])], case
    copy = Board(Board())
    copy.board_contents[3, 4] = 2
    again = Board(copy)
    assert again.board_contents[3, 4] == 2 and again.board_contents is not copy.board_contents
    again.reset_board()
    assert not again.board_contents.any()
    with pytest.raises(ValueError):
        AI(None, 2).update_player("no such piece")


def test_step_id_of_any_action_string_reference_golden(golden):
    """A.string_to_step_id on the 497 strings of blokus_illegal.npz: where the reference's next_state raises before it touches
    the board (split / int / piece name / shift digit / shift without a cell / index outside numpy's range) the same class is
    raised; otherwise the id equals the oracle's own reading of the string (dense, or extended for an index off the board)."""
    from oracle import oracle as O
    from blokus_replay import illegal_ids
    g, keep, ids = illegal_ids(golden, A.string_to_step_id)
    g2, keep2, ids2 = illegal_ids(golden, O.blokus_step_action_id)
    assert np.array_equal(keep, keep2) and np.array_equal(ids, ids2) and len(keep) == 364
    assert A.string_to_step_id("") == A.PASS == -1
    assert A.string_to_step_id("domino1;(5, 5);foo1") == A.encode(1, 5, 5, 2, 1)                  # unknown name = east
    assert A.string_to_step_id("domino1;(-1, 5);east0") == A.EXT_BASE + ((1 * 1600 + 25 * 40 + 19) * 8 + 2) * 5
    for s, exc in (("zzz;(0, 0);", KeyError), ("domino1;(3, 3);", IndexError), ("domino1;(3,3);eastx", ValueError),
                   ("domino1;(3, 3)", ValueError), ("domino1;(5, 5);east3", IndexError), ("monomino1;(25, 5);east0", IndexError),
                   ("monomino1;(5, -21);east0", IndexError)):
        with pytest.raises(exc):
            A.string_to_step_id(s)


def test_update_board_numpy_rules_and_check_valid_corner_reference_golden(golden):
    """Board.update_board called on the record itself: cells go down one by one under numpy's index rules, so a piece hanging
    over the left / top edge wraps, one over the right / bottom edge raises IndexError AFTER its earlier cells were written,
    an unknown orientation name places 'east' (board.py:87-103).  Board.check_valid_corner on all 1,600 (colour, cell) pairs
    of three states, occupied cells included (board.py:127-154 does not look at the cell)."""
    g = golden("blokus_illegal")
    seen = set()
    for spec, exc, want in zip(g["ub_spec"], g["ub_exc"], g["ub_board"]):
        piece, x, y, o, k, color = [int(v) for v in spec]
        b = Board()
        b.board_contents[:] = g["base_board"][1]
        name = (A.ORIENTATIONS[o] if o >= 0 else "bogus") + str(k)
        try:
            b.update_board(color, A.PIECE_NAMES[piece], (x, y), name, 3, True)
            got = 0
        except IndexError:
            got = 1
        seen.add(got)
        assert got == int(exc) and np.array_equal(b.board_contents, want), spec
    assert seen == {0, 1}
    probe = Board()
    n_true = n_occupied = 0
    for k in range(len(g["base_board"])):
        bc = g["base_board"][k].astype(np.int64)
        for c in (1, 2, 3, 4):
            for y in range(20):
                for x in range(20):
                    got = probe.check_valid_corner(bc, c, y, x)
                    assert got == bool(g["corner_grid"][k, c - 1, y, x]), (k, c, y, x)
                    n_true += got
                    n_occupied += got and bc[y, x] != 0
    assert n_true == 134 and n_occupied == 31
