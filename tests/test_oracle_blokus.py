"""CPU restatement of Blokus vs games played by the reference itself + SURVEY 8c known answers."""
import numpy as np

from oracle import oracle as O
from backends import OracleBlokus
from blokus_replay import GAMES, check_bonus, check_illegal, replay_games


def test_piece_table_kat():
    # 21 pieces / 89 cells; 712 (piece, orientation, shift) combos of which 414 are distinct cell sets
    sizes = [len(O.blokus_placement(p, 2, 0)) for p in range(21)]
    assert sum(sizes) == 89 and sizes[:9] == [1, 2, 3, 3, 4, 4, 4, 4, 4] and all(s == 5 for s in sizes[9:])
    combos, n = set(), 0
    for p in range(21):
        for o in range(8):
            for k in range(sizes[p]):
                cells = O.blokus_placement(p, o, k)
                assert (0, 0) in set(map(tuple, cells.tolist()))            # the shift cell sits on the anchor
                combos.add((p, tuple(sorted(map(tuple, cells.tolist())))))
                n += 1
    assert n == 712 and len(combos) == 414
    # orientation map of SURVEY B2 on the offset (2, -1): north (dy,-dx) ... northwest (-dy,-dx)
    want = {0: (-1, -2), 1: (2, 1), 2: (2, -1), 3: (-1, 2), 4: (1, 2), 5: (-2, -1), 6: (-2, 1), 7: (1, -2)}
    for o, w in want.items():
        assert tuple(O.blokus_placement(7, o, 0)[3]) == w          # tetrominoes4's 4th cell is (2, -1)


def test_opening_kat():
    # 116 opening actions for every player; the four scripted openers
    st = O.BlokusState(1)
    for pl in range(4):
        cnt, ids = O.blokus_valid(st, player=np.array([pl], np.int8), cap=256)
        assert cnt[0] == 116
    cnt, ids = O.blokus_valid(st, cap=256)
    assert O.blokus_action_string(ids[0, 0]) == "monomino1;(0, 0);north0"
    for s in ["trominoe1;(0, 0);east0", "domino1;(19, 0);south0", "monomino1;(0, 19);east0", "tetrominoes3;(19, 19);west0"]:
        r, t, w = O.blokus_step(st, np.array([O.blokus_action_id(s)], np.int32))
        assert r[0] == 0 and t[0] == 0 and w[0] == 0
    assert st.round[0] == 1 and st.score[0].tolist() == [3, 2, 1, 4] and st.to_move[0] == 0
    b = st.board[0]
    assert b[0, 0] == 1 and b[0, 1] == 1 and b[1, 1] == 1 and b[1, 0] == 0
    cnt, ids = O.blokus_valid(st, cap=512)
    assert cnt[0] == 260
    assert O.blokus_action_string(ids[0, 0]) == "monomino1;(0, 2);north0"
    assert O.blokus_action_string(ids[0, 259]) == "pentominoe12;(2, 2);northwest4"


def test_reference_games(golden):
    assert replay_games(golden, OracleBlokus(len(GAMES))) >= 62


def test_lattice_boards_reference_lists(golden):
    """Hand-made boards run through the REFERENCE (oracle/gen_golden_blokus.py lattice): up to ~170 anchors per player, walls,
    foreign cells, partial inventories -- the oracle's ordered lists and membership == the reference's valid_actions /
    is_valid_action.  (The 8 reference-played games never have more than ~60 anchors.)"""
    g = golden("blokus_lattice")
    K = len(g["count"])
    st = O.BlokusState(K)
    st.set_board(g["board"])
    st.inv[:] = g["inv"]
    st.round[:] = g["round"]
    count, ids = O.blokus_valid(st, player=g["player"], cap=g["ids"].shape[1])
    assert np.array_equal(count, g["count"]) and np.array_equal(ids, g["ids"])
    assert g["count"].max() > 2048 and (g["count"] > 0).all()
    for k in range(K):
        legal = set(ids[k, :count[k]].tolist())
        n = int((g["probes"][k] >= 0).sum())
        assert n >= 4 and g["probes_ok"][k, :n].sum() >= 3                     # three legal ids, up to three neighbours
        assert [int(p in legal) for p in g["probes"][k, :n]] == g["probes_ok"][k, :n].tolist(), k


def test_last_piece_bonus(golden):
    check_bonus(golden, OracleBlokus)


def test_not_listed_actions_reference_golden(golden):
    # overlaps overwrite, negative cells wrap, cells >= 20 / shift ids without a cell / pieces not held raise and leave the state
    check_illegal(golden, OracleBlokus, O.blokus_step_action_id)


def test_check_valid_corner_every_cell_reference_golden(golden):
    # board.py:127-154 does not look at the cell itself: occupied cells answer True too (31 of the fixture's 134)
    g = golden("blokus_illegal")
    grid = O.blokus_valid_corner_grid(g["base_board"])
    assert np.array_equal(grid, g["corner_grid"]) and (grid & (g["base_board"][:, None] != 0)).sum() == 31


def test_rollout_equals_stepwise():
    B, T, seed, first = 6, 90, 77, 3
    a = O.BlokusState(B)
    O.blokus_rollout(a, seed, first, T, n_threads=6)
    b = O.BlokusState(B)
    n_ep = 0
    for t in range(T):
        cnt, ids = O.blokus_valid(b, cap=4096)
        act = np.full(B, -1, np.int32)
        for e in range(B):
            if cnt[e]:
                w = O.philox4x32([first + e, t >> 2, 0, O.TAG_BLOKUS], [seed, 0])
                act[e] = ids[e, (int(w[t & 3]) * int(cnt[e])) >> 32]
        r, term, win = O.blokus_step(b, act)
        for e in np.nonzero(term)[0]:
            n_ep += 1
            one = O.BlokusState(1)
            for k in ("occ", "inv", "score", "round", "to_move"):
                getattr(b, k)[e] = getattr(one, k)[0]
    assert np.array_equal(a.occ, b.occ) and np.array_equal(a.inv, b.inv) and np.array_equal(a.score, b.score)
    assert np.array_equal(a.round, b.round) and np.array_equal(a.to_move, b.to_move)
    assert a.n_episodes.sum() == n_ep and n_ep >= 5 and (a.tcount == T).all()
    assert a.len_sum.sum() + a.tstep.sum() == B * T


def test_observe_golden(golden):
    g = golden("blokus_observe")
    n = len(g["player"])
    st = O.BlokusState(n)
    st.set_board(g["board"])
    st.inv[:] = g["inv"]
    st.score[:] = g["score"]
    ob, op, osc = O.blokus_observe(st, g["player"].astype(np.int8))
    assert np.array_equal(ob, g["obs_board"]) and np.array_equal(op, g["obs_pieces"]) and np.array_equal(osc, g["obs_score"])
