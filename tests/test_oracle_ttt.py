"""CPU restatement of TicTacToe vs golden vectors captured from the reference + known answers."""
import numpy as np
import pytest

from oracle import oracle as O
from backends import OracleTTT
from replay import replay_ttt
from ttt_tree import count_games


def test_line_counts_kat():
    # 8 lines on 3x3, 20 on 3x5, 49 on 3x3x3 (K=3)
    assert len(O.ttt_lines(1, 3, 3, 3)) == 8
    assert len(O.ttt_lines(1, 3, 5, 3)) == 20
    assert len(O.ttt_lines(3, 3, 3, 3)) == 49
    assert len(O.ttt_lines(1, 5, 5, 4)) == 28
    assert all(bin(int(m)).count("1") == 3 for m in O.ttt_lines(3, 3, 3, 3))
    assert len(set(int(m) for m in O.ttt_lines(3, 3, 3, 3))) == 49


def test_lines_match_scipy_patterns():
    """Independent check of the window enumeration against the correlate-with-pattern definition
    the reference uses (2p:297-300): a mask wins iff some pattern correlation hits K."""
    scipy_signal = pytest.importorskip("scipy.signal")
    rng = np.random.default_rng(0)
    pats2 = [np.ones((3, 1), np.int8), np.ones((1, 3), np.int8), np.identity(3, dtype=np.int8), np.rot90(np.identity(3, dtype=np.int8))]
    for shape in [(3, 3), (3, 5)]:
        lines = O.ttt_lines(1, shape[0], shape[1], 3)
        for _ in range(300):
            m = rng.random(shape) < 0.55
            want = any((scipy_signal.correlate2d(m.astype(np.int8), p, "valid") == 3).any() for p in pats2)
            bits = int(sum(1 << int(i) for i in np.nonzero(m.ravel())[0]))
            got = any((bits & int(l)) == int(l) for l in lines)
            assert want == got


@pytest.mark.parametrize("name", ["2p_reset", "2p_noreset", "3p_reset", "3p_noreset", "4p_reset", "4p_noreset"])
def test_traj(golden, name):
    assert replay_ttt(golden("ttt_traj_" + name), OracleTTT) > 0


def test_2p_exhaustive_tree_kat():
    assert count_games(OracleTTT) == (255168, 131184, 77904, 46080)


def test_rollout_equals_stepwise():
    dims, K, P, B, T, seed, first = (3, 5), 3, 3, 41, 60, 99, 7
    a = O.TTTState(dims, K, P, B)
    O.ttt_rollout(a, seed, first, T)
    b = OracleTTT(dims, K, P, B)
    n_ep = 0
    ts = np.zeros(B, np.uint32)
    wins = np.zeros((P, B), np.uint32)
    draws = np.zeros(B, np.uint32)
    for t in range(T):
        valid = b.valid()
        act = np.full(B, -1, np.int8)
        for e in range(B):
            cells = [c for c in range(15) if (int(valid[e]) >> c) & 1]
            if cells:
                # one Philox call per 8 plies, one word per 2: the odd ply reads what the even ply's extraction left over
                w = O.philox4x32([first + e, t >> 3, 0, O.TAG_TTT], [seed, 0])
                word = int(w[(t >> 1) & 3])
                if t & 1:
                    word = (word * (len(cells) + 1)) & 0xffffffff
                act[e] = cells[(word * len(cells)) >> 32]
        r, term, win = b.step(act, auto_reset=True)
        ts += 1
        tm = term.astype(bool)
        for p in range(P):
            wins[p] += (tm & (win == p))
        draws += (tm & (win < 0))
        n_ep += int(term.sum())
        ts[tm] = 0
    assert np.array_equal(a.occ, b.st.occ) and np.array_equal(a.winner, b.st.winner)
    assert np.array_equal(a.to_move, b.st.to_move)
    assert (a.tcount == T).all() and np.array_equal(a.tstep, ts) and a.n_episodes.sum() == n_ep
    assert np.array_equal(a.win_count, wins) and np.array_equal(a.draw_count, draws) and n_ep > 50
