"""CPU restatement (oracle/) vs golden vectors captured from the reference + SURVEY 8c known answers."""
import numpy as np
import pytest

from oracle import oracle as O
from backends import OracleTron
from replay import replay_tron, replay_tron_fused_reset

TRAJ = ["n20p4", "n40p4", "n20p2", "n21p3", "n20p6", "n7p5", "n20p4_noreset", "n9p8_noreset"]

# SURVEY.md T5 known answers (container independent)
KAT_START = {
    (20, 4): ([31, 221, 370, 178], [2, 3, 0, 1]), (40, 4): ([61, 841, 1540, 758], None),
    (19, 4): ([30, 226, 330, 134], None), (15, 4): ([24, 148, 200, 76], None),
    (20, 2): ([61, 338], [3, 1]), (21, 3): ([36, 416, 232], [2, 0, 1]),
    (20, 6): ([29, 81, 301, 372, 338, 118], [2, 3, 3, 0, 1, 1]), (6, 3): ([10, 27, 16], [3, 0, 1]),
}


def test_start_positions_kat():
    for (N, P), (heads, dirs) in KAT_START.items():
        h, d = O.tron_start_positions(N, P, 1, 2)
        assert h.tolist() == heads
        if dirs is not None:
            assert d.tolist() == dirs
    h, d = O.tron_start_positions(20, 4, 3, 0)
    assert h.tolist() == [69, 183, 332, 216] and d.tolist() == [2, 3, 0, 1]


def test_start_positions_golden(golden):
    g = golden("tron_reset")
    assert len(g["cfg"]) > 40
    for (N, P, ro, so), heads, dirs in zip(g["cfg"], g["heads"], g["dirs"]):
        h, d = O.tron_start_positions(int(N), int(P), int(ro), int(so))
        assert np.array_equal(h, heads[:P]) and np.array_equal(d, dirs[:P]), (N, P, ro, so)


@pytest.mark.parametrize("name", TRAJ)
def test_traj(golden, name):
    n_term = replay_tron(golden("tron_traj_" + name), OracleTron)
    assert n_term > 0


def test_traj_fused_reset(golden):
    replay_tron_fused_reset(golden("tron_traj_n20p4"), OracleTron)


def test_edge_cases(golden):
    g = golden("tron_edge")
    N, P = int(g["N"]), int(g["P"])
    E = len(g["names"])
    be = OracleTron(N, P, E, [0, 1, 2], [0, 0, 0])
    be.set_state(g["pre_board"], g["pre_heads"].T, g["pre_dirs"].T, g["pre_deaths"].T)
    rew, term, win = be.step(np.ascontiguousarray(g["actions"].T))
    s = be.state()
    for i, nm in enumerate(g["names"]):
        assert np.array_equal(s["board"][i], g["post_board"][i]), nm
        assert np.array_equal(s["heads"][:, i], g["post_heads"][i]), nm
        assert np.array_equal(s["dirs"][:, i], g["post_dirs"][i]), nm
        assert np.array_equal(s["deaths"][:, i], g["post_deaths"][i]), nm
        assert np.array_equal(rew[:, i], g["rewards"][i]), nm
        assert term[i] == g["terminal"][i] and win[i] == g["winners"][i], nm
    # the order-dependence triplet of SURVEY T2-order, stated independently of the fixture
    names = list(g["names"])
    assert g["post_deaths"][names.index("same_cell")].tolist()[:2] == [2, 1]
    assert g["post_deaths"][names.index("into_current_head")].tolist()[:2] == [2, 1]
    assert g["post_deaths"][names.index("into_vacated")].tolist()[:2] == [0, 1]
    assert g["post_deaths"][names.index("dead_head_overwrite")].tolist()[:2] == [2, 1]


def test_all_forward_trace_kat():
    """SURVEY 8a deterministic trace: N20 P4, everyone 'forward'."""
    sh, sd = O.tron_start_positions(20, 4)
    be = OracleTron(20, 4, 1, sh, sd)
    fwd = np.zeros((4, 1), np.int8)
    be.step(fwd)
    assert be.state()["heads"][:, 0].tolist() == [51, 220, 350, 179]
    r, t, w = be.step(fwd)
    assert be.state()["deaths"][:, 0].tolist() == [0, 2, 0, 4] and r[:, 0].tolist() == [1, -1, 1, -1] and not t[0]
    for _ in range(17):
        r, t, w = be.step(fwd)
    s = be.state()
    assert s["deaths"][:, 0].tolist() == [1, 2, 3, 4] and t[0] == 1 and w[0] == 0
    assert r[:, 0].tolist() == [-1, -1, -1, -1] and s["heads"][:, 0].tolist() == [391, 220, 10, 179]


@pytest.mark.parametrize("name", ["n20p4", "n9p6", "wrap_n20p4", "wrap_n9p6"])
def test_observe(golden, name):
    # wrap_*: observer ids outside 0..P-1 (negative, P, beyond: numpy's modulo for the vectors, C's remainder for the board)
    g = golden("tron_observe_" + name)
    N, P = int(g["N"]), int(g["P"])
    E = len(g["player"])
    be = OracleTron(N, P, E, list(range(P)), [0] * P)
    be.set_state(g["board"], g["heads"].T, g["dirs"].T, g["deaths"].T)
    ob, oh, od, ok = be.observe(g["player"])
    assert np.array_equal(ob, g["obs_board"]) and np.array_equal(oh, g["obs_heads"].T)
    assert np.array_equal(od, g["obs_dirs"].T) and np.array_equal(ok, g["obs_deaths"].T)


def tron_random_actions(seed, g, c, P):
    """The documented RNG contract (include/colosseum_hip.h, crl_tron_rollout), restated in Python."""
    acts = []
    for p in range(P):
        w = O.philox4x32([g, c >> 3, p >> 2, O.TAG_TRON], [seed & 0xffffffff, seed >> 32])
        j = c & 7
        v = (int(w[j >> 1]) * 3 ** ((j & 1) * 4 + (p & 3))) & 0xffffffff
        acts.append([0, 1, -1][(v * 3) >> 32])
    return acts


@pytest.mark.parametrize("P,T0", [(4, 0), (6, 5)])
def test_rollout_equals_stepwise(P, T0):
    """The fused rollout is the same computation as explicit contract actions + step + reset
    (also when it starts at a step count that is not a multiple of the 8-step Philox block)."""
    N, B, T, seed, first = 12, 37, 50, 0x1234567890, 1000
    sh, sd = O.tron_start_positions(N, P)
    a = O.TronState(N, P, B)
    O.tron_reset(a, sh, sd)
    a.tcount[:] = T0
    O.tron_rollout(a, seed, first, T, sh, sd)
    b = OracleTron(N, P, B, sh, sd)
    ts = np.zeros(B, np.uint32)
    n_ep = np.zeros(B, np.uint32)
    ret = np.zeros((P, B), np.int64)
    for t in range(T):
        acts = np.zeros((P, B), np.int8)
        for e in range(B):
            acts[:, e] = tron_random_actions(seed, first + e, T0 + t, P)
        r, term, win = b.step(acts, auto_reset=True)
        ret += r
        ts += 1
        n_ep += term
        ts[term.astype(bool)] = 0
    s = b.state()
    assert np.array_equal(a.board, s["board"]) and np.array_equal(a.heads, s["heads"])
    assert np.array_equal(a.deaths, s["deaths"]) and np.array_equal(a.dirs, s["dirs"])
    assert (a.tcount == T0 + T).all() and np.array_equal(a.tstep, ts) and np.array_equal(a.n_episodes, n_ep)
    assert np.array_equal(a.ret_sum, ret) and n_ep.sum() > 0
    assert a.len_sum.sum() + a.tstep.sum() == B * T


def test_random_agent_is_uniform():
    """Base-3 digit extraction: every (step mod 8, player) slot is uniform over the three moves."""
    counts = np.zeros((8, 4, 3), np.int64)
    for g in range(1500):
        for c in range(8):
            for p, a in enumerate(tron_random_actions(42, g, c, 4)):
                counts[c, p, a] += 1
    assert (np.abs(counts / 1500.0 - 1 / 3) < 0.05).all()


@pytest.mark.parametrize("name", ["n20p4", "n9p6", "n12p2", "n7p8"])
def test_ranking_golden(golden, name):
    """compute_ranking incl. the deaths[-1] / missing-Counter-key quirk for survivors (TronGridEnvironment.py:483-508)."""
    g = golden("tron_ranking_" + name)
    r = O.tron_ranking(int(g["N"]), int(g["P"]), g["board"], np.ascontiguousarray(g["deaths"].T))
    assert np.array_equal(r.T, g["rank"])
