"""Replay golden trajectories (generated from the reference, oracle/gen_golden.py) through a backend."""
import numpy as np

BOARD_WTS = None


def board_checksum(board, NN):
    wts = ((np.arange(NN, dtype=np.uint64) * 2654435761 + 12345) & 0xFFFFFFFF)
    return ((board.astype(np.uint64) * wts[None, :]).sum(axis=1) & 0xFFFFFFFF).astype(np.uint32)


def replay_tron(g, make_backend):
    N, P, E, T = int(g["N"]), int(g["P"]), int(g["E"]), int(g["T"])
    be = make_backend(N, P, E, g["start_heads"], g["start_dirs"])
    auto = bool(int(g["auto_reset"]))
    for t in range(T):
        rew, term, win = be.step(np.ascontiguousarray(g["actions"][t]))
        s = be.state()
        assert np.array_equal(s["heads"], g["heads"][t]), "heads differ at step %d" % t
        assert np.array_equal(s["dirs"], g["dirs"][t]), "dirs differ at step %d" % t
        assert np.array_equal(s["deaths"], g["deaths"][t]), "deaths differ at step %d" % t
        assert np.array_equal(rew, g["rewards"][t]), "rewards differ at step %d" % t
        assert np.array_equal(term, g["terminal"][t]), "terminal differs at step %d" % t
        assert np.array_equal(win, g["winners"][t]), "winners differ at step %d" % t
        assert np.array_equal(board_checksum(s["board"], N * N), g["board_sum"][t]), "board differs at step %d" % t
        if auto:
            be.reset(term)
    assert np.array_equal(be.state()["board"], g["final_board"])
    return int(g["terminal"].sum())


def replay_tron_fused_reset(g, make_backend):
    """Same trajectory, but with the kernel's fused auto-reset instead of a separate masked reset."""
    N, P, E, T = int(g["N"]), int(g["P"]), int(g["E"]), int(g["T"])
    assert int(g["auto_reset"]) == 1
    be = make_backend(N, P, E, g["start_heads"], g["start_dirs"])
    sh, sd = g["start_heads"], g["start_dirs"]
    for t in range(T):
        rew, term, win = be.step(np.ascontiguousarray(g["actions"][t]), auto_reset=True)
        assert np.array_equal(rew, g["rewards"][t]) and np.array_equal(term, g["terminal"][t])
        assert np.array_equal(win, g["winners"][t])
        s = be.state()
        tm = g["terminal"][t].astype(bool)
        assert np.array_equal(s["heads"][:, ~tm], g["heads"][t][:, ~tm])
        assert np.array_equal(s["deaths"][:, ~tm], g["deaths"][t][:, ~tm])
        assert np.array_equal(s["dirs"][:, ~tm], g["dirs"][t][:, ~tm])
        if tm.any():
            assert (s["heads"][:, tm] == sh[:, None]).all() and (s["dirs"][:, tm] == sd[:, None]).all()
            assert (s["deaths"][:, tm] == 0).all()
            fresh = np.zeros(N * N, np.int8)
            fresh[sh] = np.arange(1, P + 1)
            assert (s["board"][tm] == fresh[None, :]).all()
    assert np.array_equal(be.state()["board"], g["final_board"])


def replay_ttt(g, make_backend):
    shape = tuple(int(x) for x in g["shape"])
    P, K, E, T = int(g["P"]), int(g["K"]), int(g["E"]), int(g["T"])
    auto = bool(int(g["auto_reset"]))
    be = make_backend(shape, K, P, E)
    n_term = 0
    for t in range(T):
        assert np.array_equal(be.valid(), g["valid"][t]), "valid_actions differ before step %d" % t
        assert np.array_equal(be.to_move(), g["mover"][t]), "mover differs at step %d" % t
        rew, term, win = be.step(np.ascontiguousarray(g["action"][t]))
        assert np.array_equal(be.board(), g["board"][t]), "board differs at step %d" % t
        assert np.array_equal(be.winner(), g["winner"][t]), "sticky winner differs at step %d" % t
        assert np.array_equal(be.to_move(), g["next_player"][t]), "next player differs at step %d" % t
        assert np.array_equal(rew, g["reward"][t]), "reward differs at step %d" % t
        assert np.array_equal(term, g["terminal"][t]), "terminal differs at step %d" % t
        assert np.array_equal(win, g["winners"][t]), "winners differ at step %d" % t
        n_term += int(term.sum())
        if auto:
            be.reset(term)
    return n_term
