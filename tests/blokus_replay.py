"""Replay the reference's Blokus games (tests/golden/blokus_game_*.npz) through a backend."""
import numpy as np

GAMES = list(range(1, 9))


def list_hash(ids):
    h = np.uint64(1469598103934665603)
    with np.errstate(over="ignore"):
        for v in ids:
            h = (h ^ np.uint64(int(v) & 0xFFFFFFFF)) * np.uint64(1099511628211)
    return h


def replay_games(golden, be):
    """be: backend over len(GAMES) envs with methods valid(cap)->(count, ids), step(action)->(r,t,w), state()."""
    games = [golden("blokus_game_%d" % g) for g in GAMES]
    T = max(len(g["action"]) for g in games)
    E = len(games)
    alive = np.ones(E, bool)
    n_lists = 0
    for t in range(T):
        count, ids = be.valid(2048)
        action = np.full(E, -1, np.int32)
        for e, g in enumerate(games):
            if t >= len(g["action"]):
                alive[e] = False
                continue
            assert count[e] == g["n_valid"][t], ("n_valid", e, t, count[e], g["n_valid"][t])
            row = ids[e, :count[e]]
            assert list_hash(row) == g["valid_hash"][t], ("valid list order", e, t)
            where = np.nonzero(g["list_step"] == t)[0]
            if len(where):
                assert np.array_equal(g["lists"][where[0]][:count[e]], row)
                n_lists += 1
            action[e] = g["action"][t]
        r, term, win = be.step(action)
        s = be.state()
        for e, g in enumerate(games):
            if not alive[e] or t >= len(g["action"]):
                continue
            assert np.array_equal(s["board"][e], g["board"][t]), ("board", e, t)
            assert np.array_equal(s["inv"][e], g["inv"][t]), ("inventory", e, t)
            assert np.array_equal(s["score"][e], g["score"][t]), ("score", e, t)
            assert s["round"][e] == g["round"][t] and s["to_move"][e] == g["next_player"][t], ("turn", e, t)
            assert r[e] == g["reward"][t] and term[e] == g["terminal"][t] and win[e] == g["winners"][t], ("outcome", e, t)
    assert n_lists > 100
    return T


def check_bonus(golden, make_backend):
    g = golden("blokus_bonus")
    n = len(g["action"])
    be = make_backend(n)
    be.set_state(g["before"], g["inv_before"], g["score_before"], g["round"], np.zeros(n, np.int32))
    r, term, win = be.step(g["action"].astype(np.int32))
    s = be.state()
    assert np.array_equal(s["board"], g["board"]) and np.array_equal(s["inv"], g["inv"])
    assert np.array_equal(s["score"], g["score"]) and s["score"][0, 0] == 101 and s["score"][1, 0] == 97   # +20 / +15
    assert np.array_equal(r, g["reward"]) and np.array_equal(term, g["terminal"]) and np.array_equal(win, g["winners"])
    assert np.array_equal(s["round"], g["next_round"]) and np.array_equal(s["to_move"], g["next_player"])


EXC_OF = {0: None, 1: IndexError, 2: ValueError, 3: KeyError}      # blokus_illegal.npz `exc`
STATUS_OF = {0: 0, 1: -1, 2: -2}                                   # ... as the code the step's reward slot carries


def illegal_ids(golden, to_id):
    """Fixture cases of blokus_illegal.npz -> (case indices that reach the stepper, their ids).  Strings the reference
    rejects before it touches the board must make `to_id` raise the same exception class."""
    g = golden("blokus_illegal")
    keep, ids = [], []
    for i, raw in enumerate(g["action"]):
        want = EXC_OF[int(g["exc"][i])]
        try:
            aid = to_id(raw.decode())
        except (IndexError, ValueError, KeyError) as e:
            assert want is not None and type(e) is want, (raw, type(e).__name__, want)
            continue
        assert want is not KeyError, raw
        keep.append(i)
        ids.append(aid)
    return g, np.array(keep), np.array(ids, np.int32)


def check_illegal(golden, make_backend, to_id):
    """next_state on actions valid_actions would not list -- overlaps, cells wrapped by numpy's negative indices, IndexError /
    ValueError cases -- against the REFERENCE's answers (oracle/gen_golden_blokus.py gen_illegal)."""
    g, keep, ids = illegal_ids(golden, to_id)
    n = len(keep)
    assert n > 300 and (ids >= 336000).sum() >= 30
    base = g["base"][keep]
    be = make_backend(n)
    be.set_state(g["base_board"][base], g["base_inv"][base], g["base_score"][base], g["base_round"][base], g["player"][keep])
    r, term, win = be.step(ids)
    s = be.state()
    status = np.array([STATUS_OF[int(e)] for e in g["exc"][keep]])
    ok = status == 0
    assert ok.sum() > 200 and (status == -1).sum() > 50 and (status == -2).sum() > 10, (ok.sum(), (status == -1).sum(), (status == -2).sum())
    assert np.array_equal(np.asarray(r)[~ok], status[~ok]), "IndexError / ValueError codes"
    assert np.array_equal(np.asarray(r)[ok], g["reward"][keep][ok])
    assert np.array_equal(np.asarray(term), g["terminal"][keep]) and np.array_equal(np.asarray(win), g["winners"][keep])
    # a raise leaves the reference's state as it was (next_state works on copies): the fixture stores the input state then,
    # with next_player = the mover
    for key, fx in (("board", "board"), ("inv", "inv"), ("score", "score"), ("round", "round"), ("to_move", "next_player")):
        assert np.array_equal(np.asarray(s[key]), g[fx][keep]), key
    changed = (g["board"][keep] != g["base_board"][base]).reshape(n, -1).any(axis=1)
    overwrote = ((g["base_board"][base] != 0) & (g["board"][keep] != g["base_board"][base])).reshape(n, -1).any(axis=1)
    assert changed.sum() > 150 and overwrote.sum() > 20            # the cases do place, and do overwrite other colours
    return n
