"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

TEST INFRASTRUCTURE.  Run as ``python -m oracle.gen_golden [tron] [ttt] [blokus]`` from the
repository root (needs /root/reference; see oracle/ref_loader.py for how it is imported).
Only integer inputs/outputs are stored: actions, states, rewards, terminal flags,
winners.  No reference source or bytecode is written anywhere.

Every fixture records the exact driver logic below, so the tests can replay the
same action sequences through the oracle restatement (CPU) and the HIP kernels (GPU).
"""
import os
import re
import sys
import time

import numpy as np

from . import ref_loader

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
TRON_ACT = ["forward", "right", "left"]          # move_array order, TronGridEnvironment.py:117
TRON_ACT_INT = np.array([0, 1, -1], dtype=np.int8)  # STRING_TO_ACTION, :62-67


def _winmask(w):
    return 0 if w is None else int(sum(1 << int(i) for i in w))


# --------------------------------------------------------------------------- Tron
def tron_reset_table(R):
    rows = []
    for (N, P) in [(20, 4), (40, 4), (19, 4), (15, 4), (20, 2), (21, 3), (20, 6), (6, 3), (7, 5), (40, 8), (64, 4)]:
        for ro in (1, 3):
            for so in (0, 2, -3):
                env = R["tron"]("%d;%d" % (N, P))
                try:
                    (board, heads, dirs, deaths), players = env.new_state(ring_offset=ro, spawn_offset=so)
                except Exception:
                    continue
                rows.append((N, P, ro, so, heads.copy(), dirs.copy()))
    return rows


def tron_lockstep(R, N, P, E, T, seed, auto_reset):
    """E independent games stepped in lockstep for T steps with uniform random actions for ALL players
    (dead players' actions are ignored by the reference). auto_reset: new_state() after a terminal step."""
    rng = np.random.default_rng(seed)
    env = R["tron"]("%d;%d" % (N, P))
    states = [env.new_state()[0] for _ in range(E)]
    A = np.zeros((T, P, E), np.int8)
    H = np.zeros((T, P, E), np.int16)
    D = np.zeros((T, P, E), np.int8)
    K = np.zeros((T, P, E), np.int8)
    RW = np.zeros((T, P, E), np.int8)
    TM = np.zeros((T, E), np.uint8)
    WM = np.zeros((T, E), np.uint8)
    BS = np.zeros((T, E), np.uint32)   # board checksum BEFORE any auto-reset
    wts = (np.arange(N * N, dtype=np.uint64) * 2654435761 + 12345) & 0xFFFFFFFF
    for t in range(T):
        a_idx = rng.integers(0, 3, size=(P, E))
        for e in range(E):
            names = [TRON_ACT[a_idx[p, e]] for p in range(P)]
            s, pl, rew, term, win = env.next_state(states[e], list(range(P)), names)
            A[t, :, e] = TRON_ACT_INT[a_idx[:, e]]
            H[t, :, e], D[t, :, e], K[t, :, e] = s[1], s[2], s[3]
            RW[t, :, e] = rew
            TM[t, e] = bool(term)
            WM[t, e] = _winmask(win)
            BS[t, e] = int((s[0].ravel().astype(np.uint64) * wts).sum() & 0xFFFFFFFF)
            states[e] = env.new_state()[0] if (term and auto_reset) else s
    final_board = np.stack([s[0].ravel() for s in states]).astype(np.int8)
    s0 = env.new_state()[0]
    return dict(N=N, P=P, E=E, T=T, auto_reset=int(auto_reset), start_heads=s0[1].astype(np.int16),
                start_dirs=s0[2].astype(np.int8), actions=A, heads=H, dirs=D, deaths=K, rewards=RW,
                terminal=TM, winners=WM, board_sum=BS, final_board=final_board)


def tron_edge_cases(R):
    """Hand-built states exercising the order-dependent rules of CyTronGrid.pyx:15-62."""
    N, P = 8, 3
    env = R["tron"]("%d;%d" % (N, P))
    cases = []

    def mk(cells, heads, dirs, deaths):
        b = np.zeros((N, N), np.int64)
        for (y, x, v) in cells:
            b[y, x] = v
        return (b, np.array(heads, np.int64), np.array(dirs, np.int64), np.array(deaths, np.int64))

    def idx(y, x):
        return y * N + x

    # (i) two players enter the same empty cell: P0 from the west moving east, P1 from the east moving west
    cases.append(("same_cell", mk([(3, 2, 1), (3, 4, 2), (6, 6, 3)], [idx(3, 2), idx(3, 4), idx(6, 6)], [1, 3, 0], [0, 0, 0]),
                  ["forward", "forward", "forward"]))
    # (ii) lower id enters higher id's CURRENT head
    cases.append(("into_current_head", mk([(3, 2, 1), (3, 3, 2), (6, 6, 3)], [idx(3, 2), idx(3, 3), idx(6, 6)], [1, 1, 0], [0, 0, 0]),
                  ["forward", "forward", "forward"]))
    # (iii) higher id enters lower id's VACATED cell (trail)
    cases.append(("into_vacated", mk([(3, 3, 1), (3, 2, 2), (6, 6, 3)], [idx(3, 3), idx(3, 2), idx(6, 6)], [1, 1, 0], [0, 0, 0]),
                  ["forward", "forward", "forward"]))
    # (iv) running into the head cell of a player that is already dead overwrites its killer
    cases.append(("dead_head_overwrite", mk([(3, 3, 1), (3, 2, 2), (6, 6, 3)], [idx(3, 3), idx(3, 2), idx(6, 6)], [1, 1, 0], [1, 0, 0]),
                  ["forward", "forward", "forward"]))
    # wall in each direction + own trail + turning
    cases.append(("walls", mk([(0, 0, 1), (7, 7, 2), (0, 7, 3)], [idx(0, 0), idx(7, 7), idx(0, 7)], [0, 2, 1], [0, 0, 0]),
                  ["forward", "forward", "forward"]))
    cases.append(("wall_west_turns", mk([(4, 0, 1), (7, 7, 2), (0, 7, 3)], [idx(4, 0), idx(7, 7), idx(0, 7)], [0, 3, 0], [0, 0, 0]),
                  ["left", "right", "left"]))
    cases.append(("own_trail", mk([(3, 3, 1), (3, 4, 1), (5, 5, 2), (6, 6, 3)], [idx(3, 3), idx(5, 5), idx(6, 6)], [0, 0, 0], [0, 0, 0]),
                  ["right", "forward", "left"]))
    # last two alive collide head-on -> everyone dead, winners == []
    cases.append(("all_dead", mk([(3, 2, 1), (3, 4, 2), (6, 6, 3)], [idx(3, 2), idx(3, 4), idx(6, 6)], [1, 3, 0], [0, 0, 2]),
                  ["forward", "forward", "forward"]))
    # one survivor -> winner gets 10
    cases.append(("one_survivor", mk([(0, 0, 1), (3, 4, 2), (6, 6, 3)], [idx(0, 0), idx(3, 4), idx(6, 6)], [0, 3, 0], [0, 0, 1]),
                  ["forward", "forward", "forward"]))
    # stepping a finished game
    cases.append(("post_terminal", mk([(0, 0, 1), (3, 4, 2), (6, 6, 3)], [idx(0, 0), idx(3, 4), idx(6, 6)], [0, 3, 0], [1, 0, 1]),
                  ["forward", "left", "forward"]))
    out = dict(N=N, P=P, names=np.array([c[0] for c in cases]))
    keys = ("board", "heads", "dirs", "deaths")
    pre = {k: [] for k in keys}
    post = {k: [] for k in keys}
    acts, rews, terms, wins = [], [], [], []
    for name, st, names in cases:
        s, pl, rew, term, win = env.next_state(st, list(range(P)), names)
        for k, a, b in zip(keys, st, s):
            pre[k].append(np.asarray(a).ravel())
            post[k].append(np.asarray(b).ravel())
        acts.append([env.STRING_TO_ACTION[n] for n in names])
        rews.append(rew)
        terms.append(bool(term))
        wins.append(_winmask(win))
    for k in keys:
        out["pre_" + k] = np.array(pre[k]).astype(np.int16 if k == "heads" else np.int8)
        out["post_" + k] = np.array(post[k]).astype(np.int16 if k == "heads" else np.int8)
    out.update(actions=np.array(acts, np.int8), rewards=np.array(rews, np.int8),
               terminal=np.array(terms, np.uint8), winners=np.array(wins, np.uint8))
    return out


def tron_observe_cases(R, N, P, E, seed, player_ids=None):
    """state_to_observation on random mid-game states; `player_ids`: draw the observer from this list instead of 0..P-1 (ids
    outside the range: heads / directions / deaths roll by (arange + player) % P, TronGridEnvironment.py:393, the board goes
    through relative_player_inplace's C remainder, CyTronGrid.pyx:1,65-71 -- cdivision=True)."""
    rng = np.random.default_rng(seed)
    env = R["tron"]("%d;%d" % (N, P))
    boards, heads, dirs, deaths, players = [], [], [], [], []
    ob, oh, od, ok = [], [], [], []
    for e in range(E):
        s, _ = env.new_state()
        for t in range(int(rng.integers(0, 14))):
            s, _, _, term, _ = env.next_state(s, list(range(P)), [TRON_ACT[i] for i in rng.integers(0, 3, size=P)])
        pl = int(rng.integers(0, P)) if player_ids is None else int(player_ids[e % len(player_ids)])
        o = env.state_to_observation(s, pl)
        boards.append(s[0].ravel()); heads.append(s[1]); dirs.append(s[2]); deaths.append(s[3]); players.append(pl)
        ob.append(o["board"].ravel()); oh.append(o["heads"]); od.append(o["directions"]); ok.append(o["deaths"])
    f = lambda x, dt: np.array(x).astype(dt)
    return dict(N=N, P=P, board=f(boards, np.int8), heads=f(heads, np.int16), dirs=f(dirs, np.int8),
                deaths=f(deaths, np.int8), player=f(players, np.int8), obs_board=f(ob, np.int8),
                obs_heads=f(oh, np.int16), obs_dirs=f(od, np.int8), obs_deaths=f(ok, np.int8))


def tron_ranking_cases(R, N, P, E, rng):
    """compute_ranking (TronGridEnvironment.py:483-508) on states reached by random play (many of them terminal)."""
    env = R["tron"]("%d;%d" % (N, P))
    boards, deaths, ranks = [], [], []
    for e in range(E):
        s, _ = env.new_state()
        w = None
        for t in range(int(rng.integers(1, 40))):
            s, pl, r, term, w = env.next_state(s, list(range(P)), [TRON_ACT[i] for i in rng.integers(0, 3, size=P)])
            if term and rng.random() < 0.7:
                break
        rk = env.compute_ranking(s, list(range(P)), [] if w is None else list(w))
        boards.append(s[0].ravel().astype(np.int8))
        deaths.append(s[3].astype(np.int8))
        ranks.append([rk[p] for p in range(P)])
    return dict(N=N, P=P, board=np.array(boards), deaths=np.array(deaths), rank=np.array(ranks, np.int8))


def gen_tron(R):
    rng = np.random.default_rng(21)
    for (N, P, E) in [(20, 4, 160), (9, 6, 96), (12, 2, 48), (7, 8, 64)]:
        np.savez_compressed(os.path.join(OUT, "tron_ranking_n%dp%d.npz" % (N, P)), **tron_ranking_cases(R, N, P, E, rng))
    rows = tron_reset_table(R)
    np.savez_compressed(os.path.join(OUT, "tron_reset.npz"),
                        cfg=np.array([r[:4] for r in rows], np.int32),
                        P=np.array([r[1] for r in rows], np.int32),
                        heads=np.array([np.pad(r[4], (0, 8 - len(r[4]))) for r in rows], np.int16),
                        dirs=np.array([np.pad(r[5], (0, 8 - len(r[5]))) for r in rows], np.int8))
    # tuple spawn offsets draw from np.random after np.random.seed(int(time())) (:224,255): pin time()
    tm = R["tron_mod"]
    real_time = tm.time
    rnd = []
    try:
        for tval in (12345, 777):
            tm.time = lambda tval=tval: tval
            for (N, P, ro, so) in [(20, 4, 1, (-3, 4)), (40, 4, 2, (0, 6)), (21, 3, 1, (-2, 2))]:
                env = R["tron"]("%d;%d" % (N, P))
                (b, h, d, k), _ = env.new_state(ring_offset=ro, spawn_offset=so)
                rnd.append((N, P, ro, so[0], so[1], tval, h.copy(), d.copy()))
    finally:
        tm.time = real_time
    np.savez_compressed(os.path.join(OUT, "tron_reset_random.npz"),
                        cfg=np.array([r[:6] for r in rnd], np.int64),
                        heads=np.array([np.pad(r[6], (0, 8 - len(r[6]))) for r in rnd], np.int16),
                        dirs=np.array([np.pad(r[7], (0, 8 - len(r[7]))) for r in rnd], np.int8))
    specs = [("n20p4", 20, 4, 256, 48, 1, True), ("n40p4", 40, 4, 128, 64, 2, True),
             ("n20p2", 20, 2, 32, 48, 3, True), ("n21p3", 21, 3, 32, 48, 4, True),
             ("n20p6", 20, 6, 32, 48, 5, True), ("n7p5", 7, 5, 64, 32, 6, True),
             ("n20p4_noreset", 20, 4, 64, 40, 7, False), ("n9p8_noreset", 9, 8, 32, 24, 8, False)]
    for name, N, P, E, T, seed, ar in specs:
        t0 = time.time()
        np.savez_compressed(os.path.join(OUT, "tron_traj_%s.npz" % name), **tron_lockstep(R, N, P, E, T, seed, ar))
        print("tron", name, "%.1fs" % (time.time() - t0))
    np.savez_compressed(os.path.join(OUT, "tron_edge.npz"), **tron_edge_cases(R))
    np.savez_compressed(os.path.join(OUT, "tron_observe_n20p4.npz"), **tron_observe_cases(R, 20, 4, 64, 11))
    np.savez_compressed(os.path.join(OUT, "tron_observe_n9p6.npz"), **tron_observe_cases(R, 9, 6, 64, 12))
    gen_tron_observe_wrap(R)


def gen_tron_observe_wrap(R):
    """Observer ids outside 0..P-1 (negative, P, beyond)."""
    np.savez_compressed(os.path.join(OUT, "tron_observe_wrap_n20p4.npz"),
                        **tron_observe_cases(R, 20, 4, 48, 13, player_ids=[-1, -2, -4, -5, -9, 4, 5, 6, 7, 8, 11, 13]))
    np.savez_compressed(os.path.join(OUT, "tron_observe_wrap_n9p6.npz"),
                        **tron_observe_cases(R, 9, 6, 48, 14, player_ids=[-1, -6, -7, -13, 6, 7, 8, 11, 12, 17, 23, 40]))


# --------------------------------------------------------------------------- TicTacToe
_INT = re.compile(r"-?\d+")


def _parse_cells(strs, shape):
    """valid_actions() strings -> sorted flat cell indices (numpy-2 repr tolerant)."""
    out = []
    for s in strs:
        if s == "":
            continue
        nums = [int(x) for x in _INT.findall(s.replace("np.int64", "").replace("int64", ""))]
        out.append(int(np.ravel_multi_index(tuple(nums), shape)))
    return out


def ttt_lockstep(R, which, E, T, seed, auto_reset):
    env = R[which]()
    shape = env.observation_shape["board"]
    P = env.max_players
    n_cells = int(np.prod(shape))
    rng = np.random.default_rng(seed)
    states = [env.new_state()[0] for _ in range(E)]
    movers = [0] * E
    ACT = np.zeros((T, E), np.int8)
    BD = np.zeros((T, E, n_cells), np.int8)
    WN = np.zeros((T, E), np.int8)
    MV = np.zeros((T, E), np.int8)      # mover of this step
    NX = np.zeros((T, E), np.int8)      # next player returned
    RW = np.zeros((T, E), np.int8)
    TM = np.zeros((T, E), np.uint8)
    WS = np.zeros((T, E), np.int8)
    VA = np.zeros((T, E), np.uint32)    # valid_actions() of the state BEFORE the step, as an empties bitmask
    for t in range(T):
        for e in range(E):
            st = states[e]
            va = _parse_cells(env.valid_actions(st, movers[e]), shape)
            VA[t, e] = sum(1 << c for c in va)
            u = rng.random()
            if u < 0.04 or not va:
                cell = -1
            elif u < 0.16:
                cell = int(rng.integers(0, n_cells))        # may be occupied -> no-op that still passes the turn
            else:
                cell = int(va[rng.integers(0, len(va))])
            astr = "" if cell < 0 else str(tuple(int(i) for i in np.unravel_index(cell, shape)))
            s, nxt, rew, term, win = env.next_state(st, [movers[e]], [astr])
            ACT[t, e] = cell
            BD[t, e] = s[0].ravel()
            WN[t, e] = -1 if s[1] is None else s[1]
            MV[t, e] = movers[e]
            NX[t, e] = nxt[0]
            RW[t, e] = rew[0]
            TM[t, e] = bool(term)
            WS[t, e] = -1 if win is None else win[0]
            if term and auto_reset:
                states[e], pl = env.new_state()
                movers[e] = pl[0]
            else:
                states[e], movers[e] = s, nxt[0]
    return dict(shape=np.array(shape, np.int32), P=P, K=3, E=E, T=T, auto_reset=int(auto_reset), action=ACT, board=BD,
                winner=WN, mover=MV, next_player=NX, reward=RW, terminal=TM, winners=WS, valid=VA)


def ttt_observe_cases(R, which, E, seed):
    env = R[which]()
    shape = env.observation_shape["board"]
    P = env.max_players
    rng = np.random.default_rng(seed)
    boards, players, obs = [], [], []
    for e in range(E):
        b = rng.integers(-1, P, size=shape).astype(np.int8)
        pl = int(rng.integers(0, P))
        o = env.state_to_observation((b, None), pl)
        boards.append(b.ravel()); players.append(pl); obs.append(np.asarray(o["board"]).ravel())
    return dict(shape=np.array(shape, np.int32), P=P, board=np.array(boards, np.int8),
                player=np.array(players, np.int8), obs_board=np.array(obs, np.int8))


def ttt_current_rewards_cases(R, which, E, seed):
    """current_rewards (reference 2p:219-238, 3p:220-239, 4p:250-269) on states the reference itself reaches: random
    games played through next_state, sampled before and after somebody has won, plus a drawn (full, no winner) board
    when the game produces one."""
    env = R[which]()
    shape = env.observation_shape["board"]
    P = env.max_players
    n_cells = int(np.prod(shape))
    rng = np.random.default_rng(seed)
    boards, winners, rewards = [], [], []
    for e in range(E):
        state, players = env.new_state()
        for t in range(n_cells + 2):
            boards.append(state[0].ravel().copy())
            winners.append(-1 if state[1] is None else int(state[1]))
            rewards.append([int(r) for r in env.current_rewards(state)])
            empty = np.flatnonzero(state[0].ravel() == -1)
            if len(empty) == 0:
                break
            cell = int(empty[int(rng.integers(0, len(empty)))])
            astr = str(tuple(int(i) for i in np.unravel_index(cell, shape)))
            state, players, _, terminal, _ = env.next_state(state, players, [astr])
            if terminal and rng.random() < 0.5:
                boards.append(state[0].ravel().copy())
                winners.append(-1 if state[1] is None else int(state[1]))
                rewards.append([int(r) for r in env.current_rewards(state)])
                break
    return dict(shape=np.array(shape, np.int32), P=P, board=np.array(boards, np.int8), winner=np.array(winners, np.int8),
                rewards=np.array(rewards, np.int8))


def gen_ttt_rewards(R):
    for which, name in (("ttt2", "2p"), ("ttt3", "3p"), ("ttt4", "4p")):
        np.savez_compressed(os.path.join(OUT, "ttt_rewards_%s.npz" % name), **ttt_current_rewards_cases(R, which, 24, 11))


def gen_ttt(R):
    for which, name in (("ttt2", "2p"), ("ttt3", "3p"), ("ttt4", "4p")):
        for ar in (True, False):
            t0 = time.time()
            E = 256 if which != "ttt4" else 96
            d = ttt_lockstep(R, which, E, 40, 100 + int(ar) + len(name), ar)
            np.savez_compressed(os.path.join(OUT, "ttt_traj_%s_%s.npz" % (name, "reset" if ar else "noreset")), **d)
            print("ttt", name, ar, "%.1fs" % (time.time() - t0))
        np.savez_compressed(os.path.join(OUT, "ttt_observe_%s.npz" % name), **ttt_observe_cases(R, which, 64, 5))


def gen_tron_wild64(R):
    """CyTronGrid.next_state_inplace called DIRECTLY (not through TronGridEnvironment, whose STRING_TO_ACTION only ever passes
    -1 / 0 / +1) with actions and stored directions outside their usual ranges: the function computes
    (directions[i] + action + 4) % 4 with C's remainder (cdivision=True), so sums below -4 give a NEGATIVE direction, none of
    the four move branches fires, the player runs into the cell it stands on and the negative direction is stored.  States are
    valid positions of random play (heads on the board, trails consistent): nothing here reads outside an array in the
    reference.  For crl_tron_next_state_inplace64, which promises the Cython function's own behaviour."""
    cy = R["cytron"]
    rng = np.random.default_rng(2024)
    out = {k: [] for k in ("pre_board", "pre_heads", "pre_dirs", "pre_deaths", "actions", "post_board", "post_heads", "post_dirs", "post_deaths")}
    N, P = 12, 4
    env = R["tron"]("%d;%d" % (N, P))
    for game in range(40):
        (board, heads, dirs, deaths), players = env.new_state()
        for t in range(int(rng.integers(0, 12))):                # a few ordinary steps first
            acts = [TRON_ACT[int(a)] for a in rng.integers(0, 3, size=P)]
            (board, heads, dirs, deaths), players, _, term, _ = env.next_state((board, heads, dirs, deaths), list(range(P)), acts)
            if term:
                break
        for rep in range(6):                                     # then chains of wild calls on the same arrays (directions carry over)
            a = rng.integers(-9, 10, size=P).astype(np.int64)
            d = dirs.copy()
            if rep % 2:
                d = d + 4 * rng.integers(-3, 3, size=P)          # stored directions outside 0..3 as well
            b, h, k = board.copy(), heads.copy(), deaths.copy()
            for key, v in (("pre_board", b.reshape(-1)), ("pre_heads", h), ("pre_dirs", d), ("pre_deaths", k), ("actions", a)):
                out[key].append(np.array(v, np.int64))
            cy.next_state_inplace(b, h, d, k, a)
            for key, v in (("post_board", b.reshape(-1)), ("post_heads", h), ("post_dirs", d), ("post_deaths", k)):
                out[key].append(np.array(v, np.int64))
            board, heads, dirs, deaths = b, h, d, k
    rec = {k: np.array(v, np.int16) for k, v in out.items()}
    rec["N"], rec["P"] = np.int64(N), np.int64(P)
    assert (rec["post_dirs"] < 0).any() and (np.abs(rec["actions"]) > 1).any()
    np.savez_compressed(os.path.join(OUT, "tron_wild64.npz"), **rec)
    print("tron_wild64:", len(out["actions"]), "calls,", int((rec["post_dirs"] < 0).sum()), "negative directions stored")


def main(argv):
    os.makedirs(OUT, exist_ok=True)
    R = ref_loader.load()
    what = argv or ["tron", "ttt", "blokus"]
    if "tron" in what:
        gen_tron(R)
    if "tron_observe_wrap" in what:
        gen_tron_observe_wrap(R)
    if "tron" in what or "tron_wild64" in what:
        gen_tron_wild64(R)
    if "ttt" in what:
        gen_ttt(R)
    if "ttt" in what or "ttt_rewards" in what:
        gen_ttt_rewards(R)
    if "blokus" in what:
        from . import gen_golden_blokus
        gen_golden_blokus.gen(R, OUT)


if __name__ == "__main__":
    main(sys.argv[1:])
