"""Blokus golden vectors from the REFERENCE (run in the build container only) -- TEST INFRASTRUCTURE.

Full random games through ``BlokusEnvironment.valid_actions`` / ``next_state`` of the reference
(numba absent here, so its helpers run as plain Python: ~30 s per game; games run in parallel
processes).  Stored per step: the chosen action (encoded), the number of legal actions, a 64-bit
hash of the ordered legal-action id list, the full ordered list for every 4th step, and the state
after the step (board, inventories, scores, round, next player, reward, terminal, winners).
Plus scripted inventory-exhaustion states for the +15 / +20 bonuses (ai.py:49-52).
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

PIECES = ["monomino1", "domino1", "trominoe1", "trominoe2", "tetrominoes1", "tetrominoes2", "tetrominoes3",
          "tetrominoes4", "tetrominoes5", "pentominoe1", "pentominoe2", "pentominoe3", "pentominoe4",
          "pentominoe5", "pentominoe6", "pentominoe7", "pentominoe8", "pentominoe9", "pentominoe10",
          "pentominoe11", "pentominoe12"]
ORIENT = ["north", "northeast", "east", "southeast", "south", "southwest", "west", "northwest"]
MAX_STEPS = 100
LIST_EVERY = 4
LIST_CAP = 2048


def encode(s):
    if s == "":
        return -1
    name, idx, orient = s.split(";")
    x, y = [int(v) for v in idx.replace("(", "").replace(")", "").split(",")]
    return ((PIECES.index(name) * 400 + y * 20 + x) * 8 + ORIENT.index(orient[:-1])) * 5 + int(orient[-1])


def list_hash(ids):
    h = np.uint64(1469598103934665603)
    for v in ids:
        h = (h ^ np.uint64(int(v) & 0xFFFFFFFF)) * np.uint64(1099511628211)
    return h


def inv_mask(ai):
    return sum(1 << PIECES.index(p) for p in ai.current_pieces)


def play_game(seed):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ref_loader
    R = ref_loader.load()
    env = R["blokus"]()
    rng = np.random.default_rng(seed)
    state, players = env.new_state()
    rec = dict(action=[], n_valid=[], valid_hash=[], board=[], inv=[], score=[], round=[], next_player=[],
               reward=[], terminal=[], winners=[], lists=[], list_step=[])
    with np.errstate(over="ignore"):
        for t in range(MAX_STEPS):
            pl = players[0]
            va = env.valid_actions(state, pl)
            ids = [encode(s) for s in va if s != ""]
            assert ids == sorted(ids), "reference order is ascending in the dense id"
            a = ids[int(rng.integers(0, len(ids)))] if ids else -1
            astr = va[ids.index(a)] if ids else ""
            if ids:
                assert env.is_valid_action(state, pl, astr)
            state, players, rewards, terminal, winners = env.next_state(state, [pl], [astr])
            rec["action"].append(a)
            rec["n_valid"].append(len(ids))
            rec["valid_hash"].append(list_hash(ids))
            if t % LIST_EVERY == 0 or len(ids) < 6:
                row = np.full(LIST_CAP, -1, np.int32)
                row[:len(ids)] = ids
                rec["lists"].append(row)
                rec["list_step"].append(t)
            rec["board"].append(state[0].board_contents.astype(np.int8).copy())
            rec["inv"].append([inv_mask(p) for p in state[2]])
            rec["score"].append([p.player_score for p in state[2]])
            rec["round"].append(state[1])
            rec["next_player"].append(players[0])
            rec["reward"].append(rewards[0])
            rec["terminal"].append(bool(terminal))
            rec["winners"].append(0 if winners is None else sum(1 << w for w in winners))
            if terminal:
                break
    return seed, rec


def bonus_cases(R):
    """Last-piece bonuses: +20 when the final piece is the monomino, +15 otherwise (ai.py:49-52),
    and a pass ('') on a finished inventory."""
    env = R["blokus"]()
    out = []
    for last, action in (("monomino1", "monomino1;(5, 5);east0"), ("domino1", "domino1;(5, 5);east0")):
        state, players = env.new_state()
        board, rnd, ais = state
        board.board_contents[4, 4] = 1            # own cell diagonal to (5,5)
        ais[0].current_pieces = [last]
        ais[0].player_score = 80
        st = (board, 3, ais)
        before = board.board_contents.astype(np.int8).copy()
        ns, npl, rew, term, win = env.next_state(st, [0], [action])
        out.append(dict(before=before, inv_before=[inv_mask(p) for p in ais], score_before=[80, 0, 0, 0], round=3,
                        action=encode(action), board=ns[0].board_contents.astype(np.int8).copy(),
                        inv=[inv_mask(p) for p in ns[2]], score=[p.player_score for p in ns[2]],
                        reward=rew[0], terminal=bool(term), winners=0 if win is None else sum(1 << w for w in win),
                        next_round=ns[1], next_player=npl[0]))
    return out


def lattice_boards():
    """Hand-made boards no game reaches but the API accepts: lattices of single cells give one player up to ~170 anchors
    (the list kernel's window table takes 128 at a time), walls and foreign cells in the windows, partial inventories, the
    board's edges and corners.  -> (board int8 [K,20,20], inv uint32 [K,4], round int32 [K], player int8 [K])"""
    K = 10
    board = np.zeros((K, 20, 20), np.int8)
    board[0, 1::3, 1::3] = 1
    board[1, 0::3, 0::3] = 2
    board[2, 1::3, 1::3] = 3
    board[2, 2::6, 2::6] = 1
    board[3, 2::3, 1::3] = 4
    board[4, 1::3, 1::3] = 1
    board[4, 10:, :] = 0
    board[5, 1::3, 1::3] = 2
    board[5, 0, :] = 4
    board[6, 1::3, 2::3] = 1
    board[6, ::7, ::5] = 3
    rng = np.random.default_rng(5)
    board[7] = (rng.random((20, 20)) < 0.12) * rng.integers(1, 5, (20, 20))
    board[8, 0::4, 0::4] = 4                                   # anchors on the edges and in the corners
    board[8, 19, 19] = 4
    board[9, 2::4, 1::2] = 3                                   # dense columns: windows full of own and forbidden cells
    player = np.array([0, 1, 2, 3, 0, 1, 0, 2, 3, 2], np.int8)
    inv = np.full((K, 4), (1 << 21) - 1, np.uint32)
    inv[3] = 0b101010101010101010101
    inv[6] = 1 << 20
    inv[9] = 0b000000000111111111000
    rnd = np.full(K, 3, np.int32)
    return board, inv, rnd, player


def _lattice_case(k):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ref_loader
    R = ref_loader.load()
    env = R["blokus"]()
    board, inv, rnd, player = lattice_boards()
    state, _ = env.new_state()
    b, _, ais = state
    b.board_contents[:] = board[k]
    for q in range(4):
        ais[q].current_pieces = [PIECES[i] for i in range(21) if (int(inv[k, q]) >> i) & 1]
    st = (b, int(rnd[k]), ais)
    va = env.valid_actions(st, int(player[k]))
    ids = [encode(s) for s in va if s != ""]
    assert ids == sorted(ids)
    # is_valid_action on a few ids only -- the reference answers it by enumerating valid_actions again (:667-719), ~1 min per
    # call on these boards without numba: three legal ones and three neighbours of legal ones (other shift / other anchor)
    n = len(ids)
    probes = sorted(set([ids[0], ids[n // 2], ids[-1], ids[n // 3] + 1, ids[n // 5] + 40, ids[-1] + 5]))
    probes = [i for i in probes if 0 <= i < 336000]
    ok = [bool(env.is_valid_action(st, int(player[k]), decode_str(i))) for i in probes]
    return k, ids, probes, ok


def decode_str(aid):
    shift, o, cell, piece = aid % 5, (aid // 5) % 8, (aid // 40) % 400, aid // 16000
    return "%s;(%d, %d);%s%d" % (PIECES[piece], cell % 20, cell // 20, ORIENT[o], shift)


def gen_lattice(out_dir):
    """`valid_actions` / `is_valid_action` of the REFERENCE on the hand-made boards of `lattice_boards`."""
    board, inv, rnd, player = lattice_boards()
    K = len(board)
    with Pool(min(8, K)) as pool:
        res = sorted(pool.map(_lattice_case, list(range(K))))
    cap = max(len(r[1]) for r in res)
    pcap = max(len(r[2]) for r in res)
    ids = np.full((K, cap), -1, np.int32)
    probes = np.full((K, pcap), -1, np.int32)
    ok = np.zeros((K, pcap), np.uint8)
    count = np.zeros(K, np.int32)
    for k, l, pr, o in res:
        ids[k, :len(l)] = l
        count[k] = len(l)
        probes[k, :len(pr)] = pr
        ok[k, :len(pr)] = o
        print("blokus lattice board", k, "player", int(player[k]), "legal", len(l), "probes", len(pr), "of them legal", int(sum(o)))
    np.savez_compressed(os.path.join(out_dir, "blokus_lattice.npz"), board=board, inv=inv, round=rnd, player=player,
                        count=count, ids=ids, probes=probes, probes_ok=ok)


def _flatten(moves):
    """{piece: {(x, y): [orientation+shift]}} -> dense ids in the dict's own iteration order."""
    out = []
    for piece, by_index in moves.items():
        for (x, y), names in by_index.items():
            for name in names:
                out.append(((PIECES.index(piece) * 400 + int(y) * 20 + int(x)) * 8 + ORIENT.index(name[:-1])) * 5 + int(name[-1]))
    return out


def gen_records(R, out_dir, steps=(1, 5, 9, 17, 30, 45)):
    """The methods of the state RECORDS (Board / AI) an agent may call: get_all_valid_moves (inventory order and a
    caller-chosen piece order), gather_empty_corner_indexes, check_moves, update_board, update_player, calculate_winner --
    on states of golden game 1 replayed through the reference."""
    g = np.load(os.path.join(out_dir, "blokus_game_1.npz"))
    env = R["blokus"]()
    state, players = env.new_state()
    rec = dict(board=[], round=[], color=[], inv=[], moves=[], n_moves=[], sub_pieces=[], sub_moves=[], n_sub=[],
               corners=[], n_corners=[], has_move=[], shift_spec=[], shift_mask=[])
    rng = np.random.default_rng(77)
    cap = 4096
    for t in range(max(steps) + 1):
        aid = int(g["action"][t])
        state, players, _, term, _ = env.next_state(state, players, [decode_str(aid) if aid >= 0 else ""])
        if t not in steps:
            continue
        board, rnd, ais = state
        for c in (1, 2, 3, 4):
            ai = ais[c - 1]
            ids = _flatten(board.get_all_valid_moves(rnd, c, ai.current_pieces))
            sub = ai.current_pieces[::-1][:6]                       # a caller-chosen order: the dict follows it
            ids2 = _flatten(board.get_all_valid_moves(rnd, c, sub))
            corners = board.gather_empty_corner_indexes(c)
            row = np.full(cap, -1, np.int32); row[:len(ids)] = ids
            row2 = np.full(cap, -1, np.int32); row2[:len(ids2)] = ids2
            crow = np.full((128, 2), -1, np.int32); crow[:len(corners)] = np.array(corners, np.int32).reshape(-1, 2)
            rec["board"].append(board.board_contents.astype(np.int8).copy()); rec["round"].append(rnd); rec["color"].append(c)
            rec["inv"].append(inv_mask(ai)); rec["moves"].append(row); rec["n_moves"].append(len(ids))
            rec["sub_pieces"].append([PIECES.index(p) for p in sub] + [-1] * (6 - len(sub)))
            rec["sub_moves"].append(row2); rec["n_sub"].append(len(ids2))
            rec["corners"].append(crow); rec["n_corners"].append(len(corners)); rec["has_move"].append(bool(ai.check_moves(board, rnd)))
            # check_orientation_shifts on anchors, on random cells (mostly no anchors, some occupied) and next to the edges
            specs, masks = [], []
            cells = [tuple(c) for c in corners[:3]] + [(int(rng.integers(0, 20)), int(rng.integers(0, 20))) for _ in range(7)] + [(0, 19), (19, 0)]
            for cell in cells:
                piece, o = int(rng.integers(0, 21)), int(rng.integers(0, 8))
                got = board.check_orientation_shifts(c, PIECES[piece], cell, ORIENT[o])
                specs.append([piece, cell[0], cell[1], o])
                masks.append(sum(1 << int(k) for k in got))
            while len(specs) < 12:
                specs.append([-1, 0, 0, 0]); masks.append(0)
            rec["shift_spec"].append(specs); rec["shift_mask"].append(masks)
    # update_board / place_piece on an empty board, incl. pieces hanging over the left / top edge (numpy wraps negative indices)
    Board, AI = state[0].__class__, state[2][0].__class__
    placed, spec = [], []
    for piece, index, orient, color in (("pentominoe5", (5, 5), "east0", 1), ("pentominoe2", (10, 3), "northwest3", 2),
                                        ("tetrominoes3", (0, 0), "south2", 3), ("trominoe1", (1, 0), "north1", 4),
                                        ("pentominoe12", (10, 14), "southwest4", 2), ("monomino1", (19, 19), "west0", 1)):
        b = Board()
        b.update_board(color, piece, index, orient, 3, True)
        placed.append(b.board_contents.astype(np.int8).copy())
        spec.append([PIECES.index(piece), index[0], index[1], ORIENT.index(orient[:-1]), int(orient[-1]), color])
    # update_player: the whole inventory played in two orders (monomino last: +20, otherwise +15)
    scores = []
    for order in (PIECES[::-1], PIECES):
        a = AI(None, 1)
        trace = []
        for p in order:
            a.update_player(p)
            trace.append(a.player_score)
        scores.append(trace)
    # calculate_winner, incl. ties and all-zero
    winners = []
    for sc in ([10, 20, 5, 7], [9, 9, 3, 9], [0, 0, 0, 0], [4, 4, 4, 4], [89, 104, 104, 60]):
        ps = [AI(None, c) for c in (1, 2, 3, 4)]
        for p, v in zip(ps, sc):
            p.player_score = v
        winners.append(sc + ["RBGYN".index(Board().calculate_winner(ps, 20)[0])])
    np.savez_compressed(os.path.join(out_dir, "blokus_records.npz"), **{k: np.array(v) for k, v in rec.items()},
                        placed=np.array(placed), placed_spec=np.array(spec, np.int32), update_scores=np.array(scores, np.int32),
                        winner_cases=np.array(winners, np.int32))
    print("blokus record fixtures", len(rec["color"]), "states; max moves", max(rec["n_moves"]), "max corners", max(rec["n_corners"]))


EXC_KINDS = {None: 0, "IndexError": 1, "ValueError": 2, "KeyError": 3}


def _illegal_base_states(R, out_dir):
    """Three states the not-listed actions are played on: after the four scripted openers of SURVEY Appendix B, and golden
    game 1 after 21 and 42 plies (replayed through the reference)."""
    env = R["blokus"]()
    state, players = env.new_state()
    for a in ("trominoe1;(0, 0);east0", "domino1;(19, 0);south0", "monomino1;(0, 19);east0", "tetrominoes3;(19, 19);west0"):
        state, players, *_ = env.next_state(state, players, [a])
    out = [(state, players[0])]
    g = np.load(os.path.join(out_dir, "blokus_game_1.npz"))
    state, players = env.new_state()
    for t in range(42):
        aid = int(g["action"][t])
        state, players, *_ = env.next_state(state, players, [decode_str(aid) if aid >= 0 else ""])
        if t in (20, 41):
            out.append((state, players[0]))
    return env, out


def _illegal_actions(k, state, mover, rng):
    """(player, action string) pairs next_state accepts or raises on although valid_actions would not list them."""
    board = state[0].board_contents
    held = state[2][mover].current_pieces
    gone = [p for p in PIECES if p not in held]
    own = [(int(x), int(y)) for y, x in zip(*np.where(board == mover + 1))]
    foreign = [(int(x), int(y)) for y, x in zip(*np.where((board > 0) & (board != mover + 1)))]
    acts = [""]                                                     # a pass by a player who has moves
    big = [p for p in held if p.startswith("pent")][:3] + [p for p in held if p.startswith("tetr")][:2] + held[:2]
    for (x, y) in own[:3] + foreign[:4]:                            # overlaps on own and foreign cells
        for p in big[:3]:
            acts.append("%s;(%d, %d);%s%d" % (p, x, y, ORIENT[int(rng.integers(0, 8))], int(rng.integers(0, 2))))
    acts.append("monomino1;(0, 0);east0" if "monomino1" in held else "%s;(0, 0);east0" % held[0])
    for p in big:                                                   # cells at x or y = -1 .. -4 (numpy wraps), >= 20 (IndexError)
        for (x, y) in ((0, 7), (7, 0), (0, 0), (1, 1), (19, 7), (7, 19), (19, 19), (18, 18), (0, 19), (19, 0)):
            acts.append("%s;(%d, %d);%s%d" % (p, x, y, ORIENT[int(rng.integers(0, 8))], int(rng.integers(0, 5 if p.startswith("pent") else 2))))
    for idx in ((-1, 5), (5, -3), (-20, -20), (-21, 0), (0, -21), (20, 0), (0, 20), (25, 5), (-4, -4), (-17, 3), (3, -18), (-19, -1)):
        p = big[int(rng.integers(0, len(big)))]                     # index cells off the board
        acts.append("%s;(%d, %d);%s%d" % (p, idx[0], idx[1], ORIENT[int(rng.integers(0, 8))], 0))
    for p in gone[:3]:                                              # piece not held: ValueError (ai.py:47), after the board update
        acts.append("%s;(9, 9);east0" % p)
        acts.append("%s;(19, 19);east0" % p)                        # ... unless a cell leaves numpy's range first
    for p in ("monomino1", "domino1", "trominoe2", "tetrominoes1"):  # shift ids that name no cell of the piece
        for s in (1, 2, 3, 4, 7, 9):
            if s >= {"m": 1, "d": 2, "t": 3}.get(p[0], 4) + (1 if p.startswith("tetr") else 0):
                acts.append("%s;(9, 9);south%d" % (p, s))
    acts += ["%s;(8, 8);foo1" % big[0], "%s;(8, 8);1" % big[0], "%s;(8, 8);Northeast0" % big[0],       # unknown names are 'east'
             "zzz;(0, 0);east0", "zzz;(0, 0);", "zzz;(99, 0);eastx", "%s;(3, 3);" % big[0], "%s;(3,3);eastx" % big[0],
             "%s;(3, 3)" % big[0], "%s;(a, 3);east0" % big[0], "%s;3, 4;east0" % big[0], "%s;((4), 5);north1" % big[0]]
    for _ in range(24):                                             # arbitrary
        p = PIECES[int(rng.integers(0, 21))]
        acts.append("%s;(%d, %d);%s%d" % (p, int(rng.integers(-6, 26)), int(rng.integers(-6, 26)), ORIENT[int(rng.integers(0, 8))],
                                          int(rng.integers(0, 5))))
    out = [(mover, a) for a in acts]
    other = (mover + 2) % 4                                          # next_state takes any player as the mover
    out += [(other, "%s;(6, 6);west0" % state[2][other].current_pieces[-1]), (other, "")]
    return [(k, pl, a) for pl, a in out]


_BASES = []


def _illegal_case(args):
    k, pl, action = args
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ref_loader
    R = ref_loader.load()
    if not _BASES:                                                  # once per worker process
        _BASES.extend(_illegal_base_states(R, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")))
    env, bases = _BASES
    state = bases[k][0]
    before = state[0].board_contents.copy()
    res = dict(exc=0, board=before.astype(np.int8), inv=[inv_mask(p) for p in state[2]], score=[p.player_score for p in state[2]],
               round=state[1], next_player=pl, reward=0, terminal=0, winners=0)
    try:
        ns, npl, rew, term, win = env.next_state(state, [pl], [action])
        res.update(board=ns[0].board_contents.astype(np.int8).copy(), inv=[inv_mask(p) for p in ns[2]],
                   score=[p.player_score for p in ns[2]], round=ns[1], next_player=npl[0], reward=rew[0], terminal=int(bool(term)),
                   winners=0 if win is None else sum(1 << w for w in win))
    except (IndexError, ValueError, KeyError) as e:
        res["exc"] = EXC_KINDS[type(e).__name__]
    assert np.array_equal(state[0].board_contents, before), "next_state left its input alone"
    return k, pl, action, res


def gen_illegal(R, out_dir):
    """next_state / Board.update_board on actions valid_actions would not list (BlokusEnvironment.py:412-422 places without a
    check; board.py:87-103 indexes board_contents[y][x] with numpy's rules; ai.py:47 raises for a piece not held), and
    Board.check_valid_corner on every cell (board.py:127-154).  The REFERENCE's answers, exception kind included."""
    env, bases = _illegal_base_states(R, out_dir)
    rng = np.random.default_rng(2026)
    todo = []
    for k, (state, mover) in enumerate(bases):
        todo += _illegal_actions(k, state, mover, rng)
    with Pool(8) as pool:
        res = pool.map(_illegal_case, todo, chunksize=4)
    rec = dict(base=[], player=[], action=[], exc=[], board=[], inv=[], score=[], round=[], next_player=[], reward=[], terminal=[], winners=[])
    for k, pl, action, r in res:
        rec["base"].append(k); rec["player"].append(pl); rec["action"].append(action.encode())
        for key in ("exc", "board", "inv", "score", "round", "next_player", "reward", "terminal", "winners"):
            rec[key].append(r[key])
    # Board.update_board called directly on the record: cells are placed one by one, so an IndexError leaves the earlier ones
    Board = bases[0][0][0].__class__
    ub_spec, ub_exc, ub_board = [], [], []
    for piece, index, orient, color in (("pentominoe1", (17, 4), "east0", 2), ("pentominoe1", (4, 17), "south0", 3),
                                        ("pentominoe9", (0, 0), "west2", 1), ("tetrominoes2", (19, 19), "northeast1", 4),
                                        ("pentominoe5", (-2, 18), "southeast3", 1), ("pentominoe12", (18, -2), "north4", 2),
                                        ("trominoe1", (1, 1), "bogus2", 3), ("pentominoe7", (22, 3), "west0", 4)):
        b = Board(bases[1][0][0])
        try:
            b.update_board(color, piece, index, orient, 3, True)
            exc = 0
        except (IndexError, ValueError, KeyError) as e:
            exc = EXC_KINDS[type(e).__name__]
        ub_spec.append([PIECES.index(piece), index[0], index[1], ORIENT.index(orient[:-1]) if orient[:-1] in ORIENT else -1, int(orient[-1]), color])
        ub_exc.append(exc); ub_board.append(b.board_contents.astype(np.int8).copy())
    # check_valid_corner on EVERY cell, occupied ones included (the method itself does not test emptiness)
    grids = np.zeros((len(bases), 4, 20, 20), np.uint8)
    for k, (state, _) in enumerate(bases):
        bc = state[0].board_contents
        for c in (1, 2, 3, 4):
            for y in range(20):
                for x in range(20):
                    grids[k, c - 1, y, x] = bool(state[0].check_valid_corner(bc, c, y, x))
    np.savez_compressed(os.path.join(out_dir, "blokus_illegal.npz"),
                        base_board=np.array([s[0].board_contents for s, _ in bases], np.int8),
                        base_inv=np.array([[inv_mask(p) for p in s[2]] for s, _ in bases], np.uint32),
                        base_score=np.array([[p.player_score for p in s[2]] for s, _ in bases], np.int32),
                        base_round=np.array([s[1] for s, _ in bases], np.int32), base_mover=np.array([m for _, m in bases], np.int32),
                        base=np.array(rec["base"], np.int32), player=np.array(rec["player"], np.int32),
                        action=np.array(rec["action"], dtype="S48"), exc=np.array(rec["exc"], np.int8),
                        board=np.array(rec["board"], np.int8), inv=np.array(rec["inv"], np.uint32), score=np.array(rec["score"], np.int32),
                        round=np.array(rec["round"], np.int32), next_player=np.array(rec["next_player"], np.int32),
                        reward=np.array(rec["reward"], np.int8), terminal=np.array(rec["terminal"], np.uint8),
                        winners=np.array(rec["winners"], np.uint8),
                        ub_spec=np.array(ub_spec, np.int32), ub_exc=np.array(ub_exc, np.int8), ub_board=np.array(ub_board),
                        corner_grid=grids)
    ex = np.array(rec["exc"])
    print("blokus illegal fixtures", len(ex), "cases: ok", int((ex == 0).sum()), "IndexError", int((ex == 1).sum()),
          "ValueError", int((ex == 2).sum()), "KeyError", int((ex == 3).sum()), "| update_board exc", ub_exc,
          "| corner cells true", int(grids.sum()), "of them occupied", int(sum((grids[k, c] & (bases[k][0][0].board_contents != 0)).sum() for k in range(len(bases)) for c in range(4))))


def gen(R, out_dir, n_games=8):
    with Pool(min(8, n_games)) as pool:
        games = pool.map(play_game, list(range(1, n_games + 1)))
    for seed, rec in games:
        T = len(rec["action"])
        np.savez_compressed(os.path.join(out_dir, "blokus_game_%d.npz" % seed),
                            action=np.array(rec["action"], np.int32), n_valid=np.array(rec["n_valid"], np.int32),
                            valid_hash=np.array(rec["valid_hash"], np.uint64), board=np.array(rec["board"], np.int8),
                            inv=np.array(rec["inv"], np.uint32), score=np.array(rec["score"], np.int32),
                            round=np.array(rec["round"], np.int32), next_player=np.array(rec["next_player"], np.int32),
                            reward=np.array(rec["reward"], np.int8), terminal=np.array(rec["terminal"], np.uint8),
                            winners=np.array(rec["winners"], np.uint8), lists=np.array(rec["lists"], np.int32),
                            list_step=np.array(rec["list_step"], np.int32))
        print("blokus game", seed, "steps", T, "max legal", max(rec["n_valid"]), "final scores", rec["score"][-1],
              "winners", rec["winners"][-1], "terminal", rec["terminal"][-1])
    cases = bonus_cases(R)
    np.savez_compressed(os.path.join(out_dir, "blokus_bonus.npz"),
                        **{k: np.array([c[k] for c in cases]) for k in cases[0]})
    gen_observe(R, out_dir)
    gen_lattice(out_dir)
    gen_records(R, out_dir)
    gen_illegal(R, out_dir)


def gen_observe(R, out_dir, n_steps=28):
    """state_to_observation + perspective conversions along the first steps of golden game 1 (replayed
    through the reference's own next_state)."""
    import re
    g = np.load(os.path.join(out_dir, "blokus_game_1.npz"))
    env = R["blokus"]()
    state, players = env.new_state()
    nums = re.compile(r"-?\d+")
    rec = dict(board=[], inv=[], score=[], round=[], player=[], obs_board=[], obs_pieces=[], obs_score=[],
               conv_in=[], conv_player=[], conv_out=[], conv_back=[])

    def dec(aid):
        shift, o, cell, piece = aid % 5, (aid // 5) % 8, (aid // 40) % 400, aid // 16000
        return "%s;(%d, %d);%s%d" % (PIECES[piece], cell % 20, cell // 20, ORIENT[o], shift)

    def enc_tolerant(s):
        # the reference emits 'np.int32(17)' inside the tuple under numpy 2
        name, idx, orient = s.split(";")
        x, y = [int(v) for v in nums.findall(idx.replace("int32", ""))]
        return ((PIECES.index(name) * 400 + y * 20 + x) * 8 + ORIENT.index(orient[:-1])) * 5 + int(orient[-1])

    for t in range(n_steps):
        aid = int(g["action"][t])
        state, players, _, term, _ = env.next_state(state, players, [dec(aid) if aid >= 0 else ""])
        for pl in range(4):
            o = env.state_to_observation(state, pl)
            rec["board"].append(state[0].board_contents.astype(np.int8).copy())
            rec["inv"].append([inv_mask(p) for p in state[2]])
            rec["score"].append([p.player_score for p in state[2]])
            rec["round"].append(state[1])
            rec["player"].append(pl)
            rec["obs_board"].append(np.asarray(o["board"]).astype(np.int8))
            rec["obs_pieces"].append(o["pieces"].copy())
            rec["obs_score"].append(np.asarray(o["score"]).astype(np.int32))
            if aid >= 0:
                conv = env.convert_real_action_to_player_perspective_action(dec(aid), pl)
                rec["conv_in"].append(aid)
                rec["conv_player"].append(pl)
                rec["conv_out"].append(enc_tolerant(conv))
                back = env.convert_player_perspective_action_to_real_action(dec(enc_tolerant(conv)), pl)
                rec["conv_back"].append(enc_tolerant(back))
    np.savez_compressed(os.path.join(out_dir, "blokus_observe.npz"), **{k: np.array(v) for k, v in rec.items()})
    print("blokus observe fixtures", len(rec["player"]))


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ref_loader
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    if "lattice" in sys.argv[1:]:
        gen_lattice(out)
    elif "records" in sys.argv[1:]:
        gen_records(ref_loader.load(), out)
    elif "illegal" in sys.argv[1:]:
        gen_illegal(ref_loader.load(), out)
    else:
        gen_observe(ref_loader.load(), out)
