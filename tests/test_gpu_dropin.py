"""The single-state drop-in path (host-mapped staging, colosseumrl_amd/single.py) and the entry points added for it:
reference-layout TicTacToe calls, the any-shape fused Tron step, Blokus valid_list / select / is_valid / pack."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from backends import HipBlokus, OracleBlokus, OracleTron
from blokus_replay import GAMES, replay_games


# ------------------------------------------------------------------ TicTacToe, reference layout
@pytest.mark.parametrize("dims,K,P,B,rel_mod", [((3, 3), 3, 2, 1000 + 7, None), ((3, 5), 3, 3, 513, None),
                                                ((3, 3, 3), 3, 4, 300, 3), ((5, 5), 4, 3, 257, None), ((4, 8), 4, 5, 65, None)])
def test_ttt_reference_layout_calls_match_the_mask_calls(dims, K, P, B, rel_mod):
    """crl_ttt_step_board / crl_ttt_observe_board (int8 boards, -1 empty) in lockstep with crl_ttt_step / _valid / _board on
    the occupancy masks and with the oracle: legal, occupied, out-of-range and '' actions, auto-reset on and off."""
    import torch
    from colosseumrl_amd.batched import TTTBatch, TTTBoards
    a, b = TTTBoards(dims, K, P, B), TTTBatch(dims, K, P, B)
    ost = O.TTTState(dims, K, P, B)
    cells = a.n_cells
    rng = np.random.default_rng(cells * 7 + P)
    ends = 0
    for t in range(3 * cells):
        auto = (t % 4) != 3
        act = rng.integers(-2, cells + 1, size=B).astype(np.int8)          # includes -2, -1 ('') and cells (out of range)
        dev = torch.from_numpy(act).cuda()
        ra, ta, wa = a.step(dev, auto_reset=auto, rel_mod=rel_mod)
        rb, tb_, wb = b.step(dev, auto_reset=auto)
        assert torch.equal(ra, rb) and torch.equal(ta, tb_) and torch.equal(wa, wb), t
        assert torch.equal(a.board, b.board()) and torch.equal(a.winner, b.winner) and torch.equal(a.to_move, b.to_move), t
        assert torch.equal(a.valid, b.valid_mask()), t
        assert torch.equal(a.obs, b.board(b.to_move, rel_mod)), t
        r2, t2, w2 = O.ttt_step(ost, act)
        assert np.array_equal(ra.cpu().numpy(), r2) and np.array_equal(ta.cpu().numpy(), t2) and np.array_equal(wa.cpu().numpy(), w2)
        if auto:
            m = t2.astype(bool)
            ost.occ[:, m] = 0
            ost.winner[m] = -1
            ost.to_move[m] = 0
            ends += int(m.sum())
        assert np.array_equal(a.board.cpu().numpy(), ost.board().reshape(B, -1)), t
        pl = torch.from_numpy(rng.integers(0, P, size=B).astype(np.int8)).cuda()
        obs, valid = a.observe(pl, rel_mod)
        assert torch.equal(obs, b.board(pl, rel_mod)) and torch.equal(valid, b.valid_mask())
        obs_abs, _ = a.observe(None)
        assert torch.equal(obs_abs, a.board)
    assert ends > 0


# ------------------------------------------------------------------ Tron: the fused step on every shape, sampled actions
@pytest.mark.parametrize("N,P,B", [(19, 4, 300 + 1), (13, 8, 130), (7, 3, 70), (45, 2, 40), (16, 8, 65)])
def test_tron_step_observe_any_shape_with_sampled_actions(N, P, B):
    """Shapes the 16-byte fused kernel cannot take run the one-game-per-workgroup kernel: one launch, actions == None
    allowed (round 2 raised there).  T x step_observe(None) leaves the state rollout(T) leaves, and every observation
    equals observe_all of that state."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    seed, first, T = 31, 70, 24
    a, b = TronBatch(N, P, B, first_env_id=first), TronBatch(N, P, B, first_env_id=first)
    out = None
    for t in range(T):
        out = a.step_observe(None, seed=seed, out=out)
        b.rollout(1, seed)
        for k in ("board", "heads", "dirs", "deaths", "tcount"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (k, t)
        obs = b.observe_all()
        for k in ("board", "heads", "directions", "deaths"):
            assert torch.equal(out[k], obs[k]), (k, t)


@pytest.mark.parametrize("config", ["", "19;4", "13;8", "20;4", "7;2", "40;3"])
def test_tron_dropin_vs_oracle(config):
    """The BaseEnvironment class on host-mapped memory, random play, against the oracle one step at a time: the
    reference's default 19x19 board among the shapes; observations of every player served after each step."""
    from colosseumrl_amd import get_environment
    env = get_environment("tron")(config)
    N, P = env.N, env.num_players
    rng = np.random.default_rng(N + P)
    names = {0: "forward", 1: "right", -1: "left"}
    sh, sd = O.tron_start_positions(N, P)
    episodes = 0
    for _ in range(6):
        state, players = env.new_state()
        orc = OracleTron(N, P, 1, sh, sd)
        assert state[0].dtype == np.int64 and np.array_equal(state[0].reshape(-1), orc.st.board[0])
        assert state[1].tolist() == list(sh) and state[2].tolist() == list(sd)
        for t in range(200):
            act = rng.integers(-1, 2, size=P)
            state, players, rewards, terminal, winners = env.next_state(state, list(range(P)), [names[int(x)] for x in act])
            r2, t2, w2 = orc.step(act.astype(np.int8).reshape(P, 1))
            assert np.array_equal(state[0].reshape(-1), orc.st.board[0]) and np.array_equal(state[1], orc.st.heads[:, 0])
            assert np.array_equal(state[2], orc.st.dirs[:, 0]) and np.array_equal(state[3], orc.st.deaths[:, 0])
            assert rewards.tolist() == r2[:, 0].tolist() and bool(terminal) == bool(t2[0])
            for p in range(P):
                ob, oh, od, ok = O.tron_observe(orc.st, np.array([p], np.int8))
                got = env.state_to_observation(state, p)
                assert np.array_equal(got["board"].reshape(-1), ob[0]) and np.array_equal(got["heads"], oh[:, 0])
                assert np.array_equal(got["directions"], od[:, 0]) and np.array_equal(got["deaths"], ok[:, 0])
            if terminal:
                assert sum(1 << int(w) for w in winners) == int(w2[0])
                episodes += 1
                break
    assert episodes >= 3


def test_tron_dropin_fuses_observations_only_while_they_are_read():
    """The drop-in Tron class writes the P observations of the new state in next_state's launch while somebody reads them, and
    stops after OBS_IDLE_STEPS unread calls (stepping-only loops then touch the board where the reference does and nothing else).
    Both forms, and the switch between them in either direction, against the oracle; an observation asked of a state whose
    launch did not fuse them comes from crl_tron_relative_player_inplace64."""
    from colosseumrl_amd import get_environment
    env = get_environment("tron")("20;4")
    N, P = 20, 4
    rng = np.random.default_rng(11)
    names = {0: "forward", 1: "right", -1: "left"}
    sh, sd = O.tron_start_positions(N, P)

    def fresh():
        st, _ = env.new_state()
        return st, OracleTron(N, P, 1, sh, sd)

    def step(state, orc):
        act = rng.integers(-1, 2, size=P)
        state, _, rewards, terminal, _ = env.next_state(state, list(range(P)), [names[int(x)] for x in act])
        r2, t2, _ = orc.step(act.astype(np.int8).reshape(P, 1))
        assert np.array_equal(state[0].reshape(-1), orc.st.board[0]) and np.array_equal(state[1], orc.st.heads[:, 0])
        assert np.array_equal(state[2], orc.st.dirs[:, 0]) and np.array_equal(state[3], orc.st.deaths[:, 0])
        assert rewards.tolist() == r2[:, 0].tolist() and bool(terminal) == bool(t2[0])
        return (state, orc) if not terminal else fresh()

    def check_obs(state, orc, p):
        ob, oh, od, ok = O.tron_observe(orc.st, np.array([p], np.int8))
        got = env.state_to_observation(state, p)
        assert np.array_equal(got["board"].reshape(-1), ob[0]) and np.array_equal(got["heads"], oh[:, 0])
        assert np.array_equal(got["directions"], od[:, 0]) and np.array_equal(got["deaths"], ok[:, 0])

    state, orc = fresh()
    for phase in range(3):
        for _ in range(env.OBS_IDLE_STEPS + 6):                  # nobody reads: after OBS_IDLE_STEPS the launch stops fusing
            state, orc = step(state, orc)
        assert env._observed is None and env._obs_idle >= env.OBS_IDLE_STEPS
        check_obs(state, orc, phase % P)                         # not fused for this state: the Cython function's own entry
        assert env._obs_idle == 0
        for _ in range(5):                                       # read every step: served from the fused launch
            state, orc = step(state, orc)
            assert env._observed is not None
            for p in range(P):
                check_obs(state, orc, p)


def test_dropin_results_are_cached_by_value_not_by_identity():
    """What next_state leaves behind for valid_actions / state_to_observation is keyed by the state's VALUE: an equal
    copy hits it, a state changed in place does not (and is evaluated afresh on the GPU)."""
    from colosseumrl_amd import get_environment
    # Tron
    env, fresh = get_environment("tron")("20;4"), get_environment("tron")("20;4")
    s, _ = env.new_state()
    s, *_ = env.next_state(s, [0, 1, 2, 3], ["left", "forward", "right", "forward"])
    twin = tuple(a.copy() for a in s)
    for p in range(4):
        a, b = env.state_to_observation(twin, p), fresh.state_to_observation(s, p)
        assert all(np.array_equal(a[k], b[k]) for k in a)
    s[0][0, 0] = 3                                                       # caller scribbles on the returned board
    a, b = env.state_to_observation(s, 1), fresh.state_to_observation(s, 1)
    assert a["board"][0, 0] == b["board"][0, 0] == 2 and np.array_equal(a["board"], b["board"])   # player 2's cell, seen by player 1
    # TicTacToe
    env, fresh = get_environment("tictactoe_3p")(), get_environment("tictactoe_3p")()
    s, pl = env.new_state()
    s, pl, *_ = env.next_state(s, pl, ["(1, 2)"])
    assert env.valid_actions(s, pl[0]) == fresh.valid_actions(s, pl[0]) and "(1, 2)" not in env.valid_actions(s, pl[0])
    assert np.array_equal(env.state_to_observation(s, pl[0])["board"], fresh.state_to_observation(s, pl[0])["board"])
    assert np.array_equal(env.state_to_observation(s, 2)["board"], fresh.state_to_observation(s, 2)["board"])
    s[0][0, 0] = 2
    assert env.valid_actions(s, pl[0]) == fresh.valid_actions(s, pl[0]) and "(0, 0)" not in env.valid_actions(s, pl[0])
    assert not env.is_valid_action(s, pl[0], "(0, 0)") and env.is_valid_action(s, pl[0], "(0, 1)")
    # Blokus
    env, fresh = get_environment("blokus")(), get_environment("blokus")()
    s, pl = env.new_state()
    for _ in range(5):
        va = env.valid_actions(s, pl[0])
        assert va == fresh.valid_actions(s, pl[0])
        assert env.is_valid_action(s, pl[0], va[len(va) // 2]) and fresh.is_valid_action(s, pl[0], va[-1])
        o1, o2 = env.state_to_observation(s, pl[0]), fresh.state_to_observation(s, pl[0])
        assert all(np.array_equal(o1[k], o2[k]) for k in o1)
        s, pl, *_ = env.next_state(s, pl, [va[len(va) // 3]])
    s[0].board_contents[10, 10] = 4                                      # in-place change: the cached list no longer applies
    assert env.valid_actions(s, pl[0]) == fresh.valid_actions(s, pl[0])
    other = (pl[0] + 2) % 4
    assert env.valid_actions(s, other) == fresh.valid_actions(s, other)


# ------------------------------------------------------------------ Blokus: the compacted ordered list, select, is_valid, pack
class HipBlokusList(HipBlokus):
    """HipBlokus whose valid() is crl_blokus_valid_list (the compacted ordered ids) instead of the dense bitmap."""

    def valid(self, cap, player=None):
        pl = None if player is None else self.torch.from_numpy(np.ascontiguousarray(player, np.int8)).to(self.bb.device)
        count, ids = self.bb.valid_list(cap, player=pl)
        return count.cpu().numpy(), ids.cpu().numpy()


def test_blokus_valid_list_reference_games_golden(golden):
    """count + ORDERED id list out of crl_blokus_valid_list == the reference's valid_actions over its 8 complete games."""
    assert replay_games(golden, HipBlokusList(len(GAMES))) >= 62


def test_blokus_list_select_is_valid_pack_vs_oracle_random():
    import torch
    B, T = 97, 84
    rng = np.random.default_rng(11)
    hip, orc = HipBlokusList(B), OracleBlokus(B)
    bb = hip.bb
    for t in range(T):
        c1, ids1 = hip.valid(2048)
        c2, ids2 = orc.valid(2048)
        assert np.array_equal(c1, c2) and np.array_equal(ids1, ids2), t
        # "play the r-th legal action" for caller-chosen ranks, incl. first, last and out-of-range ones
        rank = np.where(c2 > 0, rng.integers(0, np.maximum(c2, 1)), 0).astype(np.int32)
        rank[t % B] = c2[t % B] - 1
        rank[(t + 1) % B] = c2[(t + 1) % B]                              # one past the end -> -1
        rank[(t + 2) % B] = -1
        act, cnt = bb.select(torch.from_numpy(rank).cuda())
        act = act.cpu().numpy()
        assert np.array_equal(cnt.cpu().numpy(), c2)
        for e in range(B):
            want = ids2[e, rank[e]] if 0 <= rank[e] < c2[e] else -1
            assert act[e] == want, (t, e, rank[e], c2[e])
        # is_valid: the selected ids, then random ids (mostly illegal), against membership in the oracle's list
        probe = np.where(act >= 0, act, rng.integers(0, 336000, size=B)).astype(np.int32)
        probe[::3] = rng.integers(-3, 336003, size=len(probe[::3]))
        ok = bb.is_valid(torch.from_numpy(probe).cuda()).cpu().numpy()
        for e in range(B):
            assert bool(ok[e]) == bool(probe[e] in set(ids2[e, :c2[e]].tolist())), (t, e, probe[e])
        play = np.full(B, -1, np.int32)
        for e in range(B):
            if c2[e]:
                play[e] = ids2[e, int(rng.integers(0, c2[e]))]
        hip.step(play)
        orc.step(play)
        if t % 9 == 0:                                                   # other players' lists, and pack == inverse of board
            pl = rng.integers(0, 4, size=B).astype(np.int8)
            c1, ids1 = hip.valid(2048, player=pl)
            c2, ids2 = orc.valid(2048, player=pl)
            assert np.array_equal(c1, c2) and np.array_equal(ids1, ids2)
            occ = bb.occ.clone()
            bb.occ.zero_()
            bb.set_board(torch.from_numpy(orc.st.board.astype(np.int8).reshape(B, 20, 20)).cuda())
            assert torch.equal(bb.occ, occ)
    # a list cut by a small cap: the first `cap` ids, count still the full length
    hip2, orc2 = HipBlokusList(5), OracleBlokus(5)
    c, ids = hip2.valid(16)
    c2, ids2 = orc2.valid(2048)
    assert c.tolist() == [116] * 5 and np.array_equal(ids, ids2[:, :16])


def test_blokus_lattice_boards_reference_golden(golden):
    """The reference's own valid_actions / is_valid_action on hand-made boards with up to ~170 anchors per player
    (tests/golden/blokus_lattice.npz) against crl_blokus_valid_list / _valid / _select / _is_valid."""
    import torch
    from colosseumrl_amd.batched import BlokusBatch
    g = golden("blokus_lattice")
    K, cap = len(g["count"]), g["ids"].shape[1]
    bb = BlokusBatch(K)
    bb.set_board(torch.from_numpy(g["board"]).cuda())
    bb.round.copy_(torch.from_numpy(g["round"]).cuda())
    bb.inv.copy_(torch.from_numpy(g["inv"].view(np.int32)).cuda().view(bb.inv.dtype))
    pl = torch.from_numpy(g["player"]).cuda()
    count, ids = bb.valid_list(cap, player=pl)
    assert np.array_equal(count.cpu().numpy(), g["count"]) and np.array_equal(ids.cpu().numpy(), g["ids"])
    assert np.array_equal(bb.valid(player=pl).cpu().numpy(), g["count"])
    for frac in (0.0, 0.37, 1.0):                                         # first, some, last action of every list
        rank = np.minimum((g["count"] * frac).astype(np.int32), g["count"] - 1)
        act, _ = bb.select(torch.from_numpy(rank).cuda(), player=pl)
        assert np.array_equal(act.cpu().numpy(), g["ids"][np.arange(K), rank])
    for j in range(g["probes"].shape[1]):                                 # one probe per board and call
        probe = np.where(g["probes"][:, j] >= 0, g["probes"][:, j], -1).astype(np.int32)
        ok = bb.is_valid(torch.from_numpy(probe).cuda(), player=pl).cpu().numpy()
        want = np.where(g["probes"][:, j] >= 0, g["probes_ok"][:, j], 0)
        assert np.array_equal(ok, want), j


def test_blokus_record_methods_reference_golden(golden):
    """The methods of the state RECORDS a reference agent may call -- Board.get_all_valid_moves (inventory order and a
    caller-chosen piece order), gather_empty_corner_indexes, check_valid_corner, AI.collect_moves / check_moves -- against the
    reference's own records on 24 (state, colour) pairs of a game it played (tests/golden/blokus_records.npz)."""
    from colosseumrl_amd.envs.blokus.actions import PIECE_NAMES, encode, ORIENTATIONS
    from colosseumrl_amd.envs.blokus.ai import AI
    from colosseumrl_amd.envs.blokus.board import Board
    g = golden("blokus_records")

    def flatten(moves):
        return [encode(PIECE_NAMES.index(p), x, y, ORIENTATIONS.index(name[:-1]), int(name[-1]))
                for p, by_index in moves.items() for (x, y), names in by_index.items() for name in names]

    assert len(g["color"]) == 24 and g["n_moves"].max() > 1000
    for k in range(len(g["color"])):
        board = Board()
        board.board_contents[:] = g["board"][k]
        color, rnd = int(g["color"][k]), int(g["round"][k])
        ai = AI(board, color)
        ai.current_pieces = [PIECE_NAMES[i] for i in range(21) if (int(g["inv"][k]) >> i) & 1]
        moves = ai.collect_moves(board, rnd)
        assert flatten(moves) == g["moves"][k, :g["n_moves"][k]].tolist(), k
        assert all(isinstance(x, int) and isinstance(y, int) for by_index in moves.values() for (x, y) in by_index)
        sub = [PIECE_NAMES[i] for i in g["sub_pieces"][k] if i >= 0]
        assert flatten(board.get_all_valid_moves(rnd, color, sub)) == g["sub_moves"][k, :g["n_sub"][k]].tolist(), k
        corners = board.gather_empty_corner_indexes(color)
        assert corners == [tuple(c) for c in g["corners"][k, :g["n_corners"][k]].tolist()], k
        assert ai.check_moves(board, rnd) == bool(g["has_move"][k]) and (ai.all_valid_moves == moves)
        if corners:
            x, y = corners[len(corners) // 2]
            assert board.check_valid_corner(board.board_contents, color, y, x)
            assert not board.check_valid_corner(board.board_contents, color, 0, 0) or (0, 0) in corners
        for (piece, x, y, o), mask in zip(g["shift_spec"][k], g["shift_mask"][k]):        # anchors, random cells, edges
            if piece < 0:
                continue
            got = board.check_orientation_shifts(color, PIECE_NAMES[piece], (int(x), int(y)), ORIENTATIONS[o])
            assert got.dtype == np.int64 and sum(1 << int(j) for j in got) == int(mask), (k, piece, x, y, o)
    assert Board().check_orientation_shifts(1, "monomino1", (20, 3), "north").tolist() == []      # off the board
    assert Board().check_orientation_shifts(1, "pentominoe3", (0, 0), "north").size < 5            # some shifts leave the board


def test_blokus_dropin_not_listed_actions_reference_golden(golden):
    """BlokusEnvironment.next_state on all 497 strings of blokus_illegal.npz -- actions valid_actions would not list: what the
    reference raises is raised (IndexError / ValueError / KeyError, same class), what it places is placed the same way
    (overwrites, numpy wrap, unknown orientation names = east), the state handed in is never modified, and the returned
    state is usable (valid_actions of the next mover equals the oracle's list on that board)."""
    from colosseumrl_amd import get_environment
    from colosseumrl_amd.envs.blokus import actions as A
    from colosseumrl_amd.envs.blokus.ai import AI
    from colosseumrl_amd.envs.blokus.board import Board
    from blokus_replay import EXC_OF
    g = golden("blokus_illegal")
    env = get_environment("blokus")()

    def state_of(k):
        board = Board()
        board.board_contents[:] = g["base_board"][k]
        ais = [AI(board, c) for c in (1, 2, 3, 4)]
        for q in range(4):
            ais[q].current_pieces = [A.PIECE_NAMES[i] for i in range(21) if (int(g["base_inv"][k, q]) >> i) & 1]
            ais[q].player_score = int(g["base_score"][k, q])
        return board, int(g["base_round"][k]), ais

    kinds = {None: 0, IndexError: 0, ValueError: 0, KeyError: 0}
    checked_lists = 0
    for i, raw in enumerate(g["action"]):
        k, pl, want = int(g["base"][i]), int(g["player"][i]), EXC_OF[int(g["exc"][i])]
        state = state_of(k)
        try:
            ns, npl, rew, term, win = env.next_state(state, [pl], [raw.decode()])
            got = None
        except (IndexError, ValueError, KeyError) as e:
            got = type(e)
        assert got is want, (raw, got, want)
        kinds[got] += 1
        assert np.array_equal(state[0].board_contents, g["base_board"][k]) and state[1] == int(g["base_round"][k])
        assert [p.player_score for p in state[2]] == g["base_score"][k].tolist()
        if got is not None:
            continue
        assert np.array_equal(ns[0].board_contents, g["board"][i]) and ns[1] == int(g["round"][i]), raw
        assert [p.player_score for p in ns[2]] == g["score"][i].tolist(), raw
        assert [sum(1 << A.PIECE_INDEX[n] for n in p.current_pieces) for p in ns[2]] == g["inv"][i].tolist(), raw
        assert npl == [int(g["next_player"][i])] and rew == [int(g["reward"][i])] and term == bool(g["terminal"][i]), raw
        assert (win is None and g["winners"][i] == 0 and not term) or \
            (win is not None and sum(1 << w for w in win) == int(g["winners"][i])), raw
        if i % 16 == 0:                                          # the legal list of the next mover on the board that came out
            ob = O.BlokusState(1)
            ob.set_board(g["board"][i][None])
            ob.inv[0] = g["inv"][i]
            ob.round[0] = g["round"][i]
            ob.to_move[0] = g["next_player"][i]
            cnt, ids = O.blokus_valid(ob, cap=8192)
            va = env.valid_actions(ns, npl[0])
            assert (va == [""] and cnt[0] == 0) or [A.string_to_id(s) for s in va] == ids[0, :cnt[0]].tolist(), raw
            checked_lists += 1
    assert kinds[None] == 260 and kinds[IndexError] == 199 and kinds[ValueError] == 29 and kinds[KeyError] == 9 and checked_lists > 10
    # the error path leaves the instance usable: the next ordinary call on the same env
    s, pl = env.new_state()
    with pytest.raises(IndexError):
        env.next_state(s, pl, ["pentominoe6;(18, 0);east0"])
    s2, pl2, *_ = env.next_state(s, pl, ["trominoe1;(0, 0);east0"])
    assert pl2 == [1] and s2[0].board_contents[0, 1] == 1 and len(env.valid_actions(s2, 1)) == 116


def test_blokus_dropin_list_grows_past_its_first_capacity(golden):
    """A hand-made board with 12,952 legal actions through the drop-in class: the mapped id list starts at 4,096 entries and
    grows (one more launch) instead of raising; strings in reference order, is_valid_action on both ends."""
    from colosseumrl_amd import get_environment
    from colosseumrl_amd.envs.blokus import actions as A
    from colosseumrl_amd.envs.blokus.ai import AI
    from colosseumrl_amd.envs.blokus.board import Board
    g = golden("blokus_lattice")
    env = get_environment("blokus")()
    for k in (8, 0, 6):                                          # 12,952 ids, 4,616 ids, 58 ids (back to a short list)
        board = Board()
        board.board_contents[:] = g["board"][k]
        ais = [AI(board, c) for c in (1, 2, 3, 4)]
        for q in range(4):
            ais[q].current_pieces = [A.PIECE_NAMES[i] for i in range(21) if (int(g["inv"][k, q]) >> i) & 1]
        state = (board, int(g["round"][k]), ais)
        pl = int(g["player"][k])
        va = env.valid_actions(state, pl)
        n = int(g["count"][k])
        assert len(va) == n and [A.string_to_id(s) for s in va] == g["ids"][k, :n].tolist(), k
        assert env.is_valid_action(state, pl, va[0]) and env.is_valid_action(state, pl, va[-1])
        d = env.valid_actions_dict(state, pl)
        assert sum(len(v) for by_index in d.values() for v in by_index.values()) == n


def test_blokus_valid_list_more_than_128_anchors():
    """Boards no game reaches but set_board accepts: a lattice of single cells gives a player ~170 anchors, more than one
    chunk (128) of the list kernel's window table -- count and ordered ids vs the oracle, uncut and cut by `cap`; plus
    select / is_valid / the count pass on the same boards."""
    import torch
    from colosseumrl_amd.batched import BlokusBatch
    B = 8
    board = np.zeros((B, 20, 20), np.int8)
    board[0, 1::3, 1::3] = 1                                   # player 0: 49 cells, anchors on all four diagonals
    board[1, 0::3, 0::3] = 2
    board[2, 1::3, 1::3] = 3
    board[2, 2::6, 2::6] = 1                                   # some of those anchors taken by another colour
    board[3, 2::3, 1::3] = 4
    board[4, 1::3, 1::3] = 1
    board[4, 10:, :] = 0                                       # exactly the top half: fewer than 128
    board[5, 1::3, 1::3] = 2
    board[5, 0, :] = 4                                         # a wall of another colour
    board[6, 1::3, 2::3] = 1
    board[6, ::7, ::5] = 3
    rng = np.random.default_rng(5)
    board[7] = (rng.random((20, 20)) < 0.12) * rng.integers(1, 5, (20, 20))
    player = np.array([0, 1, 2, 3, 0, 1, 0, 2], np.int8)
    bb = BlokusBatch(B)
    bb.set_board(torch.from_numpy(board).cuda())
    bb.round.fill_(3)
    inv = np.full((B, 4), (1 << 21) - 1, np.uint32)
    inv[3] = 0b101010101010101010101                           # a partial inventory
    inv[6] = 1 << 20
    bb.inv.copy_(torch.from_numpy(inv.view(np.int32)).cuda().view(bb.inv.dtype))
    st = O.BlokusState(B)
    st.set_board(board)
    st.inv[:] = inv
    st.round[:] = 3
    pl = torch.from_numpy(player).cuda()
    c2, ids2 = O.blokus_valid(st, player=player, cap=16384)
    assert c2.max() > 2048 and c2[0] > 2048                    # the big lists do not fit the usual cap
    for cap in (16384, 2048, 100):
        count, ids = bb.valid_list(cap, player=pl)
        assert np.array_equal(count.cpu().numpy(), c2), cap
        got = ids.cpu().numpy()
        for e in range(B):
            k = min(int(c2[e]), cap)
            assert np.array_equal(got[e, :k], ids2[e, :k]), (cap, e)
            assert (got[e, k:] == -1).all(), (cap, e)            # nothing is written past the list or past `cap`
    assert np.array_equal(bb.valid(player=pl).cpu().numpy(), c2)
    rank = np.where(c2 > 0, c2 - 1, 0).astype(np.int32)       # the LAST action of every list
    act, cnt = bb.select(torch.from_numpy(rank).cuda(), player=pl)
    want = np.array([ids2[e, rank[e]] if c2[e] else -1 for e in range(B)], np.int32)
    assert np.array_equal(act.cpu().numpy(), want) and np.array_equal(cnt.cpu().numpy(), c2)
    assert bb.is_valid(act, player=pl).cpu().numpy().tolist() == [int(c > 0) for c in c2]
    # count only (ids NULL): no store at all
    cnt_only = torch.empty((B,), dtype=torch.int32, device="cuda")
    from colosseumrl_amd._native import check, lib
    from colosseumrl_amd.batched import _ptr, _stream
    check(lib().crl_blokus_valid_list(bb._ctx.handle, B, *bb._state(), _ptr(pl), None, _ptr(cnt_only), 0, _stream()), "list")
    assert np.array_equal(cnt_only.cpu().numpy(), c2)


def test_blokus_valid_list_full_size_counts():
    """BASELINE config 4's batch: list lengths == the count pass at B = 16,384 on mid-game positions, lists ascending."""
    import torch
    from colosseumrl_amd.batched import BlokusBatch
    bb = BlokusBatch(16384)
    bb.rollout(30, 3)
    count, ids = bb.valid_list(2048)
    assert torch.equal(count, bb.valid())
    ids = ids.cpu().numpy()
    c = count.cpu().numpy()
    assert c.max() <= 2048 and c.max() > 200
    for e in range(0, 16384, 257):
        row = ids[e, :c[e]]
        assert (np.diff(row) > 0).all() and (ids[e, c[e]:] == -1).all()
    st = O.BlokusState(64)
    st.occ[:] = bb.occ.cpu().numpy().view(np.uint32)[:64]
    st.inv[:] = bb.inv.cpu().numpy().view(np.uint32)[:64]
    st.score[:] = bb.score.cpu().numpy()[:64]
    st.round[:] = bb.round.cpu().numpy()[:64]
    st.to_move[:] = bb.to_move.cpu().numpy()[:64]
    c2, ids2 = O.blokus_valid(st, cap=2048, n_threads=8)
    assert np.array_equal(c[:64], c2) and np.array_equal(ids[:64], ids2)


def test_blokus_dropin_vs_oracle_full_game():
    """The Blokus BaseEnvironment class on host-mapped memory plays a whole random game next to the oracle: every
    valid_actions list (strings in reference order), is_valid_action, next_state outcome and observation."""
    from colosseumrl_amd import get_environment
    from colosseumrl_amd.envs.blokus import actions as A
    env = get_environment("blokus")()
    orc = OracleBlokus(1)
    rng = np.random.default_rng(3)
    state, players = env.new_state()
    for t in range(120):
        pl = players[0]
        va = env.valid_actions(state, pl)
        c2, ids2 = orc.valid(4096)
        want = [A.id_to_string(i) for i in ids2[0, :c2[0]]] or [""]
        assert va == want, t
        pick = va[int(rng.integers(0, len(va)))]
        if pick:
            assert env.is_valid_action(state, pl, pick)
            assert not env.is_valid_action(state, (pl + 1) % 4, pick) or pick in env.valid_actions(state, (pl + 1) % 4)
        state, players, rewards, terminal, winners = env.next_state(state, players, [pick])
        r2, t2, w2 = orc.step(np.array([A.string_to_id(pick) if pick else -1], np.int32))
        s2 = orc.state()
        assert np.array_equal(state[0].board_contents, s2["board"][0].reshape(20, 20))
        assert [p.player_score for p in state[2]] == s2["score"][0].tolist() and state[1] == s2["round"][0]
        assert players == [int(s2["to_move"][0])] and rewards == [int(r2[0])] and terminal == bool(t2[0])
        if terminal:
            assert sum(1 << w for w in winners) == int(w2[0])
            break
    assert terminal and t > 50


def test_dropin_instances_are_independent_across_threads():
    """Several env instances live in one process under the GIL (MatchmakingServer.py:128-135): each owns its context, its
    host-mapped staging block and its stream, so four threads playing four games at once reproduce what each game gives
    when played alone."""
    import threading
    from colosseumrl_amd import get_environment
    names = {0: "forward", 1: "right", -1: "left"}

    def play(seed, out):
        env = get_environment("tron")("20;4")
        rng = np.random.default_rng(seed)
        state, players = env.new_state()
        trace = []
        for _ in range(60):
            acts = [names[int(x)] for x in rng.integers(-1, 2, size=4)]
            state, players, rewards, terminal, winners = env.next_state(state, [0, 1, 2, 3], acts)
            obs = env.state_to_observation(state, int(rng.integers(0, 4)))
            trace.append((state[1].tolist(), state[3].tolist(), rewards.tolist(), int(obs["board"].sum())))
            if terminal:
                state, players = env.new_state()
        out[seed] = trace

    alone = {}
    for seed in range(4):
        play(seed, alone)
    together = {}
    threads = [threading.Thread(target=play, args=(seed, together)) for seed in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert together == alone
