"""HIP TicTacToe kernels vs golden vectors from the reference and vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from backends import HipTTT, OracleTTT
from replay import replay_ttt
from ttt_tree import count_games


@pytest.mark.parametrize("name", ["2p_reset", "2p_noreset", "3p_reset", "3p_noreset", "4p_reset", "4p_noreset"])
def test_traj_golden(golden, name):
    assert replay_ttt(golden("ttt_traj_" + name), HipTTT) > 0


def test_2p_exhaustive_tree_kat():
    assert count_games(HipTTT) == (255168, 131184, 77904, 46080)


@pytest.mark.parametrize("dims,K,P", [((3, 3), 3, 2), ((3, 5), 3, 3), ((3, 3, 3), 3, 4), ((5, 5), 4, 3), ((4, 8), 4, 5), ((2, 4, 4), 3, 8)])
def test_lines_and_step_vs_oracle(dims, K, P):
    B, T = 3001, 45
    hip, orc = HipTTT(dims, K, P, B), OracleTTT(dims, K, P, B)
    assert hip.lines() == orc.lines()
    n_cells = int(np.prod(dims))
    rng = np.random.default_rng(P * 100 + n_cells)
    for t in range(T):
        a = rng.integers(-1, n_cells, size=B).astype(np.int8)
        # steer most moves to empty cells so games progress
        v = orc.valid()
        for e in range(0, B, 2):
            cells = [c for c in range(n_cells) if (int(v[e]) >> c) & 1]
            if cells:
                a[e] = cells[int(rng.integers(0, len(cells)))]
        auto = (t % 4) != 3
        r1, t1, w1 = hip.step(a, auto_reset=auto)
        r2, t2, w2 = orc.step(a, auto_reset=auto)
        assert np.array_equal(r1, r2) and np.array_equal(t1, t2) and np.array_equal(w1, w2), t
        assert np.array_equal(hip.board(), orc.board()) and np.array_equal(hip.winner(), orc.winner())
        assert np.array_equal(hip.to_move(), orc.to_move()) and np.array_equal(hip.valid(), orc.valid())


@pytest.mark.parametrize("dims,K,P,B,chunks", [((3, 3), 3, 2, 5000, (64, 13)), ((3, 5), 3, 3, 4099, (100,)),
                                                ((3, 3, 3), 3, 4, 2048, (80,)), ((5, 5), 4, 3, 8192, (64, 64))])
def test_rollout_vs_oracle(dims, K, P, B, chunks):
    seed, first = 0xABCDEF0123, 555
    hip = HipTTT(dims, K, P, B)
    hip.tb.first_env_id = first
    ost = O.TTTState(dims, K, P, B)
    for T in chunks:
        hip.tb.rollout(T, seed)
        O.ttt_rollout(ost, seed, first, T, n_threads=8)
    tb = hip.tb
    for k in ("occ", "winner", "to_move", "tcount", "tstep", "n_episodes", "win_count", "draw_count", "len_sum"):
        want = getattr(ost, k)
        assert np.array_equal(getattr(tb, k).cpu().numpy().view(want.dtype), want), k
    assert ost.n_episodes.sum() > B
    import torch
    assert torch.equal(tb.results(), tb.results_from_columns())      # the row the kernel packs == the column statistics


@pytest.mark.parametrize("dims,K,P,B,chunks", [((3, 3), 3, 2, 1500, (600, 1003, 5)), ((3, 3, 3), 3, 4, 700, (1024, 8)),
                                                ((4, 8), 4, 5, 600, (520, 3, 253)), ((2, 4, 4), 3, 8, 500, (504, 77, 300)),
                                                ((2, 3), 2, 1, 300, (250, 250)), ((4, 4), 3, 7, 300, (999,))])
def test_rollout_long_launches_every_player_count(dims, K, P, B, chunks):
    """The rollout kernel counts outcomes in byte fields that it flushes every 248 plies (32-bit fields up to 3 players, 64-bit
    up to 7, 8 players without a draw field) and draws one Philox block per 8 plies: launches long enough to cross several
    flushes, lengths that leave the step counter off a multiple of 8, and every accumulator variant -- against the oracle."""
    test_rollout_vs_oracle(dims, K, P, B, chunks)


@pytest.mark.parametrize("dims,K,P,B,chunks", [((3, 3), 3, 2, 3000, (64, 13, 200)), ((3, 5), 3, 3, 2049, (100, 7)), ((4, 4), 4, 2, 500, (96,))])
def test_rollout_small_boards_without_the_win_table(dims, K, P, B, chunks, monkeypatch):
    """Boards of <= 16 cells normally run the rollout instance whose win test is a lookup in the context's table of winning
    masks; a context without that table (no device at create time, or a rollout on another device) falls back to the
    shift-and test.  CRL_TTT_NO_WIN_TABLE makes crl_ttt_create skip the table: the fallback against the oracle."""
    monkeypatch.setenv("CRL_TTT_NO_WIN_TABLE", "1")
    test_rollout_vs_oracle(dims, K, P, B, chunks)


@pytest.mark.parametrize("dims,K,P,B", [((3, 3), 3, 2, 3000), ((3, 5), 3, 3, 1000 + 7), ((2, 2, 2), 2, 4, 300)])
def test_rollout_from_finished_states_that_were_not_restarted(dims, K, P, B):
    """A rollout may come in on states the step API left FINISHED (stepped without auto-reset: sticky winner, full board)
    and at odd step counters: the kernel's first ply is then the general one (no move, terminal, restart), every later one
    the ply of a running game.  Against the oracle, which has only the general ply."""
    rng = np.random.default_rng(11)
    hip, orc = HipTTT(dims, K, P, B), OracleTTT(dims, K, P, B)
    n_cells = int(np.prod(dims))
    for t in range(n_cells + 2):                           # random legal-or-not moves, NO auto-reset: most games end
        act = rng.integers(-1, n_cells, size=B).astype(np.int8)
        r1, t1, w1 = hip.step(act)
        r2, t2, w2 = orc.step(act)
        assert np.array_equal(t1, t2)
    assert (orc.winner() >= 0).any() and (orc.winner() < 0).any()
    ost = orc.st
    seed, first = 31337, 12
    hip.tb.first_env_id = first
    tc = rng.integers(0, 50, size=B).astype(ost.tcount.dtype)
    ost.tcount[:] = tc
    import torch
    hip.tb.tcount.copy_(torch.from_numpy(tc.view(np.int32)))
    for T in (1, 1, 30, 64):
        hip.tb.rollout(T, seed)
        O.ttt_rollout(ost, seed, first, T, n_threads=8)
        for k in ("occ", "winner", "to_move", "tcount", "tstep", "n_episodes", "win_count", "draw_count", "len_sum"):
            want = getattr(ost, k)
            assert np.array_equal(getattr(hip.tb, k).cpu().numpy().view(want.dtype), want), (k, T)


@pytest.mark.parametrize("dims,K,P", [((5, 5), 4, 3), ((3, 5), 3, 3)])
def test_rollout_full_size_vs_oracle(dims, K, P):
    """BASELINE config 3 at FULL size (B = 262,144; 5x5 K4 and the reference-pinned 3x5 K3) against the oracle itself."""
    test_rollout_vs_oracle(dims, K, P, 262144, (24, 8))


def test_rollout_full_size_properties():
    """BASELINE config 3 size: 3-player 5x5 K=4, B=262144."""
    import torch
    from colosseumrl_amd.batched import TTTBatch
    B, T, seed = 262144, 100, 3
    tb = TTTBatch((5, 5), 4, 3, B)
    tb.rollout(T, seed)
    assert int(tb.len_sum.sum().item()) + int(tb.tstep.sum().item()) == B * T
    n_ep = tb.n_episodes.cpu().numpy().astype(np.int64)
    assert (tb.win_count.cpu().numpy().sum(axis=0) + tb.draw_count.cpu().numpy() == n_ep).all()
    occ = tb.occ.cpu().numpy().view(np.uint32)
    assert (occ[0] & occ[1]).max() == 0 and (occ[0] & occ[2]).max() == 0 and (occ[1] & occ[2]).max() == 0
    half = TTTBatch((5, 5), 4, 3, B // 2, first_env_id=B // 2)
    half.rollout(T, seed)
    assert torch.equal(half.occ, tb.occ[:, B // 2:]) and torch.equal(half.win_count, tb.win_count[:, B // 2:])


@pytest.mark.parametrize("name,env_name", [("2p", "tictactoe"), ("3p", "tictactoe_3p"), ("4p", "tictactoe_4p")])
def test_dropin_env_golden(golden, name, env_name):
    from colosseumrl_amd import get_environment
    g = golden("ttt_traj_%s_noreset" % name)
    shape = tuple(int(x) for x in g["shape"])
    env = get_environment(env_name)()
    for e in range(4):
        state, players = env.new_state()
        for t in range(int(g["T"])):
            cell = int(g["action"][t, e])
            astr = "" if cell < 0 else str(tuple(int(i) for i in np.unravel_index(cell, shape)))
            va = env.valid_actions(state, players[0])
            want = [str(tuple(int(i) for i in np.unravel_index(c, shape))) for c in range(int(np.prod(shape))) if (int(g["valid"][t, e]) >> c) & 1]
            assert va == (want if want else [""])
            if cell >= 0:
                assert env.is_valid_action(state, players[0], astr) == ((int(g["valid"][t, e]) >> cell) & 1 == 1)
            state, players, rewards, terminal, winners = env.next_state(state, players, [astr])
            assert state[0].dtype == np.int8 and np.array_equal(state[0].ravel(), g["board"][t, e])
            assert (state[1] is None and g["winner"][t, e] < 0) or state[1] == g["winner"][t, e]
            assert players == [int(g["next_player"][t, e])] and rewards == [int(g["reward"][t, e])]
            assert terminal == bool(g["terminal"][t, e])
            assert (winners is None and g["winners"][t, e] < 0) or winners == [int(g["winners"][t, e])]
    # python indexing semantics of the reference: negative indices wrap, out of range raises
    state, players = env.new_state()
    neg = "(" + ", ".join(["-1"] * len(shape)) + ")"
    s2, *_ = env.next_state(state, [0], [neg])
    assert s2[0].ravel()[-1] == 0
    with pytest.raises(IndexError):
        env.next_state(state, [0], ["(" + ", ".join(["7"] * len(shape)) + ")"])
    assert env.is_valid_action(state, 0, "") is False


@pytest.mark.parametrize("name,env_name", [("2p", "tictactoe"), ("3p", "tictactoe_3p"), ("4p", "tictactoe_4p")])
def test_observation_golden(golden, name, env_name):
    from colosseumrl_amd import get_environment
    g = golden("ttt_observe_" + name)
    env = get_environment(env_name)()
    shape = tuple(int(x) for x in g["shape"])
    for b, pl, want in list(zip(g["board"], g["player"], g["obs_board"]))[:24]:
        obs = env.state_to_observation((b.reshape(shape), None), int(pl))
        assert np.array_equal(obs["board"].ravel(), want)


def test_vector_env_adapters_vs_oracle():
    """The vector-env adapters (colosseumrl_amd.vector; reference envs/wrappers/rllib.py:37-55 reset/step contract) played
    for whole episodes next to the oracle stepped in lockstep: observations, rewards, dones, winners, auto-reset."""
    import torch
    from backends import OracleTron
    from colosseumrl_amd.vector import TicTacToeVectorEnv, TronVectorEnv
    N, P, B = 12, 4, 256 + 9
    env = TronVectorEnv(N, P, B)
    sh, sd = O.tron_start_positions(N, P)
    orc = OracleTron(N, P, B, sh, sd)
    obs = env.reset()
    assert obs[0]["board"].shape == (B, N, N) and int(obs[2]["board"].max()) == P
    rng = np.random.default_rng(0)
    done_total = 0
    for t in range(40):
        a = rng.integers(-1, 2, size=(P, B)).astype(np.int8)
        obs, rew, done, info = env.step(torch.from_numpy(a).cuda())
        r2, t2, w2 = orc.step(a, auto_reset=True)
        assert np.array_equal(rew.cpu().numpy(), r2) and np.array_equal(done.cpu().numpy(), t2)
        assert np.array_equal(info["winners"].cpu().numpy(), w2)
        for p in range(P):
            ob, oh, od, ok = O.tron_observe(orc.st, np.full(B, p, np.int8))
            assert np.array_equal(obs[p]["board"].reshape(B, -1).cpu().numpy(), ob) and np.array_equal(obs[p]["heads"].cpu().numpy(), oh)
            assert np.array_equal(obs[p]["directions"].cpu().numpy(), od) and np.array_equal(obs[p]["deaths"].cpu().numpy(), ok)
        done_total += int(t2.sum())
    assert done_total > B

    dims, K, PT, BT = (3, 5), 3, 3, 512 + 3
    tenv = TicTacToeVectorEnv(dims, K, PT, BT)
    ot = OracleTTT(dims, K, PT, BT)
    obs, mover, valid = tenv.reset()
    assert int(valid[0]) == (1 << 15) - 1 and obs["board"].shape == (BT, 15)
    finished = 0
    for t in range(40):
        empties = valid.cpu().numpy().view(np.uint32)
        assert np.array_equal(empties, ot.valid())
        act = np.full(BT, -1, np.int8)
        for e in range(BT):
            cells = [c for c in range(15) if (int(empties[e]) >> c) & 1]
            if cells and rng.random() < 0.95:
                act[e] = cells[int(rng.integers(0, len(cells)))]
            elif rng.random() < 0.5:
                act[e] = int(rng.integers(0, 15))                      # sometimes an occupied cell: a no-op that passes the turn
        obs, mover, valid, rew, done, info = tenv.step(torch.from_numpy(act).cuda())
        r2, t2, w2 = ot.step(act, auto_reset=True)
        assert np.array_equal(rew.cpu().numpy(), r2) and np.array_equal(done.cpu().numpy(), t2) and np.array_equal(info["winners"].cpu().numpy(), w2)
        assert np.array_equal(mover.cpu().numpy(), ot.to_move())
        ab = ot.board().reshape(BT, -1).astype(np.int64)                # absolute ids; the observation is relative to the mover
        want = np.where(ab >= 0, (ab - ot.to_move().astype(np.int64)[:, None]) % PT, -1)     # reference 2p:26-27,382-407
        assert np.array_equal(obs["board"].cpu().numpy().astype(np.int64), want)
        finished += int(t2.sum())
    assert finished > BT


@pytest.mark.parametrize("dims,K,P,B,rel_mod", [((3, 3), 3, 2, 1000 + 7, None), ((3, 5), 3, 3, 513, None), ((3, 3, 3), 3, 4, 300, 3),
                                                ((5, 5), 4, 3, 4096 + 1, None), ((1, 1), 2, 2, 70, None), ((4, 8), 4, 5, 257, None)])
def test_step_observe_fused_matches_separate_calls(dims, K, P, B, rel_mod):
    """crl_ttt_step_observe (one launch: [sample ->] next_state -> valid mask + observation of the next mover) equals
    crl_ttt_sample + crl_ttt_step + crl_ttt_valid + crl_ttt_board on a twin batch: sampled and external actions (incl.
    occupied cells and passes), auto-reset on and off, ragged batches, the 4-player `% 3` observation quirk, a one-cell board."""
    import torch
    from colosseumrl_amd.batched import TTTBatch
    seed, first = 31, 4000
    a, b = TTTBatch(dims, K, P, B, first_env_id=first), TTTBatch(dims, K, P, B, first_env_id=first)
    n_cells = int(np.prod(dims))
    rng = np.random.default_rng(n_cells + P)
    out = None
    for t in range(3 * n_cells + 5):
        auto = (t % 4) != 3
        if t % 2 == 0:
            act = b.sample(seed)
            out = a.step_observe(None, seed=seed, auto_reset=auto, rel_mod=rel_mod, out=out)
        else:
            act = torch.from_numpy(rng.integers(-1, n_cells, size=B).astype(np.int8)).cuda()
            out = a.step_observe(act, seed=seed, auto_reset=auto, rel_mod=rel_mod, out=out)
        r, tm, w = b.step(act, auto_reset=auto)
        for k in ("occ", "winner", "to_move", "tcount"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (k, t)
        assert torch.equal(out["reward"], r) and torch.equal(out["terminal"], tm) and torch.equal(out["winners"], w), t
        assert torch.equal(out["valid"], b.valid_mask()), t
        assert torch.equal(out["board"], b.board(b.to_move, rel_mod)), t
    assert int(b.terminal.sum()) >= 0


@pytest.mark.parametrize("dims,K,P,rel_mod", [((3, 3), 3, 2, 2), ((3, 5), 3, 3, 3), ((3, 3, 3), 3, 4, 4), ((5, 5), 4, 3, 3), ((4, 8), 4, 8, 8)])
def test_single_state_one_call_form_matches_the_two_call_form(dims, K, P, rel_mod):
    """crl_ttt_step_board_host (the state by value in the kernel arguments, completion published by the kernel, one blocking
    call) against crl_ttt_step_board + crl_stream_wait_mapped on the same states: random games incl. occupied / empty / ''
    actions and steps after the game has ended."""
    from colosseumrl_amd.single import SingleTTT
    rng = np.random.default_rng(sum(dims) + P)
    one, two = SingleTTT(dims, K, P, rel_mod), SingleTTT(dims, K, P, rel_mod)
    assert one._unified
    two._unified = False
    n = int(np.prod(dims))
    for game in range(30):
        board, winner, mover = np.full(n, -1, np.int8), None, 0
        for ply in range(n + 4):
            cell = int(rng.integers(-1, n))                       # -1 = '', occupied cells included
            outs = []
            for st in (one, two):
                st.load(board, winner, mover)
                st.step(cell)
                v = st.v
                outs.append((v["board"].copy(), int(v["winner"][0]), int(v["to_move"][0]), int(v["reward"][0]), int(v["terminal"][0]),
                             int(v["winners"][0]), int(v["valid"][0]), v["obs_board"].copy()))
            a, b = outs
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (game, ply, cell)
            board, winner, mover = a[0], (None if a[1] < 0 else a[1]), a[2]
