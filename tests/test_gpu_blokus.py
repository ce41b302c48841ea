"""HIP Blokus kernels vs games played by the reference itself (golden) and vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from backends import HipBlokus, OracleBlokus
from blokus_replay import GAMES, check_bonus, check_illegal, replay_games


def test_placement_table_matches_oracle():
    import ctypes as C
    from colosseumrl_amd import _native
    lib = _native.lib()
    for p in range(21):
        for o in range(8):
            want0 = O.blokus_placement(p, o, 0)
            for k in range(len(want0)):
                buf = (C.c_int8 * 10)()
                n = lib.crl_blokus_placement(p, o, k, buf)
                got = np.array(buf[:2 * n], np.int8).reshape(n, 2)
                assert np.array_equal(got, O.blokus_placement(p, o, k)), (p, o, k)


def test_reference_games_golden(golden):
    """valid_actions (count + ordered id list) and next_state over 8 complete reference games."""
    assert replay_games(golden, HipBlokus(len(GAMES))) >= 62


def test_last_piece_bonus_golden(golden):
    check_bonus(golden, HipBlokus)


def test_not_listed_actions_reference_golden(golden):
    """crl_blokus_step on the 364 actions of blokus_illegal.npz that reach the board in the reference -- overlaps on own and
    foreign cells (overwritten, still scored), cells wrapped by numpy's negative indices, indices off the board (extended
    ids), cells >= 20 / pieces not held (IndexError / ValueError codes, state untouched), passes by players who have moves --
    against the REFERENCE's own answers."""
    from colosseumrl_amd.envs.blokus import actions as A
    assert check_illegal(golden, HipBlokus, A.string_to_step_id) == 364


def test_step_arbitrary_ids_vs_oracle():
    """next_state on ARBITRARY ids -- dense, extended, beyond both, passes -- from mid-game states: step and the fused
    step_observe against the oracle (which the reference's answers pin, test_oracle_blokus.py), auto-reset on and off; then
    play on with legal moves from the boards the overwrites left behind."""
    import torch
    from colosseumrl_amd.batched import BlokusBatch
    B, seed = 1021, 23
    rng = np.random.default_rng(8)
    hip, orc = HipBlokus(B), OracleBlokus(B)
    hip.bb.rollout(30, seed)
    O.blokus_rollout(orc.st, seed, 0, 30)
    seen = set()
    for t in range(12):
        kind = rng.integers(0, 6, size=B)
        act = np.where(kind == 0, -1,
              np.where(kind <= 2, rng.integers(0, 336000, size=B),
              np.where(kind <= 4, rng.integers(336000, 336000 + 1344000, size=B), rng.integers(1680000, 2**31 - 1, size=B)))).astype(np.int32)
        if t % 2:                                                  # bias towards the board: small pieces near the middle wrap / overlap
            near = 336000 + ((rng.integers(0, 9, size=B) * 1600 + rng.integers(14, 26, size=B) * 40 + rng.integers(14, 26, size=B)) * 8
                             + rng.integers(0, 8, size=B)) * 5 + rng.integers(0, 3, size=B)
            act = np.where(kind == 3, near, act).astype(np.int32)
        s_before = orc.state()
        r2, t2, w2 = orc.step(act)
        if t % 3 == 2:
            out = hip.bb.step_observe(torch.from_numpy(act).cuda(), seed=seed, auto_reset=False)
            r1, t1, w1 = out["reward"].cpu().numpy(), out["terminal"].cpu().numpy(), out["winners"].cpu().numpy()
            cnt, _ = orc.valid(1)
            assert np.array_equal(out["n_valid"].cpu().numpy(), cnt), t
            assert np.array_equal(out["player"].cpu().numpy().reshape(-1), orc.st.to_move), t
        else:
            r1, t1, w1 = hip.step(act)
        assert np.array_equal(r1, r2) and np.array_equal(t1, t2) and np.array_equal(w1, w2), t
        s1, s2 = hip.state(), orc.state()
        for k in s1:
            assert np.array_equal(s1[k], s2[k]), (k, t)
        err = r2 < 0
        for k in s2:                                               # where the reference raises the game did not move
            assert np.array_equal(s2[k][err], s_before[k][err]), k
        seen |= set(r2[err].tolist())
        assert ((r2 >= 0) & (act >= 0)).sum() > 20, t              # and a good share of the ids did place
    assert seen == {-1, -2, -3}
    for t in range(20):                                            # legal play continues from whatever is on the boards now
        c1, ids1 = hip.valid(4096)
        c2, ids2 = orc.valid(4096)
        assert np.array_equal(c1, c2) and np.array_equal(ids1, ids2), t
        act = np.array([ids2[e, int(rng.integers(0, c2[e]))] if c2[e] else -1 for e in range(B)], np.int32)
        r1, t1, w1 = hip.step(act)
        r2, t2, w2 = orc.step(act)
        assert np.array_equal(r1, r2) and np.array_equal(t1, t2) and np.array_equal(w1, w2), t
    s1, s2 = hip.state(), orc.state()
    for k in s1:
        assert np.array_equal(s1[k], s2[k]), k


def test_step_and_valid_vs_oracle_random():
    """Lockstep random self-play of 96 games (ragged: not a multiple of 4 games per workgroup is covered by 97)."""
    B, T = 97, 80
    rng = np.random.default_rng(5)
    hip, orc = HipBlokus(B), OracleBlokus(B)
    for t in range(T):
        c1, ids1 = hip.valid(2048)
        c2, ids2 = orc.valid(2048)
        assert np.array_equal(c1, c2), t
        assert np.array_equal(ids1, ids2), t
        act = np.full(B, -1, np.int32)
        for e in range(B):
            if c2[e]:
                act[e] = ids2[e, int(rng.integers(0, c2[e]))]
        r1, t1, w1 = hip.step(act)
        r2, t2, w2 = orc.step(act)
        assert np.array_equal(r1, r2) and np.array_equal(t1, t2) and np.array_equal(w1, w2), t
        s1, s2 = hip.state(), orc.state()
        for k in s1:
            assert np.array_equal(s1[k], s2[k]), (k, t)
    # valid_actions for a player who is NOT to move
    pl = rng.integers(0, 4, size=B).astype(np.int8)
    c1, ids1 = hip.valid(2048, player=pl)
    c2, ids2 = orc.valid(2048, player=pl)
    assert np.array_equal(c1, c2) and np.array_equal(ids1, ids2)


@pytest.mark.parametrize("B,chunks", [(64, (30, 50)), (130, (100,)), (33, (7, 1, 80))])
def test_rollout_vs_oracle(B, chunks):
    """Fused random-agent rollout == oracle rollout bit for bit (action choice = r-th legal action in reference order)."""
    import torch
    from colosseumrl_amd.batched import BlokusBatch
    seed, first = 0xB10C05, 4242
    bb = BlokusBatch(B, first_env_id=first)
    ost = O.BlokusState(B)
    for T in chunks:
        bb.rollout(T, seed)
        O.blokus_rollout(ost, seed, first, T, n_threads=16)
    for k in ("occ", "inv", "score", "round", "to_move", "tcount", "tstep", "n_episodes", "win_count", "len_sum", "score_sum"):
        want = getattr(ost, k)
        assert np.array_equal(getattr(bb, k).cpu().numpy().view(want.dtype), want), k
    if sum(chunks) >= 80:
        assert ost.n_episodes.sum() > 0
    assert torch.equal(bb.results(), bb.results_from_columns())      # the row the kernel packs == the column statistics


def test_rollout_full_size_vs_oracle():
    """BASELINE config 4 at FULL size (B = 16,384) against the oracle itself: the first plies of every game (the oracle
    plays ~600 plies/s per core, so the full batch affords a handful of plies; longer games run at small B above)."""
    test_rollout_vs_oracle(16384, (5, 2))


def test_full_games_full_size_vs_oracle():
    """BASELINE config 4 at FULL size, WHOLE games: B = 16,384 games x 84 plies against the oracle on the host's threads.
    A random game lasts 62-74 plies, so every game ends at least once inside the launch: passes, the one-step terminal
    lag, the +15 / +20 last-piece bonuses, ties and the auto-reset are all compared at the full batch, every state array
    and every statistic bit for bit (round 2 compared the opening plies only at this size)."""
    import os
    import torch
    from colosseumrl_amd.batched import BlokusBatch
    B, seed, first = 16384, 0x5EED, 7 * 16384
    threads = max(8, min(32, os.cpu_count() or 8))
    bb = BlokusBatch(B, first_env_id=first)
    ost = O.BlokusState(B)
    for T in (60, 24):
        bb.rollout(T, seed)
        O.blokus_rollout(ost, seed, first, T, n_threads=threads)
    for k in ("occ", "inv", "score", "round", "to_move", "tcount", "tstep", "n_episodes", "win_count", "len_sum", "score_sum"):
        want = getattr(ost, k)
        assert np.array_equal(getattr(bb, k).cpu().numpy().view(want.dtype), want), k
    assert int(ost.n_episodes.min()) >= 1                                 # every game ended inside the launches
    assert int(ost.win_count.sum()) >= int(ost.n_episodes.sum())          # ties: more winners than games
    assert torch.equal(bb.results(), bb.results_from_columns())


def test_rollout_full_size_properties():
    """BASELINE config 4 size (B=16384): conservation laws of the fused rollout + shard invariance."""
    import torch
    from colosseumrl_amd.batched import BlokusBatch
    B, T, seed = 16384, 24, 9
    bb = BlokusBatch(B)
    bb.rollout(T, seed)
    occ = bb.occ.cpu().numpy().view(np.uint32)
    inv = bb.inv.cpu().numpy().view(np.uint32)
    score = bb.score.cpu().numpy()
    assert int(bb.len_sum.sum().item()) + int(bb.tstep.sum().item()) == B * T
    # colours never overlap; cells on the board per colour == cells of the pieces that left the inventory
    assert (occ[:, 0] & occ[:, 1]).max() == 0 and (occ[:, 2] & occ[:, 3]).max() == 0 and ((occ[:, 0] | occ[:, 1]) & (occ[:, 2] | occ[:, 3])).max() == 0
    sizes = np.array([1, 2, 3, 3, 4, 4, 4, 4, 4] + [5] * 12)
    played = ((~inv[:, :, None] >> np.arange(21)[None, None, :]) & 1) @ sizes
    cells = np.zeros((B, 4), np.int64)
    for c in range(4):
        cells[:, c] = sum(((occ[:, c, :] >> x) & 1).sum(axis=1) for x in range(20))
    assert np.array_equal(cells, played) and np.array_equal(score, played)       # no bonus possible in 24 steps
    half = BlokusBatch(B // 2, first_env_id=B // 2)
    half.rollout(T, seed)
    assert torch.equal(half.occ, bb.occ[B // 2:]) and torch.equal(half.score, bb.score[B // 2:])


def test_dropin_env_golden(golden):
    """BlokusEnvironment drop-in: strings in, (Board, round, [AI]) out, replaying a reference game."""
    from colosseumrl_amd import get_environment
    from colosseumrl_amd.envs.blokus import actions as A
    g = golden("blokus_game_4")
    env = get_environment("blokus")()
    state, players = env.new_state()
    T = len(g["action"])
    for t in range(T):
        va = env.valid_actions(state, players[0])
        assert len([a for a in va if a]) == g["n_valid"][t]
        where = np.nonzero(g["list_step"] == t)[0]
        if len(where) and t % 12 == 0:
            assert va == [A.id_to_string(i) for i in g["lists"][where[0]][:g["n_valid"][t]]] or (va == [""] and g["n_valid"][t] == 0)
        astr = A.id_to_string(int(g["action"][t]))
        if astr and t % 9 == 0:
            assert env.is_valid_action(state, players[0], astr)
            d = env.valid_actions_dict(state, players[0])
            piece, idx, orient = A.string_to_action(astr)
            assert orient in d[piece][idx]
        state, players, rewards, terminal, winners = env.next_state(state, players, [astr])
        assert np.array_equal(state[0].board_contents, g["board"][t]) and state[0].board_contents.dtype == np.int64
        assert [p.player_score for p in state[2]] == g["score"][t].tolist() and state[1] == g["round"][t]
        assert [p.inventory_mask() for p in state[2]] == g["inv"][t].tolist()
        assert players == [int(g["next_player"][t])] and rewards == [int(g["reward"][t])] and terminal == bool(g["terminal"][t])
        assert (winners is None and not terminal) or sum(1 << w for w in winners) == g["winners"][t]
    assert terminal
    assert env.is_valid_action(state, 0, "") is False
    assert env.is_valid_action(env.new_state()[0], 0, "monomino1;(5, 5);north0") is False
    with pytest.raises(ValueError):
        env.next_state(state, [0], [A.id_to_string(int(g["action"][0]))])      # piece already played


def test_observe_golden_and_oracle(golden):
    import torch
    g = golden("blokus_observe")
    n = len(g["player"])
    be = HipBlokus(n)
    be.set_state(g["board"], g["inv"], g["score"], g["round"], np.zeros(n, np.int32))
    obs = be.bb.observe(torch.from_numpy(g["player"].astype(np.int8)).cuda())
    assert np.array_equal(obs["board"].cpu().numpy(), g["obs_board"])
    assert np.array_equal(obs["pieces"].cpu().numpy(), g["obs_pieces"])
    assert np.array_equal(obs["score"].cpu().numpy(), g["obs_score"])
    # larger random check against the oracle after a rollout
    from colosseumrl_amd.batched import BlokusBatch
    bb = BlokusBatch(2048)
    bb.rollout(30, 5)
    pl = torch.randint(0, 4, (2048,), dtype=torch.int8, device="cuda")
    o = bb.observe(pl)
    st = O.BlokusState(2048)
    st.occ[:] = bb.occ.cpu().numpy().view(np.uint32)
    st.inv[:] = bb.inv.cpu().numpy().view(np.uint32)
    st.score[:] = bb.score.cpu().numpy()
    ob, op, osc = O.blokus_observe(st, pl.cpu().numpy())
    assert np.array_equal(o["board"].cpu().numpy(), ob) and np.array_equal(o["pieces"].cpu().numpy(), op)
    assert np.array_equal(o["score"].cpu().numpy(), osc)


def test_dropin_observation_golden(golden):
    from colosseumrl_amd.envs.blokus import actions as A
    from colosseumrl_amd.envs.blokus.BlokusEnvironment import BlokusEnvironment
    from colosseumrl_amd.envs.blokus.ai import AI
    from colosseumrl_amd.envs.blokus.board import Board
    g = golden("blokus_observe")
    env = BlokusEnvironment()
    for i in range(0, len(g["player"]), 5):
        b = Board()
        b.board_contents = np.asarray(g["board"][i], dtype=np.int64)
        players = []
        for c in range(4):
            ai = AI(b, c + 1)
            ai.player_score = int(g["score"][i][c])
            ai.current_pieces = [n for k, n in enumerate(A.PIECE_NAMES) if (int(g["inv"][i][c]) >> k) & 1]
            players.append(ai)
        obs = env.state_to_observation((b, int(g["round"][i]), players), int(g["player"][i]))
        assert np.array_equal(obs["board"], g["obs_board"][i]) and obs["board"].shape == (20, 20)
        assert np.array_equal(obs["pieces"], g["obs_pieces"][i]) and obs["pieces"].dtype == np.uint8
        assert np.array_equal(obs["score"], g["obs_score"][i]) and obs["player"].tolist() == [int(g["player"][i])]


def test_vector_env_adapter_vs_oracle():
    """BlokusVectorEnv (reference envs/wrappers/rllib.py:37-55 reset/step contract) played next to the oracle
    in lockstep until games finish and restart: observation of the mover, mover, legal-action counts, reward, done, winners."""
    import torch
    from colosseumrl_amd.vector import BlokusVectorEnv
    B = 24
    env = BlokusVectorEnv(B)
    orc = OracleBlokus(B)
    obs, mover, n_valid = env.reset()
    assert obs["board"].shape == (B, 20, 20) and int(n_valid[0]) == 116 and mover.tolist() == [0] * B
    rng = np.random.default_rng(1)
    finished = 0
    for t in range(100):
        c2, ids2 = orc.valid(2048)
        assert np.array_equal(n_valid.cpu().numpy(), c2), t
        act = np.full(B, -1, np.int32)
        for e in range(B):
            if c2[e]:
                # bias towards big pieces so that games end within the test
                act[e] = ids2[e, int(c2[e] - 1 - rng.integers(0, max(1, c2[e] // 3)))]
        if t == 0:
            assert np.array_equal(env.sample_valid(3).cpu().numpy() >= 0, c2 > 0)
        obs, mover, n_valid, rew, done, info = env.step(torch.from_numpy(act).cuda(), list_cap=2048 if t % 3 == 0 else 0)
        r2, t2, w2 = orc.step(act)
        assert np.array_equal(rew.cpu().numpy(), r2) and np.array_equal(done.cpu().numpy(), t2) and np.array_equal(info["winners"].cpu().numpy(), w2), t
        if t2.any():
            finished += int(t2.sum())
            orc.reset(t2)
        s2 = orc.state()
        assert np.array_equal(mover.cpu().numpy(), s2["to_move"].astype(np.int8)), t
        ob, op, osc = O.blokus_observe(orc.st, s2["to_move"].astype(np.int8))
        assert np.array_equal(obs["board"].cpu().numpy(), ob) and np.array_equal(obs["pieces"].cpu().numpy(), op)
        assert np.array_equal(obs["score"].cpu().numpy(), osc)
        if t % 3 == 0:                                          # the next mover's ordered ids, queued behind the step
            c3, ids3 = orc.valid(2048)
            assert np.array_equal(n_valid.cpu().numpy(), c3) and np.array_equal(info["ids"].cpu().numpy(), ids3), t
        else:
            assert "ids" not in info
    assert finished > 0


def test_step_observe_fused_matches_separate_calls():
    """crl_blokus_step_observe (one launch: [sample ->] next_state -> legal-action count + observation of the next mover)
    equals crl_blokus_sample + crl_blokus_step + crl_blokus_valid + crl_blokus_observe on a twin batch, through whole games:
    sampled and external (legal) actions, passes, auto-reset on and off, a ragged batch (not a multiple of 4 games)."""
    import torch
    from colosseumrl_amd.batched import BlokusBatch
    B, seed, first = 45, 17, 300
    a, b = BlokusBatch(B, first_env_id=first), BlokusBatch(B, first_env_id=first)
    out = None
    finished = 0
    for t in range(120):
        auto = (t % 7) != 6
        act = b.sample(seed)                       # a legal action (or -1) for every game; advances b.tcount
        if t % 3 == 0:
            out = a.step_observe(None, seed=seed, auto_reset=auto, out=out)      # the fused call samples the same action itself
        else:
            a.tcount.copy_(b.tcount)                                              # external actions do not touch the counter
            out = a.step_observe(act, seed=seed, auto_reset=auto, out=out)
        r, tm, w = b.step(act, auto_reset=auto)
        for k in ("occ", "inv", "score", "round", "to_move", "tcount"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (k, t)
        assert torch.equal(out["reward"], r) and torch.equal(out["terminal"], tm) and torch.equal(out["winners"], w), t
        assert torch.equal(out["n_valid"], b.valid()), t
        mover = b.to_move.to(torch.int8)
        assert torch.equal(out["player"].view(-1), mover), t
        o2 = b.observe(mover)
        assert torch.equal(out["board"], o2["board"]) and torch.equal(out["pieces"], o2["pieces"]) and torch.equal(out["score"], o2["score"]), t
        finished += int(tm.sum())
    assert finished > 0
