"""Longer randomized differential runs (HIP fused rollouts vs the CPU oracle, bit for bit): rare paths such as tag
wrap-around, > 64 anchors, dead-player caching, every terminal flavour.  ~30 s on a GPU box."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O


@pytest.mark.parametrize("N,P,B,T,kernel", [(20, 4, 4096, 1500, "quad"), (13, 3, 1024 + 9, 1200, "quad"), (6, 2, 512, 3000, "quad"), (8, 4, 777, 2000, "quad"),
                                            (20, 4, 4096, 1500, "bits"), (20, 4, 4096, 1500, "bytes"), (20, 4, 2048, 1000, "global"),
                                            (13, 7, 1024, 1200, "bits"), (13, 7, 1024, 1200, "bytes"), (30, 3, 1024, 800, "bits"),
                                            (30, 3, 1024, 800, "bytes"), (6, 2, 512, 3000, "bits"), (6, 2, 512, 3000, "bytes"),
                                            (12, 8, 512, 1500, "bits"), (12, 8, 512, 1500, "bytes"), (23, 4, 512, 1200, "auto"),
                                            (16, 5, 700, 1200, "bytes"), (40, 4, 1000, 900, "bits"), (40, 4, 1000, 900, "bytes"),
                                            (36, 8, 300, 700, "bits"), (36, 8, 300, 700, "bytes"), (40, 4, 512, 600, "global"), (40, 4, 1000, 900, "gquad"), (20, 4, 4096, 1500, "gquad"), (13, 3, 1024 + 9, 1200, "gquad"), (6, 2, 512, 3000, "gquad"),
                                            (50, 4, 300, 700, "gquad"), (45, 2, 130, 17000, "gquad"), (19, 1, 200, 500, "gquad"), (20, 2, 4096, 1500, "pair"), (13, 2, 1024 + 9, 1200, "pair"), (6, 2, 512, 3000, "pair"), (19, 1, 300, 900, "pair"),
                                            (9, 2, 130, 40001, "pair"), (16, 2, 777, 2000, "pair"),
                                            (40, 4, 1000, 900, "qbits"), (30, 3, 1024 + 7, 800, "qbits"), (20, 4, 4096, 1500, "qbits"),
                                            (6, 2, 512, 3000, "qbits"), (13, 3, 1024 + 9, 1200, "qbits"), (33, 2, 130, 20000, "qbits"),
                                            (9, 3, 130, 40001, "quad")])      # (> 16,383 steps: the lane-per-player kernels split the launch)
def test_tron_long_rollout(N, P, B, T, kernel):
    from colosseumrl_amd.batched import TronBatch
    seed, first = 0x5EED + N, 10 ** 6
    tb = TronBatch(N, P, B, first_env_id=first)
    tb.rollout(T, seed, kernel=kernel)
    sh, sd = O.tron_start_positions(N, P)
    ost = O.TronState(N, P, B)
    O.tron_reset(ost, sh, sd)
    O.tron_rollout(ost, seed, first, T, sh, sd, n_threads=16)
    for k in ("board", "heads", "dirs", "deaths", "tcount", "tstep", "n_episodes", "win_count", "len_sum", "ret_sum",
              "last_winners", "last_len"):
        want = getattr(ost, k)
        assert np.array_equal(getattr(tb, k).cpu().numpy().view(want.dtype), want), k
    assert int(ost.n_episodes.min()) > 32          # every game wrapped its 5-bit episode tag at least once


def test_tron_random_configurations_all_kernels():
    """Randomised differential run (tools/debug/tron_fuzz.py with a fixed seed, 150 configurations): board sizes 4..40,
    2..8 players, ragged batches whose last workgroup has waves wholly beyond the batch (a 3,000-configuration run of this
    found an out-of-bounds read there), split launches, launches longer than the lane-per-player kernels' step limit --
    every rollout kernel against the oracle, bit for bit."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    rng = np.random.default_rng(424242)
    for case in range(150):
        N, P, B = int(rng.integers(4, 41)), int(rng.integers(2, 9)), int(rng.integers(1, 3000))
        P = min(P, 4) if N == 4 else P                  # (a 4x4 board's start ring holds at most 4 players)
        chunks = [int(rng.integers(1, 700)) for _ in range(int(rng.integers(1, 4)))]
        if rng.random() < 0.1:
            chunks.append(int(rng.integers(16384, 18000)))
            B = min(B, 200)
        seed, first = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 40))
        sh, sd = O.tron_start_positions(N, P)
        ost = O.TronState(N, P, B)
        O.tron_reset(ost, sh, sd)
        for T in chunks:
            O.tron_rollout(ost, seed, first, T, sh, sd, n_threads=16)
        for kernel in ("auto", "quad", "pair", "qbits", "bytes", "bits", "global", "gquad"):
            tb = TronBatch(N, P, B, first_env_id=first)
            for T in chunks:
                tb.rollout(T, seed, kernel=kernel)
            for k in ("board", "heads", "dirs", "deaths", "tstep", "n_episodes", "win_count", "len_sum", "ret_sum", "last_winners", "last_len"):
                want = getattr(ost, k)
                assert np.array_equal(getattr(tb, k).cpu().numpy().view(want.dtype), want), (case, N, P, B, chunks, kernel, k)


def test_blokus_long_rollout():
    from colosseumrl_amd.batched import BlokusBatch
    B, T, seed, first = 768, 160, 20240, 5000
    bb = BlokusBatch(B, first_env_id=first)
    ost = O.BlokusState(B)
    for chunk in (100, 60):
        bb.rollout(chunk, seed)
        O.blokus_rollout(ost, seed, first, chunk, n_threads=16)
    for k in ("occ", "inv", "score", "round", "to_move", "tcount", "tstep", "n_episodes", "win_count", "len_sum", "score_sum"):
        want = getattr(ost, k)
        assert np.array_equal(getattr(bb, k).cpu().numpy().view(want.dtype), want), k
    assert int(ost.n_episodes.sum()) >= B * 2


@pytest.mark.parametrize("dims,K,P", [((3, 3), 3, 2), ((5, 5), 4, 3), ((3, 3, 3), 3, 4), ((4, 8), 5, 8)])
def test_ttt_long_rollout(dims, K, P):
    from colosseumrl_amd.batched import TTTBatch
    B, T, seed, first = 16384, 2000, 99, 123
    tb = TTTBatch(dims, K, P, B, first_env_id=first)
    tb.rollout(T, seed)
    ost = O.TTTState(dims, K, P, B)
    O.ttt_rollout(ost, seed, first, T, n_threads=16)
    for k in ("occ", "winner", "to_move", "tcount", "tstep", "n_episodes", "win_count", "draw_count", "len_sum"):
        want = getattr(ost, k)
        assert np.array_equal(getattr(tb, k).cpu().numpy().view(want.dtype), want), k


def test_ttt_and_blokus_random_configurations():
    """The differential fuzz of tools/debug/ttt_blokus_fuzz.py as part of the suite (as was done for Tron in round 2): 600
    random TicTacToe configurations (1-3 dimensions, up to 32 cells, K 1-6, 2-8 players, ragged batches, split launches)
    and 60 Blokus ones, every state array and statistic against the oracle."""
    from colosseumrl_amd.batched import BlokusBatch, TTTBatch
    rng = np.random.default_rng(20261004)
    done = 0
    for case in range(600):
        nd = int(rng.integers(1, 4))
        while True:
            dims = tuple(int(rng.integers(1, 9)) for _ in range(nd))
            if 1 <= int(np.prod(dims)) <= 32:
                break
        K, P, B = int(rng.integers(1, 7)), int(rng.integers(2, 9)), int(rng.integers(1, 3000))
        chunks = [int(rng.integers(1, 500)) for _ in range(int(rng.integers(1, 4)))]
        seed, first = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 40))
        try:
            ost = O.TTTState(dims, K, P, B)
            tb = TTTBatch(dims, K, P, B, first_env_id=first)
        except Exception:                                   # shapes the library rejects (too many win lines, K out of range)
            continue
        for T in chunks:
            O.ttt_rollout(ost, seed, first, T, n_threads=16)
            tb.rollout(T, seed)
        for k in ("occ", "winner", "to_move", "tcount", "tstep", "n_episodes", "win_count", "draw_count", "len_sum"):
            want = getattr(ost, k)
            assert np.array_equal(getattr(tb, k).cpu().numpy().view(want.dtype), want), (case, dims, K, P, B, chunks, k)
        done += 1
    assert done > 450
    for case in range(60):
        B = int(rng.integers(1, 400))
        chunks = [int(rng.integers(1, 100)) for _ in range(int(rng.integers(1, 4)))]
        seed, first = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 40))
        bb = BlokusBatch(B, first_env_id=first)
        ost = O.BlokusState(B)
        for T in chunks:
            bb.rollout(T, seed)
            O.blokus_rollout(ost, seed, first, T, n_threads=16)
        for k in ("occ", "inv", "score", "round", "to_move", "tcount", "tstep", "n_episodes", "win_count", "len_sum", "score_sum"):
            want = getattr(ost, k)
            assert np.array_equal(getattr(bb, k).cpu().numpy().view(want.dtype), want), (case, B, chunks, k)


def test_blokus_lists_on_arbitrary_boards():
    """valid_list / valid / select / is_valid on boards no game produces (random cells of random colours at densities from
    1 % to 70 %, random inventories, round 0 and later, every player) against the oracle: the window x pattern list kernel
    and the row-bitboard count / select kernels answer for ANY board `set_board` accepts, not only reachable ones."""
    import torch
    from colosseumrl_amd.batched import BlokusBatch
    rng = np.random.default_rng(41)
    B, cap = 192, 16384
    for trial in range(6):
        density = rng.choice([0.01, 0.05, 0.15, 0.3, 0.5, 0.7], size=B)
        board = ((rng.random((B, 20, 20)) < density[:, None, None]) * rng.integers(1, 5, (B, 20, 20))).astype(np.int8)
        if trial == 0:
            board[: B // 2] = 0                                   # empty boards: round 0 openings and anchorless later rounds
        inv = rng.integers(0, 1 << 21, size=(B, 4)).astype(np.uint32)
        inv[rng.random((B, 4)) < 0.15] = (1 << 21) - 1
        inv[rng.random((B, 4)) < 0.05] = 0
        rnd = np.where(rng.random(B) < 0.25, 0, rng.integers(1, 20, B)).astype(np.int32)
        player = rng.integers(0, 4, B).astype(np.int8)
        bb = BlokusBatch(B)
        bb.set_board(torch.from_numpy(board).cuda())
        bb.inv.copy_(torch.from_numpy(inv.view(np.int32)).cuda().view(bb.inv.dtype))
        bb.round.copy_(torch.from_numpy(rnd).cuda())
        st = O.BlokusState(B)
        st.set_board(board)
        st.inv[:] = inv
        st.round[:] = rnd
        c2, ids2 = O.blokus_valid(st, player=player, cap=cap, n_threads=16)
        assert c2.max() <= cap
        pl = torch.from_numpy(player).cuda()
        count, ids = bb.valid_list(cap, player=pl)
        assert np.array_equal(count.cpu().numpy(), c2), trial
        assert np.array_equal(ids.cpu().numpy(), ids2), trial
        assert np.array_equal(bb.valid(player=pl).cpu().numpy(), c2), trial
        rank = np.where(c2 > 0, rng.integers(0, np.maximum(c2, 1)), 0).astype(np.int32)
        act, _ = bb.select(torch.from_numpy(rank).cuda(), player=pl)
        want = np.where(c2 > 0, ids2[np.arange(B), rank], -1)
        assert np.array_equal(act.cpu().numpy(), want), trial
        probe = np.where(rng.random(B) < 0.5, np.maximum(want, 0), rng.integers(0, 336000, B)).astype(np.int32)
        ok = bb.is_valid(torch.from_numpy(probe).cuda(), player=pl).cpu().numpy()
        member = np.array([int(probe[e] in set(ids2[e, :c2[e]].tolist())) for e in range(B)], np.uint8)
        assert np.array_equal(ok, member), trial
        # fits = is_valid without the anchor / inventory conditions: every legal action fits; a legal action of a FULL
        # inventory on this board that is not legal for the real one still fits
        legal = np.where(c2 > 0, want, -1).astype(np.int32)
        fits = bb.fits(torch.from_numpy(np.maximum(legal, 0)).cuda(), pl).cpu().numpy()
        assert (fits[c2 > 0] == 1).all(), trial
