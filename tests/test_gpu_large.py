"""Batches whose arrays pass 2^32 bytes (the reference tests nothing of the kind: it steps one state per call): every byte
offset a kernel forms must be 64-bit there.  A batch that size cannot be checked against the CPU oracle in seconds, so the
test uses what the contract guarantees instead: a game's results depend on (seed, GLOBAL game id, step counter) alone, so
the first, the last and the games around byte 2^32 of the big batch must equal, bit for bit, a small batch created with the same
``first_env_id`` -- and the small batches are what every other test compares with the oracle.  A few seconds on a GPU box (27 GB of HBM)."""
import pytest

pytestmark = pytest.mark.gpu

K = 1088        # games per compared window (17 workgroups of 64)


def _windows(B, per_game_bytes):
    """Head, tail, and the games either side of byte offset 2^32 of the largest array."""
    mid = 2 ** 32 // per_game_bytes - K // 2
    assert K < mid < B - 2 * K
    return ((0, K), (mid, mid + K), (B - K, B))


@pytest.mark.parametrize("N,P,B", [(40, 4, 2_700_016), (19, 4, 11_900_016), (20, 4, 10_800_001)])
def test_tron_batches_beyond_4_gib(N, P, B):
    """Tron boards of 4.3-4.4 GB (observations of all players: 17 GB): rollouts on the LDS kernels and on the global kernel,
    `step_observe` with the sampled agent, `observe_all`, `observe`, `ranking`, masked `reset` -- head and tail of the batch
    against small batches with the same global ids."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    assert B * N * N > 2 ** 32
    seed = 0xB16B00 + N
    big = TronBatch(N, P, B, first_env_id=7)
    smalls = [(lo, hi, TronBatch(N, P, hi - lo, first_env_id=7 + lo)) for lo, hi in _windows(B, N * N)]

    def both(fn):
        fn(big, 0, B)
        for lo, hi, s in smalls:
            fn(s, lo, hi)

    def check(what):
        for lo, hi, s in smalls:
            for k in ("board", "tcount", "tstep", "n_episodes", "len_sum", "last_winners", "terminal", "winners"):
                assert torch.equal(getattr(big, k)[lo:hi], getattr(s, k)), (what, k, lo)
            for k in ("heads", "dirs", "deaths", "ret_sum", "win_count", "rewards"):
                assert torch.equal(getattr(big, k)[:, lo:hi], getattr(s, k)), (what, k, lo)
            assert torch.equal(big.results(copy=False)[lo:hi], s.results(copy=False)), (what, "rows", lo)
            assert torch.equal(big.results_packed(copy=False)[lo:hi], s.results_packed(copy=False)), (what, "packed rows", lo)

    both(lambda t, lo, hi: t.rollout(23, seed))                          # short launch
    check("rollout 23")
    both(lambda t, lo, hi: t.rollout(150, seed))                         # past 20 steps: bitboards on the 40x40 board
    check("rollout 150")
    both(lambda t, lo, hi: t.rollout(9, seed, kernel="global"))
    check("rollout 9 global")
    both(lambda t, lo, hi: t.rollout(11, seed, kernel="gquad"))
    check("rollout 11 gquad")
    # per-step API
    outs = {}
    def so(t, lo, hi):
        outs[lo, hi] = t.step_observe(None, seed, True)
    both(so)
    check("step_observe")
    ob = outs[0, B]
    for lo, hi, s in smalls:
        o = outs[lo, hi]
        assert torch.equal(ob["board"][:, lo:hi], o["board"]), ("step_observe boards", lo)
        for k in ("heads", "directions", "deaths"):
            assert torch.equal(ob[k][:, :, lo:hi], o[k]), ("step_observe", k, lo)
    del outs, ob
    torch.cuda.empty_cache()
    acts = {}
    def st(t, lo, hi):
        a = t.sample(seed)
        acts[lo, hi] = a
        t.step(a, auto_reset=True)
    both(st)
    check("sample + step")
    for lo, hi, s in smalls:
        assert torch.equal(acts[0, B][:, lo:hi], acts[lo, hi]), ("sample", lo)
    del acts
    oa = big.observe_all()
    for lo, hi, s in smalls:
        o = s.observe_all()
        assert torch.equal(oa["board"][:, lo:hi], o["board"]) and torch.equal(oa["heads"][:, :, lo:hi], o["heads"]), ("observe_all", lo)
    del oa
    torch.cuda.empty_cache()
    who = (torch.arange(B, device="cuda") % P).to(torch.int8)
    o1 = big.observe(who)
    rk = big.ranking()
    for lo, hi, s in smalls:
        o = s.observe(who[lo:hi].contiguous())
        assert torch.equal(o1["board"][lo:hi], o["board"]) and torch.equal(o1["deaths"][:, lo:hi], o["deaths"]), ("observe", lo)
        assert torch.equal(rk[:, lo:hi], s.ranking()), ("ranking", lo)
    del o1, rk
    mask = (torch.arange(B, device="cuda") % 3 == 0).to(torch.uint8)
    both(lambda t, lo, hi: t.reset(mask[lo:hi].contiguous()))
    both(lambda t, lo, hi: t.rollout(5, seed))
    check("masked reset + rollout 5")


def test_blokus_id_lists_beyond_4_gib():
    """Blokus: 600,000 games x 2,048 ids = 4.9 GB of `valid_list` output (and 240 MB of boards): rollout, list, select,
    step_observe at the head and the tail of the batch against small batches with the same global ids."""
    import torch
    from colosseumrl_amd.batched import BlokusBatch
    B, cap, seed = 600_000, 2048, 99
    assert B * cap * 4 > 2 ** 32
    big = BlokusBatch(B, first_env_id=5)
    smalls = [(lo, hi, BlokusBatch(hi - lo, first_env_id=5 + lo)) for lo, hi in _windows(B, cap * 4)]
    for t in [big] + [s for _, _, s in smalls]:
        t.rollout(21, seed)
    out = torch.empty((B, cap), dtype=torch.int32, device="cuda")
    n_big = big.valid_list(cap, None, out)[0]
    rank = (torch.arange(B, device="cuda", dtype=torch.int32) * 7) % 50
    sel_big, cnt_big = big.select(rank)
    board_big = big.board()
    for lo, hi, s in smalls:
        assert torch.equal(board_big[lo:hi], s.board()), ("board", lo)
        n_s, so = s.valid_list(cap)
        assert torch.equal(n_big[lo:hi], n_s) and torch.equal(cnt_big[lo:hi], n_s), ("list counts", lo)
        cols = torch.arange(cap, device="cuda")[None, :] < n_s[:, None].clamp(max=cap)
        assert torch.equal(torch.where(cols, out[lo:hi], 0), torch.where(cols, so, 0)), ("list ids", lo)
        assert torch.equal(sel_big[lo:hi], s.select(rank[lo:hi].contiguous())[0]), ("select", lo)
    del out
    ob = big.step_observe(None, seed, True)
    for lo, hi, s in smalls:
        o = s.step_observe(None, seed, True)
        for k in o:
            if torch.is_tensor(o[k]) and o[k].shape[0] == hi - lo:
                assert torch.equal(ob[k][lo:hi], o[k]), ("step_observe", k, lo)
        assert torch.equal(big.results(copy=False)[lo:hi], s.results(copy=False)), ("rows", lo)


def test_ttt_batches_beyond_4_gib():
    """TicTacToe 3x3x3, four players, 170,000,000 games: result rows 4.8 GB, boards of `board()` 4.6 GB, masks 2.7 GB --
    rollout, step_observe, sample + step, board / observe at the head, around byte 2^32 of the rows, and at the tail."""
    import torch
    from colosseumrl_amd.batched import TTTBatch
    dims, k, P, B, seed = (3, 3, 3), 3, 4, 170_000_000, 31337
    assert B * 27 > 2 ** 32 and B * (3 + P) * 4 > 2 ** 32
    big = TTTBatch(dims, k, P, B, first_env_id=3)
    smalls = [(lo, hi, TTTBatch(dims, k, P, hi - lo, first_env_id=3 + lo)) for lo, hi in _windows(B, (3 + P) * 4)]
    smalls += [(lo, hi, TTTBatch(dims, k, P, hi - lo, first_env_id=3 + lo)) for lo, hi in _windows(B, 27)[1:2]]

    def both(fn):
        fn(big, 0, B)
        for lo, hi, s in smalls:
            fn(s, lo, hi)

    def check(what):
        for lo, hi, s in smalls:
            for name in ("occ", "winner", "to_move", "tcount", "tstep", "n_episodes", "win_count", "draw_count", "len_sum",
                         "reward", "terminal", "winners"):
                a, b = getattr(big, name), getattr(s, name)
                a = a[lo:hi] if a.shape[0] == B else a[:, lo:hi]
                assert torch.equal(a, b), (what, name, lo)
            assert torch.equal(big.results(copy=False)[lo:hi], s.results(copy=False)), (what, "rows", lo)

    both(lambda t, lo, hi: t.rollout(37, seed))
    check("rollout 37")
    outs = {}
    def so(t, lo, hi):
        outs[lo, hi] = t.step_observe(None, seed, True)
    both(so)
    check("step_observe")
    for lo, hi, s in smalls:
        assert torch.equal(outs[0, B]["board"][lo:hi], outs[lo, hi]["board"]) and torch.equal(outs[0, B]["valid"][lo:hi], outs[lo, hi]["valid"]), lo
    del outs
    torch.cuda.empty_cache()
    both(lambda t, lo, hi: t.step(t.sample(seed), auto_reset=True))
    check("sample + step")
    who = (torch.arange(B, device="cuda") % P).to(torch.int8)
    bb, vm = big.board(who), big.valid_mask()
    for lo, hi, s in smalls:
        assert torch.equal(bb[lo:hi], s.board(who[lo:hi].contiguous())) and torch.equal(vm[lo:hi], s.valid_mask()), ("board", lo)


def test_blokus_states_beyond_4_gib():
    """Blokus with 13,500,000 games: 4.3 GB of row bitboards, 5.4 GB of observation boards -- a short rollout and
    step_observe at the head, around byte 2^32 of the bitboards, and at the tail."""
    import torch
    from colosseumrl_amd.batched import BlokusBatch
    B, seed = 13_500_000, 7
    assert B * 320 > 2 ** 32
    big = BlokusBatch(B, first_env_id=11)
    smalls = [(lo, hi, BlokusBatch(hi - lo, first_env_id=11 + lo)) for lo, hi in _windows(B, 320)]
    for t in [big] + [s for _, _, s in smalls]:
        t.rollout(3, seed)
    ob = big.step_observe(None, seed, True)
    for lo, hi, s in smalls:
        o = s.step_observe(None, seed, True)
        for name in ("board", "pieces", "score", "player", "n_valid", "reward", "terminal", "winners"):
            assert torch.equal(ob[name][lo:hi], o[name]), ("step_observe", name, lo)
        for name in ("occ", "inv", "score", "tcount", "n_episodes"):
            assert torch.equal(getattr(big, name)[lo:hi], getattr(s, name)), (name, lo)
        assert torch.equal(big.results(copy=False)[lo:hi], s.results(copy=False)), ("rows", lo)
