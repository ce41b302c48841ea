"""HIP Tron kernels (through the C ABI) vs golden vectors from the reference and vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from backends import HipTron, OracleTron
from replay import replay_tron, replay_tron_fused_reset

TRAJ = ["n20p4", "n40p4", "n20p2", "n21p3", "n20p6", "n7p5", "n20p4_noreset", "n9p8_noreset"]


class HipTronStaged(HipTron):
    step_kernel = "staged"


class HipTronBytes(HipTron):
    step_kernel = "bytes"


@pytest.mark.parametrize("backend", [HipTronBytes, HipTronStaged], ids=["bytes", "staged"])
@pytest.mark.parametrize("name", TRAJ)
def test_traj_golden(golden, name, backend):
    assert replay_tron(golden("tron_traj_" + name), backend) > 0


@pytest.mark.parametrize("backend", [HipTronBytes, HipTronStaged], ids=["bytes", "staged"])
@pytest.mark.parametrize("name", ["n20p4", "n40p4", "n7p5", "n21p3"])
def test_traj_golden_fused_reset(golden, name, backend):
    replay_tron_fused_reset(golden("tron_traj_" + name), backend)


@pytest.mark.parametrize("backend", [HipTronBytes, HipTronStaged], ids=["bytes", "staged"])
def test_edge_cases_golden(golden, backend):
    g = golden("tron_edge")
    N, P = int(g["N"]), int(g["P"])
    E = len(g["names"])
    be = backend(N, P, E, [0, 1, 2], [0, 0, 0])
    be.set_state(g["pre_board"], g["pre_heads"].T, g["pre_dirs"].T, g["pre_deaths"].T)
    rew, term, win = be.step(np.ascontiguousarray(g["actions"].T))
    s = be.state()
    assert np.array_equal(s["board"], g["post_board"])
    assert np.array_equal(s["heads"], g["post_heads"].T) and np.array_equal(s["dirs"], g["post_dirs"].T)
    assert np.array_equal(s["deaths"], g["post_deaths"].T) and np.array_equal(rew, g["rewards"].T)
    assert np.array_equal(term, g["terminal"]) and np.array_equal(win, g["winners"])


@pytest.mark.parametrize("name", ["n20p4", "n9p6", "wrap_n20p4", "wrap_n9p6"])
def test_observe_golden(golden, name):
    # wrap_*: observer ids outside 0..P-1 (negative, P, beyond P) as the reference answers them
    g = golden("tron_observe_" + name)
    N, P = int(g["N"]), int(g["P"])
    E = len(g["player"])
    be = HipTron(N, P, E, list(range(P)), [0] * P)
    be.set_state(g["board"], g["heads"].T, g["dirs"].T, g["deaths"].T)
    ob, oh, od, ok = be.observe(g["player"])
    assert np.array_equal(ob, g["obs_board"]) and np.array_equal(oh, g["obs_heads"].T)
    assert np.array_equal(od, g["obs_dirs"].T) and np.array_equal(ok, g["obs_deaths"].T)


# ---- the Cython module's own signature: int64 arrays in the reference's layout, in place (ABI 110) ----------------
class Inplace64:
    """B games in the reference's layout on the device: board int64 [B][N*N], heads / directions / deaths [B][P]."""

    def __init__(self, N, P, B):
        import ctypes as C
        import torch
        from colosseumrl_amd import _native
        self.C, self.torch, self.lib, self.N, self.P, self.B = C, torch, _native.require_gpu(), N, P, B
        self.h = C.c_void_p()
        sh, sd = (C.c_int16 * P)(*range(P)), (C.c_int8 * P)(*([0] * P))
        _native.check(self.lib.crl_tron_create(N, P, sh, sd, C.byref(self.h)), "crl_tron_create")
        z = lambda *shape, dt=torch.int64: torch.zeros(shape, dtype=dt, device="cuda")     # noqa: E731
        self.board, self.heads, self.dirs, self.deaths = z(B, N * N), z(B, P), z(B, P), z(B, P)
        self.rewards, self.terminal, self.winners = z(B, P), z(B, dt=torch.uint8), z(B, dt=torch.uint8)
        self.ob, self.oh, self.od, self.ok = z(B, P, N * N), z(B, P, P), z(B, P, P), z(B, P, P)

    def __del__(self):
        self.lib.crl_destroy(self.h)

    def set_state(self, board, heads, dirs, deaths):       # [B][NN], [B][P] numpy
        for dst, src in ((self.board, board), (self.heads, heads), (self.dirs, dirs), (self.deaths, deaths)):
            dst.copy_(self.torch.from_numpy(np.ascontiguousarray(src).astype(np.int64)))

    def step(self, actions, obs):
        from colosseumrl_amd import _native
        p = lambda t: self.C.c_void_p(t.data_ptr())                                        # noqa: E731
        a = self.torch.from_numpy(np.ascontiguousarray(actions).astype(np.int64)).cuda()
        o = [p(self.ob), p(self.oh), p(self.od), p(self.ok)] if obs else [None] * 4
        _native.check(self.lib.crl_tron_next_state_inplace64(self.h, self.B, p(self.board), p(self.heads), p(self.dirs), p(self.deaths),
                                                             p(a), p(self.rewards), p(self.terminal), p(self.winners), *o, None),
                      "crl_tron_next_state_inplace64")
        self.torch.cuda.synchronize()

    def relative(self, board, num_players, player):
        from colosseumrl_amd import _native
        b = self.torch.from_numpy(np.ascontiguousarray(board).astype(np.int64)).cuda()
        pl = self.torch.from_numpy(np.ascontiguousarray(player).astype(np.int64)).cuda()
        _native.check(self.lib.crl_tron_relative_player_inplace64(self.h, b.shape[0], self.C.c_void_p(b.data_ptr()), num_players,
                                                                  self.C.c_void_p(pl.data_ptr()), None), "crl_tron_relative_player_inplace64")
        return b.cpu().numpy()

    def np(self, *names):
        return [getattr(self, n).cpu().numpy() for n in names]


@pytest.mark.parametrize("obs", [False, True])
def test_next_state_inplace64_edge_cases_golden(golden, obs):
    """The reference's order-dependence cases (SURVEY T2-order i-iv, stale heads, walls) through the Cython signature."""
    g = golden("tron_edge")
    N, P, E = int(g["N"]), int(g["P"]), len(g["names"])
    be = Inplace64(N, P, E)
    be.set_state(g["pre_board"], g["pre_heads"], g["pre_dirs"], g["pre_deaths"])
    be.step(g["actions"], obs)
    board, heads, dirs, deaths, rew, term, win = be.np("board", "heads", "dirs", "deaths", "rewards", "terminal", "winners")
    assert np.array_equal(board, g["post_board"]) and np.array_equal(heads, g["post_heads"]) and np.array_equal(dirs, g["post_dirs"])
    assert np.array_equal(deaths, g["post_deaths"]) and np.array_equal(rew, g["rewards"])
    assert np.array_equal(term, g["terminal"]) and np.array_equal(win, g["winners"])


def test_cytrongrid_stand_in_module_golden(golden):
    """colosseumrl_amd.envs.tron.CyTronGrid: the reference's native module under its own two signatures, plain numpy arrays in
    ordinary memory, mutated in place -- against calls of the reference's functions (order-dependence cases, wild actions /
    directions, relabelling for observer ids of any size)."""
    from colosseumrl_amd.envs.tron import CyTronGrid
    for name, keys in (("tron_edge", ("pre_board", "pre_heads", "pre_dirs", "pre_deaths", "actions")),
                       ("tron_wild64", ("pre_board", "pre_heads", "pre_dirs", "pre_deaths", "actions"))):
        g = golden(name)
        N = int(g["N"])
        for i in range(len(g["actions"])):
            board, heads, dirs, deaths, acts = (np.array(g[k][i], np.int64) for k in keys)
            board = board.reshape(N, N)
            assert CyTronGrid.next_state_inplace(board, heads, dirs, deaths, acts) is None
            assert np.array_equal(board.reshape(-1), g["post_board"][i]) and np.array_equal(heads, g["post_heads"][i]), (name, i)
            assert np.array_equal(dirs, g["post_dirs"][i]) and np.array_equal(deaths, g["post_deaths"][i]), (name, i)
    for name in ("n20p4", "wrap_n20p4", "wrap_n9p6"):
        g = golden("tron_observe_" + name)
        N, P = int(g["N"]), int(g["P"])
        for i in range(0, len(g["player"]), 3):
            board = np.array(g["board"][i], np.int64).reshape(N, N)
            CyTronGrid.relative_player_inplace(board, P, int(g["player"][i]) + 1)
            assert np.array_equal(board.reshape(-1), g["obs_board"][i]), (name, i)


@pytest.mark.parametrize("obs", [False, True])
def test_next_state_inplace64_wild_actions_and_directions_golden(golden, obs):
    """The Cython function called directly with actions / stored directions outside their usual ranges (the reference's C
    remainder gives negative directions: the player runs into its own cell, the negative direction is stored): 240 calls of
    the reference's CyTronGrid.next_state_inplace, chains on the same arrays included (tests/golden/tron_wild64.npz)."""
    g = golden("tron_wild64")
    N, P, E = int(g["N"]), int(g["P"]), len(g["actions"])
    be = Inplace64(N, P, E)
    be.set_state(g["pre_board"], g["pre_heads"], g["pre_dirs"], g["pre_deaths"])
    be.step(g["actions"], obs)
    board, heads, dirs, deaths = be.np("board", "heads", "dirs", "deaths")
    assert np.array_equal(board, g["post_board"]) and np.array_equal(heads, g["post_heads"])
    assert np.array_equal(dirs, g["post_dirs"]) and np.array_equal(deaths, g["post_deaths"])
    assert (dirs < 0).sum() > 100                                 # the negative directions are really there


@pytest.mark.parametrize("N,P,B,T", [(20, 4, 300, 40), (40, 4, 64, 60), (19, 3, 77, 40), (9, 8, 130, 30), (5, 2, 100, 20)])
def test_next_state_inplace64_vs_oracle_random(N, P, B, T):
    """Seeded random play: the int64 in-place entry (with and without the fused observations) against the oracle's step
    and observe on the same states; terminal games are reset on both sides."""
    rng = np.random.default_rng(N * 77 + P)
    sh, sd = O.tron_start_positions(N, P)
    orc = OracleTron(N, P, B, sh, sd)
    be = Inplace64(N, P, B)
    for t in range(T):
        s = orc.state()
        be.set_state(s["board"], s["heads"].T, s["dirs"].T, s["deaths"].T)
        a = rng.integers(-1, 2, size=(P, B)).astype(np.int8)
        obs = (t % 2) == 0
        be.step(a.T, obs)
        r, term, w = orc.step(a)
        s = orc.state()
        board, heads, dirs, deaths, rew, tm, wn = be.np("board", "heads", "dirs", "deaths", "rewards", "terminal", "winners")
        assert np.array_equal(board, s["board"]) and np.array_equal(heads, s["heads"].T) and np.array_equal(dirs, s["dirs"].T), t
        assert np.array_equal(deaths, s["deaths"].T) and np.array_equal(rew, r.T) and np.array_equal(tm, term) and np.array_equal(wn, w), t
        if obs:
            ob, oh, od, ok = be.np("ob", "oh", "od", "ok")
            for p in range(P):
                eb, eh, ed, ek = orc.observe(np.full(B, p, np.int8))
                assert np.array_equal(ob[:, p], eb) and np.array_equal(oh[:, p], eh.T), (t, p)
                assert np.array_equal(od[:, p], ed.T) and np.array_equal(ok[:, p], ek.T), (t, p)
        orc.reset(term.astype(bool))


@pytest.mark.parametrize("N,P", [(20, 4), (19, 3), (9, 8), (40, 4)])
def test_next_state_inplace64_host_single_state_call(N, P):
    """The one-call single-state form on crl_host_alloc memory (player vectors by value, completion published by the kernel)
    against the two-call form (launch + crl_stream_wait_mapped) and the oracle, with and without fused observations."""
    from colosseumrl_amd.single import SingleTron
    rng = np.random.default_rng(N + P)
    sh, sd = O.tron_start_positions(N, P)
    orc = OracleTron(N, P, 1, sh, sd)
    one, two = SingleTron(N, P, sh, sd), SingleTron(N, P, sh, sd)
    assert one._unified                                          # ROCm's unified addressing: mapped memory has one address
    two._unified = False
    for t in range(300):
        s = orc.state()
        state = (s["board"].reshape(N, N).astype(np.int64), s["heads"][:, 0].astype(np.int64), s["dirs"][:, 0].astype(np.int64),
                 s["deaths"][:, 0].astype(np.int64))
        a = rng.integers(-1, 2, size=(P, 1)).astype(np.int8)
        obs = (t % 3) != 0
        for st in (one, two):
            st.s64["obs"][:] = -7
            st.next_state64(*state, a[:, 0].astype(np.int64), obs)
        r, term, w = orc.step(a)
        s = orc.state()
        for st in (one, two):
            v = st.s64
            assert np.array_equal(v["board"].reshape(-1), s["board"][0]) and np.array_equal(v["heads"], s["heads"][:, 0]), t
            assert np.array_equal(v["dirs"], s["dirs"][:, 0]) and np.array_equal(v["deaths"], s["deaths"][:, 0]), t
            assert np.array_equal(v["rewards"], r[:, 0]) and int(v["terminal"][0]) == int(term[0]), t
        assert np.array_equal(one.s64["obs"], two.s64["obs"]) and (obs == bool((one.s64["obs"] != -7).any())), t
        if obs:
            NN = N * N
            for p in range(P):
                eb, eh, ed, ek = orc.observe(np.full(1, p, np.int8))
                assert np.array_equal(one.s64["obs"][p * NN:(p + 1) * NN], eb[0]), (t, p)
                o = P * NN + p * P
                assert np.array_equal(one.s64["obs"][o:o + P], eh[:, 0]) and np.array_equal(one.s64["obs"][o + P * P:o + P * P + P], ed[:, 0]), (t, p)
                assert np.array_equal(one.s64["obs"][o + 2 * P * P:o + 2 * P * P + P], ek[:, 0]), (t, p)
        orc.reset(term.astype(bool))


@pytest.mark.parametrize("name", ["n20p4", "n9p6", "wrap_n20p4", "wrap_n9p6"])
def test_relative_player_inplace64_golden(golden, name):
    """CyTronGrid.relative_player_inplace with its own arguments (player = observer id + 1, any integer) against the
    reference's boards, incl. the observer ids outside 0..P-1."""
    g = golden("tron_observe_" + name)
    N, P = int(g["N"]), int(g["P"])
    be = Inplace64(N, P, 1)
    got = be.relative(g["board"], P, g["player"].astype(np.int64) + 1)
    assert np.array_equal(got, g["obs_board"])
    wild = np.array([-(1 << 40) + 3, (1 << 40) + 1, -7, 0][: len(g["player"])], np.int64)   # far outside int8: C remainder on int64
    b = g["board"][: len(wild)].astype(np.int64)
    got = be.relative(b, P, wild)
    want = np.where(b > 0, np.fmod(b - wild[:, None] + P, P) + 1, b)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kernel", ["bytes", "staged"])
@pytest.mark.parametrize("N,P,B,T", [(20, 4, 4096 + 37, 40), (40, 4, 1000, 60), (19, 3, 777, 40), (9, 8, 513, 30), (5, 2, 300, 20), (12, 6, 63, 40),
                                     (4, 2, 200, 12), (100, 4, 70, 30)])
def test_step_vs_oracle_random(N, P, B, T, kernel):
    """Seeded random actions, ragged batch sizes (not a multiple of the wave or of a workgroup's 64 / 16 games), odd boards
    (byte reset path), on both interchangeable kernels of crl_tron_step (byte probes / boards staged through LDS; boards that
    are not whole 16-byte chunks or do not fit LDS take the byte kernel under either pin)."""
    rng = np.random.default_rng(N * 1000 + P)
    sh, sd = O.tron_start_positions(N, P)
    hip, orc = HipTron(N, P, B, sh, sd), OracleTron(N, P, B, sh, sd)
    hip.step_kernel = kernel
    for t in range(T):
        a = rng.integers(-1, 2, size=(P, B)).astype(np.int8)
        auto = (t % 3) != 2
        r1, t1, w1 = hip.step(a, auto_reset=auto)
        r2, t2, w2 = orc.step(a, auto_reset=auto)
        assert np.array_equal(r1, r2) and np.array_equal(t1, t2) and np.array_equal(w1, w2)
        s1, s2 = hip.state(), orc.state()
        for k in s1:
            assert np.array_equal(s1[k], s2[k]), (k, t)


def _rollout_pair(N, P, B, chunks, seed, first, kernel="auto"):
    sh, sd = O.tron_start_positions(N, P)
    hip = HipTron(N, P, B, sh, sd)
    hip.tb.first_env_id = first
    ost = O.TronState(N, P, B)
    O.tron_reset(ost, sh, sd)
    for T in chunks:
        hip.tb.rollout(T, seed, kernel=kernel)
        O.tron_rollout(ost, seed, first, T, sh, sd, n_threads=8)
    tb = hip.tb
    got = dict(board=tb.board, heads=tb.heads, dirs=tb.dirs, deaths=tb.deaths, tcount=tb.tcount, tstep=tb.tstep,
               n_episodes=tb.n_episodes, win_count=tb.win_count, len_sum=tb.len_sum, ret_sum=tb.ret_sum,
               last_winners=tb.last_winners, last_len=tb.last_len)
    for k, v in got.items():
        want = getattr(ost, k)
        have = v.cpu().numpy().view(want.dtype)
        assert np.array_equal(have, want), k
    import torch
    assert torch.equal(tb.results(), tb.results_from_columns())      # the row the kernel packs == the column statistics
    assert torch.equal(tb.results_packed(), tb.results_packed_from_columns())   # and its 16-bit encoding (low 16 bits)
    assert tb.packed_rows_exact() == (sum(chunks) <= 3276)
    return ost


@pytest.mark.parametrize("kernel", ["quad", "pair", "qbits", "bits", "bytes", "global", "gquad", "auto"])
@pytest.mark.parametrize("N,P,B,chunks", [(20, 4, 8192 + 5, (64, 1, 31)), (20, 3, 1000 + 3, (40, 9)), (12, 2, 321, (60,)), (16, 4, 64 * 5 + 1, (25, 25)), (40, 4, 2048, (100,)), (19, 5, 1000, (50, 3, 47)),
                                           (9, 8, 640, (40,)), (25, 2, 300, (7, 9, 30)), (24, 6, 129, (33,)),
                                           (8, 4, 320, (700,)), (7, 8, 200, (400, 100)), (11, 3, 100, (900,)),
                                           (20, 4, 1000, (300, 2, 260)), (40, 4, 700, (1, 1, 290)), (37, 7, 130, (280,)),
                                           (4, 2, 200, (50,)), (4, 4, 70, (20,)), (4, 3, 130, (30,)), (5, 4, 100, (40,))])
def test_rollout_vs_oracle(N, P, B, chunks, kernel):
    """Fused random-agent rollout == oracle rollout, bit for bit, for the seven kernels behind crl_tron_rollout (lane per
    player on byte slabs "quad" / two lanes per game "pair" / on bitboards with a replay kernel "qbits" / in global memory
    "gquad" -- where they do not apply, P > 4 (P > 2) or for quad boards above 20x20, the flag falls through to the
    library's choice -- lane-per-game LDS bitboard with replay epilogue, LDS byte slabs, global memory) and the library's
    own choice: ragged batches, odd
    boards (byte copy path), split launches (state and RNG position carry over; a 1-step launch makes the bitboard
    kernel replay from the incoming state), and launches long enough to wrap the byte kernel's episode tags."""
    ost = _rollout_pair(N, P, B, chunks, seed=0xC0FFEE12345, first=123456, kernel=kernel)
    assert ost.n_episodes.sum() > B


@pytest.mark.parametrize("kernel", ["quad", "pair", "qbits", "bits", "bytes", "global", "gquad"])
def test_rollout_after_scripted_steps(kernel):
    """A rollout continues from whatever state the step API left (mid-episode, some players dead, histories that no
    RNG stream produced): the bitboard kernel must resume from the incoming board, not from the episode start."""
    N, P, B, seed, first = 20, 4, 2000, 99, 5
    sh, sd = O.tron_start_positions(N, P)
    hip, ora = HipTron(N, P, B, sh, sd), OracleTron(N, P, B, sh, sd)
    hip.tb.first_env_id = first
    rng = np.random.default_rng(3)
    for t in range(6):
        act = rng.choice(np.array([0, 1, -1], dtype=np.int8), size=(P, B))
        r1, t1, w1 = hip.step(act)
        r2, t2, w2 = ora.step(act)
        assert np.array_equal(t1, t2)
    ost = ora.st
    ost.tcount[:] = 0
    ost.tstep[:] = 0
    for T in (3, 40):
        hip.tb.rollout(T, seed, kernel=kernel)
        O.tron_rollout(ost, seed, first, T, sh, sd, n_threads=8)
        for k in ("board", "heads", "dirs", "deaths", "tstep", "n_episodes", "win_count", "ret_sum", "len_sum"):
            want = getattr(ost, k)
            assert np.array_equal(getattr(hip.tb, k).cpu().numpy().view(want.dtype), want), (k, T)


@pytest.mark.parametrize("kernel", ["quad", "pair", "qbits", "bits", "bytes", "global", "gquad"])
@pytest.mark.parametrize("N,P,B", [(20, 4, 1000), (33, 3, 300), (12, 7, 257)])
def test_rollout_with_step_counters_that_differ_inside_a_wave(N, P, B, kernel):
    """Games of one batch (and of one wave) may stand at different step counters -- states assembled from several sources.
    The kernels then leave their wave-uniform fast paths (scalar action countdown of the lane-per-player kernels, 4-step
    trips of the bitboard one): per-game counters from 0..1000, odd and even, against the oracle."""
    seed, first = 77, 9
    sh, sd = O.tron_start_positions(N, P)
    hip = HipTron(N, P, B, sh, sd)
    hip.tb.first_env_id = first
    ost = O.TronState(N, P, B)
    O.tron_reset(ost, sh, sd)
    rng = np.random.default_rng(5)
    tc = rng.integers(0, 1000, size=B).astype(ost.tcount.dtype)
    tc[: B // 3] = 40                                    # (a stretch of equal counters: some waves are uniform, some are not)
    ost.tcount[:] = tc
    import torch
    hip.tb.tcount.copy_(torch.from_numpy(tc.view(np.int32)))        # (oracle: uint32, device tensor: int32)
    for T in (37, 300):
        hip.tb.rollout(T, seed, kernel=kernel)
        O.tron_rollout(ost, seed, first, T, sh, sd, n_threads=8)
    for k in ("board", "heads", "dirs", "deaths", "tcount", "tstep", "n_episodes", "win_count", "ret_sum", "len_sum", "last_winners", "last_len"):
        want = getattr(ost, k)
        assert np.array_equal(getattr(hip.tb, k).cpu().numpy().view(want.dtype), want), k


@pytest.mark.parametrize("N,T", [(20, 128), (40, 128), (40, 300)])
def test_rollout_full_size_properties(N, T):
    """BASELINE config 2 (N=20, LDS byte slabs, a lane per player) and config 5 per-GPU shard (N=40: bitboards + replay,
    a lane per player), B=65536, P=4: size-independent properties of the fused rollout.
    sum(len_sum)+sum(tstep) == B*T; wins <= episodes; board consistent with heads; shard invariance."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    B, seed = 65536, 7
    tb = TronBatch(N, 4, B)
    tb.rollout(T, seed)
    n_ep = tb.n_episodes.cpu().numpy().astype(np.int64)
    assert int(tb.len_sum.sum().item()) + int(tb.tstep.sum().item()) == B * T
    assert (tb.win_count.cpu().numpy().sum(axis=0) <= n_ep).all()
    assert abs(tb.len_sum.sum().item() / max(1, n_ep.sum()) - 9.4) < 1.0          # mean episode 9.3 / 9.5
    board = tb.board.cpu().numpy()
    heads = tb.heads.cpu().numpy().astype(np.int64)
    for p in range(4):
        assert (board[np.arange(B), heads[p]] == p + 1).all()                      # every head sits on its own trail
    assert (np.count_nonzero(board, axis=1) <= 4 * (tb.tstep.cpu().numpy() + 1)).all()
    # shard invariance: envs [B/2, B) computed as their own shard give the same answer
    half = TronBatch(N, 4, B // 2, first_env_id=B // 2)
    half.rollout(T, seed)
    assert torch.equal(half.board, tb.board[B // 2:]) and torch.equal(half.ret_sum, tb.ret_sum[:, B // 2:])
    assert torch.equal(half.n_episodes, tb.n_episodes[B // 2:])


@pytest.mark.parametrize("N,chunks,kernel", [(20, (32,), "auto"), (20, (20, 12), "quad"), (20, (20, 12), "bits"), (20, (20, 12), "bytes"), (20, (20, 12), "qbits"), (40, (32,), "auto"), (40, (32,), "bytes"), (40, (300,), "auto"), (40, (300,), "bits"),
                                                 # the reference's default 19 x 19 (rows at any alignment, flat copy back) and 15 x 15
                                                 (19, (20, 17), "auto"), (15, (20, 9), "quad")])
def test_rollout_full_size_vs_oracle(N, chunks, kernel):
    """BASELINE config 2 / the config-5 shard at FULL size (B = 65,536, P = 4) against the oracle itself, not only through
    invariants: every state array and every statistic, bit for bit (the oracle needs ~10-100 ms per launch on 8 threads).
    T = 300 on 40x40 is the lane-per-player bitboard kernel with its replay epilogue ('bits': the lane-per-game one); 20x20
    'bits' / 'qbits' pin the bitboard kernels there."""
    ost = _rollout_pair(N, 4, 65536, chunks, seed=20261004, first=3 * 65536, kernel=kernel)
    assert ost.n_episodes.sum() > 65536


@pytest.mark.parametrize("B,chunks", [(65536 + 64 + 13, (20, 7)), (65536 + 1, (1, 64, 5)), (2 * 65536, (20,))])
def test_rollout_beyond_full_size_ragged_vs_oracle(B, chunks):
    """Short launches of 20x20 boards on more than a full chip's worth of games (B > 65,536): ragged batches whose last
    workgroup is partly filled or has waves wholly beyond the batch, one-step launches -- against the oracle.  (Written for
    a variant of the lane-per-player kernel that played two sets of 64 games per workgroup on short launches; the variant
    was correct and 3 us slower, DESIGN 4.1, and is gone; the cases stay.)"""
    ost = _rollout_pair(20, 4, B, chunks, seed=77, first=11, kernel="quad")
    assert ost.n_episodes.sum() > 0


def test_dropin_env_golden(golden):
    """The BaseEnvironment drop-in (strings in, numpy tuples out) replays a golden game exactly."""
    from colosseumrl_amd import get_environment
    g = golden("tron_traj_n20p4")
    env = get_environment("tron")("20;4")
    names = {0: "forward", 1: "right", -1: "left"}
    for e in range(3):
        state, players = env.new_state()
        assert state[0].dtype == np.int64 and state[1].tolist() == g["start_heads"].tolist()
        for t in range(int(g["T"])):
            acts = [names[int(a)] for a in g["actions"][t, :, e]]
            state, players, rewards, terminal, winners = env.next_state(state, list(range(4)), acts)
            assert state[1].tolist() == g["heads"][t, :, e].tolist()
            assert state[3].tolist() == g["deaths"][t, :, e].tolist()
            assert rewards.tolist() == g["rewards"][t, :, e].tolist() and rewards.dtype == np.int64
            assert bool(terminal) == bool(g["terminal"][t, e])
            assert players.tolist() == [p for p in range(4) if g["deaths"][t, p, e] == 0]
            if terminal:
                assert sum(1 << int(w) for w in winners) == g["winners"][t, e]
                break
            assert winners is None
    assert env.valid_actions(state, 0) == ["forward", "right", "left"] and env.is_valid_action(state, 0, "x")
    with pytest.raises(KeyError):
        env.next_state(state, [0], ["up"])
    # stale-move replay (reference :118,297-298): an omitted live player repeats its previous move
    state, _ = env.new_state()
    s1, *_ = env.next_state(state, [0, 1, 2, 3], ["left", "forward", "forward", "forward"])
    s2, *_ = env.next_state(s1, [1, 2, 3], ["forward", "forward", "forward"])
    s2b, *_ = env.next_state(s1, [0, 1, 2, 3], ["left", "forward", "forward", "forward"])
    assert all(np.array_equal(a, b) for a, b in zip(s2, s2b))
    obs = env.state_to_observation(s2, 2)
    assert obs["board"].shape == (20, 20) and obs["heads"][0] == s2[1][2]
    # observer ids outside 0..P-1: the reference's answers (numpy modulo for the vectors, C remainder for the board); ids that
    # do not fit int8 fold on the host; the state next_state just returned is served from the fused launch up to id == P
    gw = golden("tron_observe_wrap_n20p4")
    for i in range(len(gw["player"])):
        st = (gw["board"][i].reshape(20, 20).astype(np.int64), gw["heads"][i].astype(np.int64), gw["dirs"][i].astype(np.int64),
              gw["deaths"][i].astype(np.int64))
        for pid in (int(gw["player"][i]), int(gw["player"][i]) + (400 if gw["player"][i] >= 8 else -400 if gw["player"][i] < 0 else 0)):
            ob = env.state_to_observation(st, pid)
            assert np.array_equal(ob["board"].reshape(-1), gw["obs_board"][i]) and np.array_equal(ob["heads"], gw["obs_heads"][i]), pid
            assert np.array_equal(ob["directions"], gw["obs_dirs"][i]) and np.array_equal(ob["deaths"], gw["obs_deaths"][i]), pid
    for pid, same_as in ((-1, 3), (4, 0), (-6, 2)):
        a, b = env.state_to_observation(s2, pid), env.state_to_observation(s2, same_as)
        assert all(np.array_equal(a[k], b[k]) for k in a)
    g2 = golden("tron_ranking_n20p4")
    for i in range(12):
        st = (g2["board"][i].reshape(20, 20).astype(np.int64), np.zeros(4, np.int64), np.zeros(4, np.int64), g2["deaths"][i].astype(np.int64))
        rk = env.compute_ranking(st, [0, 1, 2, 3], [])
        assert [rk[p] for p in range(4)] == g2["rank"][i].tolist()


def test_philox_device_kat():
    import ctypes as C
    import torch
    from colosseumrl_amd import _native
    lib = _native.require_gpu()
    ctr = torch.tensor([[0, 0, 0, 0], [-1, -1, -1, -1], [0x243f6a88, 0x85a308d3 - 2**32, 0x13198a2e, 0x03707344]],
                       dtype=torch.int32, device="cuda")
    out = torch.zeros_like(ctr)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _native.check(lib.crl_philox4x32(C.c_void_p(ctr.data_ptr()), 0, 0, C.c_void_p(out.data_ptr()), 1, stream))
    assert out[0].cpu().numpy().view(np.uint32).tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    _native.check(lib.crl_philox4x32(C.c_void_p(ctr[1:].data_ptr()), 0xffffffff, 0xffffffff, C.c_void_p(out[1:].data_ptr()), 1, stream))
    assert out[1].cpu().numpy().view(np.uint32).tolist() == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    _native.check(lib.crl_philox4x32(C.c_void_p(ctr[2:].data_ptr()), 0xa4093822, 0x299f31d0, C.c_void_p(out[2:].data_ptr()), 1, stream))
    assert out[2].cpu().numpy().view(np.uint32).tolist() == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_bad_arguments_fail_loudly():
    import torch
    from colosseumrl_amd import _native
    from colosseumrl_amd.batched import TronBatch
    tb = TronBatch(8, 2, 16)
    with pytest.raises(ValueError):
        tb.step(torch.zeros((2, 15), dtype=torch.int8, device="cuda"))
    with pytest.raises(_native.NativeError):
        TronBatch(8, 9, 4)
    with pytest.raises(_native.NativeError):
        TronBatch(200, 2, 4)


@pytest.mark.parametrize("name", ["n20p4", "n9p6", "n12p2", "n7p8"])
def test_ranking_golden(golden, name):
    g = golden("tron_ranking_" + name)
    N, P = int(g["N"]), int(g["P"])
    E = len(g["rank"])
    be = HipTron(N, P, E, list(range(P)), [0] * P)
    z = np.zeros((P, E))
    be.set_state(g["board"], z.astype(np.int16), z.astype(np.int8), np.ascontiguousarray(g["deaths"].T))
    assert np.array_equal(be.tb.ranking().cpu().numpy().T, g["rank"])


@pytest.mark.parametrize("N,P,B", [(19, 4, 4096), (19, 4, 4099), (15, 3, 1008), (39, 4, 160), (5, 2, 96), (21, 7, 333), (4, 2, 64), (64, 4, 48),
                                   (19, 4, 1), (17, 4, 7), (5, 2, 1), (23, 8, 65)])
def test_streaming_calls_on_boards_that_are_not_whole_chunks(N, P, B):
    """crl_tron_reset (all games and masked), crl_tron_observe (a random observer per game, ids outside 0..P-1 among them) and
    crl_tron_ranking on boards whose N * N is not a multiple of 16 -- the reference's default 19 x 19 among them --, which take
    the batch as one byte stream in 16-byte chunks that straddle games; batches that are and are not whole chunks."""
    import torch
    rng = np.random.default_rng(N * 31 + B)
    sh, sd = O.tron_start_positions(N, P)
    hip, orc = HipTron(N, P, B, sh, sd), OracleTron(N, P, B, sh, sd)
    for rnd in range(3):
        for t in range(int(rng.integers(3, 12))):
            a = rng.integers(-1, 2, size=(P, B)).astype(np.int8)
            hip.step(a), orc.step(a)
        s1, s2 = hip.state(), orc.state()
        assert np.array_equal(s1["board"], s2["board"])
        want = O.tron_ranking(N, P, s2["board"], s2["deaths"])
        assert np.array_equal(hip.tb.ranking().cpu().numpy(), want), rnd
        pl = rng.integers(-3, P + 4, size=B).astype(np.int8)
        ob, oh, od, ok = hip.observe(pl)
        eb, eh, ed, ek = orc.observe(pl)
        assert np.array_equal(ob, eb) and np.array_equal(oh, eh) and np.array_equal(od, ed) and np.array_equal(ok, ek), rnd
        mask = (rng.random(B) < (0.5 if rnd else 0.08)).astype(np.uint8)       # sparse and dense masks: chunk halves stored apart
        hip.reset(mask), orc.reset(mask)
        s1, s2 = hip.state(), orc.state()
        for k in s1:
            assert np.array_equal(s1[k], s2[k]), (k, rnd)
    hip.reset(), orc.reset()
    assert np.array_equal(hip.state()["board"], orc.state()["board"])


def test_ranking_full_size_vs_oracle():
    from colosseumrl_amd.batched import TronBatch
    tb = TronBatch(20, 4, 65536)
    tb.rollout(40, 3)
    want = O.tron_ranking(20, 4, tb.board.cpu().numpy(), tb.deaths.cpu().numpy())
    assert np.array_equal(tb.ranking().cpu().numpy(), want)


def test_largest_board_and_single_game():
    """N=181 is the largest board int16 heads can address; B=1 is the drop-in case."""
    N, P = 181, 2
    sh, sd = O.tron_start_positions(N, P)
    for B in (1, 3):
        hip, orc = HipTron(N, P, B, sh, sd), OracleTron(N, P, B, sh, sd)
        rng = np.random.default_rng(B)
        for t in range(12):
            a = rng.integers(-1, 2, size=(P, B)).astype(np.int8)
            r1, t1, w1 = hip.step(a)
            r2, t2, w2 = orc.step(a)
            assert np.array_equal(r1, r2) and np.array_equal(t1, t2) and np.array_equal(w1, w2)
        s1, s2 = hip.state(), orc.state()
        assert all(np.array_equal(s1[k], s2[k]) for k in s1)
        hip.tb.rollout(300, 5)
        ost = O.TronState(N, P, B)
        ost.board[:], ost.heads[:], ost.dirs[:], ost.deaths[:] = s2["board"], s2["heads"], s2["dirs"], s2["deaths"]
        O.tron_rollout(ost, 5, 0, 300, sh, sd)
        assert np.array_equal(hip.tb.board.cpu().numpy(), ost.board) and np.array_equal(hip.tb.heads.cpu().numpy(), ost.heads)


@pytest.mark.parametrize("N,P,B", [(20, 4, 5000), (9, 6, 777), (13, 8, 300), (40, 4, 1024), (19, 4, 4096), (19, 4, 4099), (15, 3, 1008), (39, 8, 64),
                                   (5, 2, 1), (19, 4, 1), (7, 3, 3), (21, 7, 1001), (5, 4, 2), (19, 4, 65553)])
def test_observe_all_matches_per_player_observe(N, P, B):
    """The fused all-observers pass (v_perm table for P <= 7, arithmetic for P = 8, byte path for odd boards)
    equals P single-observer calls, which equal the oracle."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    tb = TronBatch(N, P, B)
    tb.rollout(17, 3)
    allobs = tb.observe_all()
    ost = O.TronState(N, P, B)
    ost.board[:], ost.heads[:] = tb.board.cpu().numpy(), tb.heads.cpu().numpy()
    ost.dirs[:], ost.deaths[:] = tb.dirs.cpu().numpy(), tb.deaths.cpu().numpy()
    for p in range(P):
        one = tb.observe(torch.full((B,), p, dtype=torch.int8, device="cuda"))
        ob, oh, od, ok = O.tron_observe(ost, np.full(B, p, np.int8))
        assert np.array_equal(allobs["board"][p].reshape(B, -1).cpu().numpy(), ob)
        assert torch.equal(allobs["board"][p], one["board"]) and torch.equal(allobs["heads"][p], one["heads"])
        assert np.array_equal(allobs["heads"][p].cpu().numpy(), oh) and np.array_equal(allobs["directions"][p].cpu().numpy(), od)
        assert np.array_equal(allobs["deaths"][p].cpu().numpy(), ok)


@pytest.mark.parametrize("N,P,B", [(20, 4, 5000 + 3), (40, 4, 1000), (12, 3, 300), (28, 7, 130), (8, 2, 65), (4, 2, 70),
                                   (9, 6, 777), (13, 8, 300), (16, 8, 200),
                                   # boards that are not whole 16-byte chunks, batches of 16 k games: the flat-stream kernel (64 and
                                   # 16 games per workgroup, last workgroup of 16 / 32 / 48 games), incl. the reference's default 19 x 19
                                   (19, 4, 4096 + 16), (19, 3, 1040), (15, 4, 2048 + 48), (39, 4, 528), (9, 6, 784), (5, 2, 96), (21, 7, 400),
                                   (19, 4, 4096 + 5),
                                   # ... and batches that are not a multiple of 16 games of such boards (the observers' planes off the 16-byte grid)
                                   (19, 4, 20000 + 9), (15, 3, 1001), (39, 4, 100 + 3), (19, 8, 1024 + 9), (5, 2, 7), (19, 4, 1), (21, 7, 70),
                                   # eight players: the fused kernels relabel by arithmetic (cell values 0..8 do not fit the permute table)
                                   (20, 8, 1024 + 7), (19, 8, 1024), (40, 8, 160)])
def test_step_observe_fused_matches_three_calls_and_oracle(N, P, B):
    """crl_tron_step_observe (one launch: [sample ->] next_state -> state_to_observation of all P players) equals
    crl_tron_sample + crl_tron_step + crl_tron_observe_all on a twin batch and the oracle stepped in lockstep: 64 and 16
    games per workgroup (LDS limit), ragged batches, auto-reset on and off, external and sampled actions, and the
    boards / player counts that run the one-game-per-workgroup kernel instead (N*N % 16 != 0, P = 8)."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    seed, first = 77, 900
    a, b = TronBatch(N, P, B, first_env_id=first), TronBatch(N, P, B, first_env_id=first)
    sh, sd = O.tron_start_positions(N, P)
    orc = OracleTron(N, P, B, sh, sd)
    ost = orc.st
    rng = np.random.default_rng(N * 10 + P)
    out = None
    resets = 0
    for t in range(26):
        auto = (t % 5) != 4
        sampled = (t % 2 == 0)            # also on the shapes that run the one-game-per-workgroup kernel (round 2 raised there)
        if sampled:
            act = b.sample(seed)                                     # advances b.tcount; the fused call advances a.tcount
            out = a.step_observe(None, seed=seed, auto_reset=auto, out=out)
        else:
            act = torch.from_numpy(rng.integers(-1, 2, size=(P, B)).astype(np.int8)).cuda()
            out = a.step_observe(act, seed=seed, auto_reset=auto, out=out)
        rew, term, win = b.step(act, auto_reset=auto)
        obs = b.observe_all()
        for k in ("board", "heads", "dirs", "deaths", "tcount"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (k, t)
        assert torch.equal(out["rewards"], rew) and torch.equal(out["terminal"], term) and torch.equal(out["winners"], win), t
        for k in ("board", "heads", "directions", "deaths"):
            assert torch.equal(out[k], obs[k]), (k, t)
        r2, t2, w2 = orc.step(act.cpu().numpy(), auto_reset=auto)
        assert np.array_equal(rew.cpu().numpy(), r2) and np.array_equal(term.cpu().numpy(), t2) and np.array_equal(win.cpu().numpy(), w2)
        if auto:
            resets += int(t2.sum())
        assert np.array_equal(a.board.cpu().numpy(), ost.board) and np.array_equal(a.heads.cpu().numpy(), ost.heads), t
        assert np.array_equal(a.deaths.cpu().numpy(), ost.deaths) and np.array_equal(a.dirs.cpu().numpy(), ost.dirs), t
        for p in (0, P - 1):
            ob, oh, od, ok = O.tron_observe(ost, np.full(B, p, np.int8))
            assert np.array_equal(out["board"][p].reshape(B, -1).cpu().numpy(), ob) and np.array_equal(out["heads"][p].cpu().numpy(), oh)
            assert np.array_equal(out["directions"][p].cpu().numpy(), od) and np.array_equal(out["deaths"][p].cpu().numpy(), ok)
    assert resets > 0


def test_step_observe_full_size_equals_rollout():
    """BASELINE config 2 at full size: T x fused step_observe(sampled actions) leaves the state crl_tron_rollout(T) leaves."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    B, T, seed = 65536, 24, 5
    a, b = TronBatch(20, 4, B), TronBatch(20, 4, B)
    out = None
    for _ in range(T):
        out = a.step_observe(None, seed=seed, out=out)
    b.rollout(T, seed)
    for k in ("board", "heads", "dirs", "deaths", "tcount"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert torch.equal(out["board"], b.observe_all()["board"])
