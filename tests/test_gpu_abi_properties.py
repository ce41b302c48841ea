"""Properties the C ABI promises (include/colosseum_hip.h): asynchronous, graph-capturable, stream-ordered,
usable from several host threads / contexts at once."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_calls_are_graph_capturable():
    """step and rollout launches record into a HIP graph and replay to the same result as eager calls."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    a, b = TronBatch(20, 4, 4096), TronBatch(20, 4, 4096)
    act = torch.randint(-1, 2, (4, 4096), dtype=torch.int8, device="cuda")
    for tb in (a, b):                       # warm-up outside capture (one-time kernel attribute opt-in happens here)
        tb.rollout(4, 1)
        tb.step(act)
    torch.cuda.synchronize()
    oa, ob = a.step_observe(None, seed=1), b.step_observe(None, seed=1)   # also allocates the observation buffers
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        a.rollout(16, 1)
        a.step(act, auto_reset=True)
        a.step_observe(None, seed=1, out=oa)        # the fused per-step call (sampled actions advance tcount)
        a.step_observe(act, out=oa)
    for _ in range(3):
        g.replay()
    for _ in range(3):                      # capture itself does not execute: 3 replays == 3 eager rounds
        b.rollout(16, 1)
        b.step(act, auto_reset=True)
        b.step_observe(None, seed=1, out=ob)
        b.step_observe(act, out=ob)
    torch.cuda.synchronize()
    assert torch.equal(a.board, b.board) and torch.equal(a.heads, b.heads) and torch.equal(a.ret_sum, b.ret_sum)
    assert torch.equal(a.tcount, b.tcount) and int(a.tcount[0]) == 4 + 1 + 3 * 17
    assert torch.equal(oa["board"], ob["board"]) and torch.equal(a.results(), b.results())


@pytest.mark.parametrize("N,P,T,kernel", [(40, 4, 64, "auto"), (40, 4, 8, "auto"), (20, 2, 64, "auto"), (20, 4, 1, "auto"), (12, 6, 50, "auto"),
                                         (45, 3, 30, "auto"), (33, 4, 300, "bits"), (20, 4, 16383 + 9, "auto")])
def test_every_rollout_kernel_is_graph_capturable(N, P, T, kernel):
    """The rollout kernels the first test does not reach -- the bitboard kernel with its replay kernel behind it (two launches),
    the lane-per-player kernel in global memory, the two-lanes-per-game kernel, the lane-per-game kernels, a split launch --
    record into a HIP graph and replay to what eager calls give."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    a, b = TronBatch(N, P, 2048 + 3), TronBatch(N, P, 2048 + 3)
    for tb in (a, b):                       # warm-up outside capture (one-time kernel attribute opt-in happens here)
        tb.rollout(T, 2, kernel=kernel)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        a.rollout(T, 2, kernel=kernel)
    for _ in range(3):
        g.replay()
        b.rollout(T, 2, kernel=kernel)
    torch.cuda.synchronize()
    for k in ("board", "heads", "dirs", "deaths", "tcount", "tstep", "n_episodes", "ret_sum", "win_count"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert torch.equal(a.results(copy=False), b.results(copy=False))


def test_side_stream_ordering():
    """Launches go to torch's CURRENT stream: work issued on a side stream is ordered on that stream."""
    import torch
    from colosseumrl_amd.batched import TTTBatch
    side = torch.cuda.Stream()
    ref = TTTBatch((3, 5), 3, 3, 8192)
    ref.rollout(64, 9)
    tb = TTTBatch((3, 5), 3, 3, 8192)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(4):
            tb.rollout(16, 9)
    side.synchronize()
    assert torch.equal(tb.occ, ref.occ) and torch.equal(tb.win_count, ref.win_count)


def test_contexts_are_independent_across_threads():
    """Several env instances live in one process (reference MatchmakingServer.py:128-135): no shared mutable state."""
    from colosseumrl_amd.batched import BlokusBatch, TronBatch
    out = {}

    def worker(i):
        tb = TronBatch(12 + i, 3, 700 + i, first_env_id=1000 * i)
        bb = BlokusBatch(40 + i, first_env_id=77 * i)
        for _ in range(3):
            tb.rollout(20, i)
            bb.rollout(6, i)
        out[i] = (tb.board.cpu().numpy().copy(), bb.occ.cpu().numpy().copy())

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(4):                      # same work done alone, sequentially
        tb = TronBatch(12 + i, 3, 700 + i, first_env_id=1000 * i)
        bb = BlokusBatch(40 + i, first_env_id=77 * i)
        for _ in range(3):
            tb.rollout(20, i)
            bb.rollout(6, i)
        assert np.array_equal(out[i][0], tb.board.cpu().numpy()) and np.array_equal(out[i][1], bb.occ.cpu().numpy())


def test_error_reporting_never_throws_across_the_abi():
    import ctypes as C
    from colosseumrl_amd import _native
    lib = _native.require_gpu()
    ctx = C.c_void_p()
    assert lib.crl_ttt_create(6, 6, 1, 3, 2, C.byref(ctx)) == -1                    # 36 cells > 32
    assert b"32" in lib.crl_last_error()
    assert lib.crl_tron_step(None, 4, None, None, None, None, None, None, None, None, 0, None) == -1
    assert lib.crl_blokus_create(C.byref(ctx)) == 0
    assert lib.crl_tron_step(ctx, 4, None, None, None, None, None, None, None, None, 0, None) == -1   # wrong context kind
    assert b"tron" in lib.crl_last_error()
    lib.crl_destroy(ctx)


@pytest.mark.parametrize("game", ["tron", "ttt", "blokus"])
def test_sample_then_step_equals_rollout(game):
    """SURVEY 8(b) batched surface: `sample()` is the fused rollout's random agent for one step, so
    T x (sample; step with auto-reset) must leave exactly the state (and step counters) of rollout(T)."""
    import torch
    from colosseumrl_amd.batched import TronBatch, TTTBatch, BlokusBatch
    seed, first = 0xABCDEF0123, 777
    if game == "tron":
        mk, T, keys = (lambda: TronBatch(20, 5, 1000, first_env_id=first)), 60, ("board", "heads", "dirs", "deaths", "tcount")
    elif game == "ttt":
        mk, T, keys = (lambda: TTTBatch((3, 5), 3, 3, 2000, first_env_id=first)), 50, ("occ", "winner", "to_move", "tcount")
    else:
        mk, T, keys = (lambda: BlokusBatch(64, first_env_id=first)), 150, ("occ", "inv", "score", "round", "to_move", "tcount")
    a, b = mk(), mk()
    a.rollout(T, seed)
    for _ in range(T):
        b.step(b.sample(seed), auto_reset=True)
    for k in keys:
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    # a sample without advance is repeatable and does not move the counter
    before = b.tcount.clone()
    s1, s2 = b.sample(seed, advance=False), b.sample(seed, advance=False)
    assert torch.equal(s1, s2) and torch.equal(before, b.tcount)


def test_tron_check_state():
    """The invariant the LDS rollout kernels rely on holds after reset / step / rollout and is reported when broken."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    tb = TronBatch(20, 4, 3000)
    assert tb.check_state() == 0
    tb.rollout(300, 5)
    for _ in range(5):
        tb.step(tb.sample(5), auto_reset=True)
    assert tb.check_state() == 0
    tb.board[7, int(tb.heads[2, 7])] = 1          # player 2's head cell now claims to be player 0's
    tb.heads[1, 11] = 399
    tb.board[11, 399] = 0
    assert tb.check_state() == 2


@pytest.mark.parametrize("kernel", ["bytes", "bits", "quad", "qbits"])
def test_hand_made_states_with_wild_heads_stay_inside_the_slab(kernel):
    """Heads outside the board (a hand-uploaded or corrupted state) are clamped onto it by the LDS rollout kernels: the
    results of such games are unspecified, but the launch completes, reports no error, and every OTHER game of the batch
    still equals an untouched twin (no write outside the broken game's own slab)."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    a, b = TronBatch(20, 4, 1024), TronBatch(20, 4, 1024)
    bad = [5, 64, 700]
    for e in bad:
        a.heads[1, e] = 30000
        a.heads[3, e] = -7
    assert a.check_state() == len(bad)
    a.rollout(64, 3, kernel=kernel)
    b.rollout(64, 3, kernel=kernel)
    torch.cuda.synchronize()
    keep = torch.ones(1024, dtype=torch.bool, device="cuda")
    keep[bad] = False
    assert torch.equal(a.board[keep], b.board[keep]) and torch.equal(a.heads[:, keep], b.heads[:, keep])
    assert torch.equal(a.n_episodes[keep], b.n_episodes[keep])


def test_observe_with_out_of_range_player_ids():
    """ANY observer id is an observer, as in the reference (ABI 109; TronGridEnvironment.py:393, CyTronGrid.pyx:70-71):
    ids congruent mod P up to id == P observe alike, and every id -- negative, P, beyond P -- matches the oracle (which the
    reference-generated tron_observe_wrap_* fixtures pin) instead of reading a wild table entry."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    from oracle import oracle as O
    tb = TronBatch(20, 4, 512)
    tb.rollout(9, 1)
    pl = torch.zeros((512,), dtype=torch.int8, device="cuda")
    ref = tb.observe(pl)
    pl[::3] = 4
    pl[1::3] = -4
    got = tb.observe(pl)
    for k in ("board", "heads", "directions", "deaths"):
        assert torch.equal(ref[k], got[k]), k
    pl = torch.randint(-128, 128, (512,), dtype=torch.int8, device="cuda")
    got = tb.observe(pl)
    st = O.TronState(20, 4, 512)
    st.board[:] = tb.board.cpu().numpy().reshape(512, -1)
    st.heads[:], st.dirs[:], st.deaths[:] = tb.heads.cpu().numpy(), tb.dirs.cpu().numpy(), tb.deaths.cpu().numpy()
    ob, oh, od, ok = O.tron_observe(st, pl.cpu().numpy())
    for k, want in (("board", ob), ("heads", oh), ("directions", od), ("deaths", ok)):
        assert np.array_equal(got[k].cpu().numpy().reshape(want.shape), want), k


def test_wait_on_mapped_memory_ends_the_queued_work():
    """`wait()` of the batched steppers (crl_stream_wait_mapped on the launch stream): when it returns, everything queued
    before it has run.  Checked on DATA, without any synchronise: an asynchronous copy of the result rows into pinned host
    memory is queued behind the rollout, and when `wait()` returns the host buffer holds the rows a twin produced that
    synchronised.  (Event / stream queries are no witness: the runtime learns of a completion only when it looks.)"""
    import torch
    from colosseumrl_amd.batched import BlokusBatch, TronBatch, TTTBatch
    from colosseumrl_amd.parallel import ShardedRollout
    for make, T in ((lambda: TronBatch(20, 4, 65536), 20), (lambda: TronBatch(20, 4, 65536), 3000), (lambda: TTTBatch((3, 5), 3, 3, 4096), 64),
                    (lambda: BlokusBatch(256), 40)):
        a, b = make(), make()
        host = torch.zeros_like(a.results(), device="cpu").pin_memory()
        torch.cuda.synchronize()
        for i in range(3):
            b.rollout(T, 9)
            torch.cuda.synchronize()
            want = b.results().cpu()
            host.fill_(-1)
            a.rollout(T, 9)
            host.copy_(a.results(copy=False), non_blocking=True)
            a.wait()
            assert torch.equal(host, want), (T, i)                # no synchronise in between: the wait covered launch and copy
    side = torch.cuda.Stream()                                    # the wait follows torch's CURRENT stream
    sr = ShardedRollout(lambda batch, first_env_id: TronBatch(20, 4, batch, first_env_id=first_env_id), 8192)
    twin = TronBatch(20, 4, 8192)
    twin.rollout(500, 1)
    torch.cuda.synchronize()
    want = twin.results().cpu()
    host = torch.zeros_like(want).pin_memory()
    with torch.cuda.stream(side):
        sr.rollout(500, 1, 250)
        host.copy_(sr.stepper.results(copy=False), non_blocking=True)
        sr.wait()
        assert torch.equal(host, want)


@pytest.mark.parametrize("N,P,T,kernel", [(20, 4, 20, "auto"), (20, 4, 16383 + 40, "auto"), (40, 4, 64, "auto"), (12, 6, 50, "auto"),
                                         (20, 4, 30, "global")])
def test_rollout_with_events_attached_to_the_dispatches(N, P, T, kernel):
    """crl_tron_rollout_timed: the start / stop events ride in the first / last dispatch of the rollout
    (hipExtLaunchKernelGGL).  Same results as the plain call, a plausible elapsed time, either event optional."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    B = 4096 + 3
    a, b = TronBatch(N, P, B), TronBatch(N, P, B)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with pytest.raises(ValueError):
        a.rollout(T, 5, kernel=kernel, events=(e0, e1))          # never recorded: torch has not created the HIP events
    e0.record(); e1.record()
    torch.cuda.synchronize()
    a.rollout(T, 5, kernel=kernel, events=(e0, e1))
    b.rollout(T, 5, kernel=kernel)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    assert 0.0 < ms < 500.0, ms          # (a lone wave per SIMD at B = 4,099: ~3-5 us per step, box clocks vary)
    for k in ("board", "heads", "dirs", "deaths", "tcount", "ret_sum", "n_episodes"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    a.rollout(7, 5, kernel=kernel, events=(None, e1))
    a.rollout(7, 5, kernel=kernel, events=(e0, None))
    b.rollout(14, 5, kernel=kernel)
    assert torch.equal(a.board, b.board) and torch.equal(a.results(), b.results())


def test_stream_wait_mapped_orders_behind_the_stream_and_falls_back():
    """crl_stream_wait_mapped: the flag arrives behind everything queued on the stream (a long rollout's results are
    complete when the call returns, with NO other synchronisation of that stream), a fresh sequence number per call, and with
    a zero time-out the call takes its fallback (hipStreamSynchronize + flag check) and still succeeds."""
    import ctypes as C
    import torch
    from colosseumrl_amd import _native
    from colosseumrl_amd._native import check
    from colosseumrl_amd.batched import TronBatch
    from colosseumrl_amd.single import HostBlob
    lib = _native.lib()
    side = torch.cuda.Stream()                                    # torch owns the stream; the C ABI sees its handle
    stream = C.c_void_p(side.cuda_stream)
    flag = HostBlob(lib, [("seq", np.uint32, 1)])
    host_ptr = C.c_void_p(flag.v["seq"].ctypes.data)
    ref = TronBatch(20, 4, 4096)
    ref.rollout(3000, 9)
    want = ref.len_sum.cpu()
    tb = TronBatch(20, 4, 4096)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        tb.rollout(3000, 9)                                       # ~1 ms of work queued on the side stream, asynchronous
    check(lib.crl_stream_wait_mapped(stream, flag.d["seq"], host_ptr, 1, 5.0), "crl_stream_wait_mapped")
    assert int(flag.v["seq"][0]) == 1
    # the copy below runs on the DEFAULT stream, which never waited for the side stream: it sees the finished rollout only
    # because the wait above really covered it
    assert torch.equal(tb.len_sum.cpu(), want)
    for seq, timeout in ((2, 5.0), (7, 0.0), (0xFFFFFFFF, 5.0), (3, 0.0)):
        check(lib.crl_stream_wait_mapped(stream, flag.d["seq"], host_ptr, seq, timeout), "crl_stream_wait_mapped")
        assert int(flag.v["seq"][0]) == seq
    assert lib.crl_stream_wait_mapped(stream, None, host_ptr, 4, 1.0) != 0 and b"NULL" in lib.crl_last_error()
    torch.cuda.synchronize()


def test_shipped_build_compiles_no_bounds_asserts():
    """crl_diag_bounds on the library the suite runs on: the shipped build has no asserts (zeros, 'compiled' False); under
    tools/gpu_bounds.sh (CRL_EXPECT_BOUNDS_BUILD=1) it is the assert build, whose self-test has then just passed."""
    import os
    from colosseumrl_amd import _native
    rep = _native.bounds_report()
    assert rep["compiled"] == (os.environ.get("CRL_EXPECT_BOUNDS_BUILD") == "1")
    if not rep["compiled"]:
        assert all(v == 0 for k in ("tron", "ttt", "blokus") for v in rep[k].values())


@pytest.mark.parametrize("N,P,B,shift", [(19, 4, 300, 1), (19, 4, 4096, 7), (5, 2, 70, 12), (21, 3, 129, 3), (41, 4, 65, 9), (15, 8, 100, 5)])
def test_boards_at_any_byte_alignment(N, P, B, shift):
    """Boards that are not whole 16-byte chunks may lie at ANY address (the header asks for 16-byte alignment only where
    N * N % 16 == 0): every call then takes its byte / one-game-per-workgroup / global-memory kernel.  A batch whose board
    buffer starts `shift` bytes off the grid must do what an aligned twin does: reset, step, sampled step_observe, rollouts on
    every kernel choice that accepts it, observe, observe_all, ranking."""
    import torch
    from colosseumrl_amd.batched import TronBatch
    NN = N * N
    a, b = TronBatch(N, P, B, first_env_id=5), TronBatch(N, P, B, first_env_id=5)
    raw = torch.zeros(B * NN + 32, dtype=torch.int8, device="cuda")
    base = (-raw.data_ptr()) % 16 + shift                         # `shift` bytes behind a 16-byte boundary
    b.board = raw[base:base + B * NN].view(B, NN)
    assert b.board.data_ptr() % 16 == shift % 16 and b.board.is_contiguous()
    b.reset()
    out_a = out_b = None
    rng = np.random.default_rng(N + shift)
    for t in range(6):
        act = torch.from_numpy(rng.integers(-1, 2, size=(P, B)).astype(np.int8)).cuda()
        a.step(act, auto_reset=True); b.step(act, auto_reset=True)
        out_a, out_b = a.step_observe(None, 3, True, out_a), b.step_observe(None, 3, True, out_b)
        assert torch.equal(out_a["board"], out_b["board"]) and torch.equal(out_a["heads"], out_b["heads"]), t
    for T, kernel in ((1, "auto"), (9, "auto"), (70, "auto"), (300, "auto"), (40, "gquad"), (40, "global"), (40, "qbits"), (40, "bytes")):
        a.rollout(T, 11, kernel=kernel); b.rollout(T, 11, kernel=kernel)
        for k in ("board", "heads", "dirs", "deaths", "tcount", "n_episodes", "ret_sum"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (T, kernel, k)
    who = (torch.arange(B, device="cuda") % P).to(torch.int8)
    oa, ob = a.observe(who), b.observe(who)
    assert torch.equal(oa["board"], ob["board"]) and torch.equal(a.ranking(), b.ranking())
    assert torch.equal(a.observe_all()["board"], b.observe_all()["board"])
    mask = (torch.arange(B, device="cuda") % 3 == 0).to(torch.uint8)
    a.reset(mask); b.reset(mask)
    assert torch.equal(a.board, b.board) and torch.equal(a.heads, b.heads)
    assert int(raw[:base].abs().sum()) == 0 and int(raw[base + B * NN:].abs().sum()) == 0      # nothing written outside the boards
