"""Uniform numpy-in/numpy-out adapters over (a) the CPU oracle and (b) the HIP product path,
so the same replay code checks both against the golden vectors."""
import numpy as np

from oracle import oracle as O


# ------------------------------------------------------------------ Tron
class OracleTron:
    name = "oracle"

    def __init__(self, N, P, B, start_heads, start_dirs):
        self.N, self.P, self.B = N, P, B
        self.sh, self.sd = np.asarray(start_heads, np.int16), np.asarray(start_dirs, np.int8)
        self.st = O.TronState(N, P, B)
        O.tron_reset(self.st, self.sh, self.sd)

    def reset(self, mask=None):
        if mask is None:
            O.tron_reset(self.st, self.sh, self.sd)
            return
        idx = np.nonzero(mask)[0]
        if len(idx) == 0:
            return
        sub = O.TronState(self.N, self.P, len(idx))
        O.tron_reset(sub, self.sh, self.sd)
        self.st.board[idx] = sub.board
        self.st.heads[:, idx] = sub.heads
        self.st.dirs[:, idx] = sub.dirs
        self.st.deaths[:, idx] = sub.deaths

    def set_state(self, board, heads, dirs, deaths):
        self.st.board[:] = board
        self.st.heads[:] = heads
        self.st.dirs[:] = dirs
        self.st.deaths[:] = deaths

    def step(self, actions, auto_reset=False):
        r, t, w = O.tron_step(self.st, actions)
        if auto_reset:
            self.reset(t)
        return r, t, w

    def state(self):
        s = self.st
        return dict(board=s.board.copy(), heads=s.heads.copy(), dirs=s.dirs.copy(), deaths=s.deaths.copy())

    def observe(self, player):
        return O.tron_observe(self.st, player)


class HipTron:
    name = "hip"

    def __init__(self, N, P, B, start_heads, start_dirs):
        import torch
        from colosseumrl_amd.batched import TronBatch
        self.torch = torch
        self.N, self.P, self.B = N, P, B
        self.tb = TronBatch(N, P, B, start=(list(map(int, start_heads)), list(map(int, start_dirs))))

    def _dev(self, a, dtype):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(self.tb.device)

    def reset(self, mask=None):
        self.tb.reset(None if mask is None else self._dev(mask.astype(np.uint8), self.torch.uint8))

    def set_state(self, board, heads, dirs, deaths):
        t = self.torch
        self.tb.board.copy_(self._dev(board, t.int8))
        self.tb.heads.copy_(self._dev(heads, t.int16))
        self.tb.dirs.copy_(self._dev(dirs, t.int8))
        self.tb.deaths.copy_(self._dev(deaths, t.int8))

    step_kernel = "auto"            # "bytes" / "staged" pin one of crl_tron_step's two interchangeable kernels

    def step(self, actions, auto_reset=False):
        r, t, w = self.tb.step(self._dev(actions, self.torch.int8), auto_reset=auto_reset, kernel=self.step_kernel)
        return r.cpu().numpy(), t.cpu().numpy(), w.cpu().numpy()

    def state(self):
        tb = self.tb
        return dict(board=tb.board.cpu().numpy(), heads=tb.heads.cpu().numpy(), dirs=tb.dirs.cpu().numpy(),
                    deaths=tb.deaths.cpu().numpy())

    def observe(self, player):
        o = self.tb.observe(self._dev(player, self.torch.int8))
        return (o["board"].reshape(self.B, -1).cpu().numpy(), o["heads"].cpu().numpy(),
                o["directions"].cpu().numpy(), o["deaths"].cpu().numpy())


# ------------------------------------------------------------------ TicTacToe
class OracleTTT:
    name = "oracle"

    def __init__(self, dims, K, P, B):
        self.st = O.TTTState(dims, K, P, B)
        self.P, self.B = P, B

    def reset(self, mask=None):
        m = np.ones(self.B, bool) if mask is None else mask.astype(bool)
        self.st.occ[:, m] = 0
        self.st.winner[m] = -1
        self.st.to_move[m] = 0

    def step(self, action, auto_reset=False):
        r, t, w = O.ttt_step(self.st, action)
        if auto_reset:
            self.reset(t)
        return r, t, w

    def board(self):
        return self.st.board()

    def valid(self):
        full = np.uint32((1 << self.st.n_cells) - 1) if self.st.n_cells < 32 else np.uint32(0xffffffff)
        return full & ~np.bitwise_or.reduce(self.st.occ, axis=0)

    def winner(self):
        return self.st.winner.copy()

    def to_move(self):
        return self.st.to_move.copy()

    def lines(self):
        return sorted(int(x) for x in self.st.lines)


class HipTTT:
    name = "hip"

    def __init__(self, dims, K, P, B):
        import torch
        from colosseumrl_amd.batched import TTTBatch
        self.torch = torch
        self.tb = TTTBatch(dims, K, P, B)
        self.P, self.B = P, B

    def reset(self, mask=None):
        self.tb.reset(None if mask is None else self.torch.from_numpy(mask.astype(np.uint8)).to(self.tb.device))

    def step(self, action, auto_reset=False):
        a = self.torch.from_numpy(np.ascontiguousarray(action, dtype=np.int8)).to(self.tb.device)
        r, t, w = self.tb.step(a, auto_reset=auto_reset)
        return r.cpu().numpy(), t.cpu().numpy(), w.cpu().numpy()

    def board(self):
        return self.tb.board().cpu().numpy()

    def valid(self):
        return self.tb.valid_mask().cpu().numpy().view(np.uint32)

    def winner(self):
        return self.tb.winner.cpu().numpy()

    def to_move(self):
        return self.tb.to_move.cpu().numpy()

    def lines(self):
        return sorted(self.tb.lines())


# ------------------------------------------------------------------ Blokus
class OracleBlokus:
    name = "oracle"

    def __init__(self, B):
        self.B = B
        self.st = O.BlokusState(B)

    def set_state(self, board, inv, score, rnd, to_move):
        self.st.set_board(board)
        self.st.inv[:] = inv
        self.st.score[:] = score
        self.st.round[:] = rnd
        self.st.to_move[:] = to_move

    def reset(self, mask=None):
        fresh = O.BlokusState(self.B)
        m = np.ones(self.B, bool) if mask is None else np.asarray(mask).astype(bool)
        for k in ("occ", "inv", "score", "round", "to_move"):
            getattr(self.st, k)[m] = getattr(fresh, k)[m]

    def valid(self, cap, player=None):
        return O.blokus_valid(self.st, player=player, cap=cap, n_threads=8)

    def step(self, action):
        return O.blokus_step(self.st, action, n_threads=8)

    def state(self):
        s = self.st
        return dict(board=s.board, inv=s.inv.copy(), score=s.score.copy(), round=s.round.copy(), to_move=s.to_move.copy())


class HipBlokus:
    name = "hip"

    def __init__(self, B):
        import torch
        from colosseumrl_amd.batched import BlokusBatch
        self.torch = torch
        self.B = B
        self.bb = BlokusBatch(B)

    def set_state(self, board, inv, score, rnd, to_move):
        t = self.torch
        board = np.asarray(board)
        occ = np.zeros((self.B, 4, 20), np.uint32)
        for c in range(4):
            occ[:, c, :] = ((board == c + 1).astype(np.uint32) << np.arange(20, dtype=np.uint32)[None, None, :]).sum(axis=2)
        dev = self.bb.device
        self.bb.occ.copy_(t.from_numpy(occ.view(np.int32)).to(dev))
        self.bb.inv.copy_(t.from_numpy(np.asarray(inv, np.uint32).view(np.int32).reshape(self.B, 4).copy()).to(dev))
        self.bb.score.copy_(t.from_numpy(np.asarray(score, np.int32).reshape(self.B, 4).copy()).to(dev))
        self.bb.round.copy_(t.from_numpy(np.asarray(rnd, np.int32).reshape(self.B).copy()).to(dev))
        self.bb.to_move.copy_(t.from_numpy(np.asarray(to_move, np.int32).reshape(self.B).copy()).to(dev))

    def valid(self, cap, player=None):
        pl = None if player is None else self.torch.from_numpy(np.ascontiguousarray(player, np.int8)).to(self.bb.device)
        count, mask = self.bb.valid(player=pl, want_mask=True)
        count = count.cpu().numpy()
        mask = mask.cpu().numpy().view(np.uint32)
        ids = np.full((self.B, cap), -1, np.int32)
        for e in range(self.B):
            bits = np.unpackbits(mask[e].view(np.uint8), bitorder="little")
            nz = np.nonzero(bits)[0]
            assert len(nz) == count[e], (len(nz), count[e])
            ids[e, :len(nz)] = nz[:cap]
        return count, ids

    def step(self, action):
        a = self.torch.from_numpy(np.ascontiguousarray(action, np.int32)).to(self.bb.device)
        r, t, w = self.bb.step(a)
        return r.cpu().numpy(), t.cpu().numpy(), w.cpu().numpy()

    def state(self):
        bb = self.bb
        return dict(board=bb.board().cpu().numpy(), inv=bb.inv.cpu().numpy().view(np.uint32), score=bb.score.cpu().numpy(),
                    round=bb.round.cpu().numpy(), to_move=bb.to_move.cpu().numpy())
