"""Exhaustive 2-player 3x3 game tree through a TTT backend (SURVEY 8c KAT: 255,168 games)."""
import numpy as np


def count_games(make_backend):
    """Breadth-first over every move sequence; returns (games, wins_p0, wins_p1, draws)."""
    occ = np.zeros((2, 1), np.uint32)
    games = wins0 = wins1 = draws = 0
    depth = 0
    while occ.shape[1] > 0:
        B = occ.shape[1]
        empt = (~(occ[0] | occ[1])) & np.uint32(0x1ff)
        # children: one per (state, empty cell)
        cells = [np.nonzero((empt >> c) & 1)[0] for c in range(9)]
        parent = np.concatenate(cells)
        action = np.concatenate([np.full(len(ix), c, np.int8) for c, ix in enumerate(cells)])
        n = len(parent)
        be = make_backend((3, 3), 3, 2, n)
        be_set(be, occ[:, parent], depth % 2)
        rew, term, win = be.step(action)
        assert (rew[win >= 0] == 1).all() and (rew[win < 0] == 0).all()
        new_occ = be_occ(be)
        tm = term.astype(bool)
        games += int(tm.sum())
        wins0 += int((win[tm] == 0).sum())
        wins1 += int((win[tm] == 1).sum())
        draws += int((win[tm] < 0).sum())
        occ = np.ascontiguousarray(new_occ[:, ~tm])
        depth += 1
    return games, wins0, wins1, draws


def be_set(be, occ, mover):
    if be.name == "oracle":
        be.st.occ[:] = occ
        be.st.winner[:] = -1
        be.st.to_move[:] = mover
    else:
        t = be.torch
        be.tb.occ.copy_(t.from_numpy(np.ascontiguousarray(occ).view(np.int32)).to(be.tb.device))
        be.tb.winner.fill_(-1)
        be.tb.to_move.fill_(mover)


def be_occ(be):
    if be.name == "oracle":
        return be.st.occ.copy()
    return be.tb.occ.cpu().numpy().view(np.uint32)
