"""N > 1 path on CPU: world_size 2 and 3 over gloo.  The stepper behind ShardedRollout is the CPU oracle
(tests may use it), so this checks the sharding arithmetic, global-id RNG keying and the single gather --
the same code path bench.py and a multi-GPU job run with the HIP steppers over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class OracleTronStepper:
    """Duck-types TronBatch (rollout / results) on the CPU oracle."""

    def __init__(self, batch, first_env_id, N=12, P=4):
        from oracle import oracle as O
        self.O, self.first = O, first_env_id
        self.sh, self.sd = O.tron_start_positions(N, P)
        self.st = O.TronState(N, P, batch)
        O.tron_reset(self.st, self.sh, self.sd)

    def rollout(self, steps, seed):
        self.O.tron_rollout(self.st, seed, self.first, steps, self.sh, self.sd)

    def results(self, copy=True):
        s = self.st
        cols = [s.n_episodes, s.len_sum, s.last_winners.astype(np.uint32)] + list(s.win_count) + [r.view(np.uint32) for r in s.ret_sum]
        return torch.from_numpy(np.stack([c.astype(np.uint32).view(np.int32) for c in cols], axis=1).copy())

    # the 16-bit rows of TronBatch.results_packed: n_episodes, len_sum, last_winners, tstep, ret_sum[P] (low 16 bits)
    def results_packed(self, copy=True):
        s = self.st
        cols = [s.n_episodes, s.len_sum, s.last_winners.astype(np.uint32), s.tstep] + [r.view(np.uint32) for r in s.ret_sum]
        return torch.from_numpy(np.stack([c.astype(np.uint32).astype(np.uint16).view(np.int16) for c in cols], axis=1).copy())

    def packed_rows_exact(self):
        return int(self.st.tcount.max()) <= 3276


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from colosseumrl_amd.parallel import ShardedRollout, shard_bounds
        sr = ShardedRollout(lambda batch, first_env_id: OracleTronStepper(batch, first_env_id), total)
        assert (sr.lo, sr.hi) == shard_bounds(total, rank, world)
        launches = sr.rollout(70, seed=5, chunk=32)
        assert launches == 3
        allres = sr.gather(packed=False)                  # a snapshot (copy=True): later gathers reuse the receive buffer
        assert allres.shape[0] == total and allres.dtype == torch.int32
        np.save(os.path.join(out_dir, "rank%d.npy" % rank), allres.numpy())
        narrow = sr.gather()                              # 70 steps: the 16-bit rows are exact and are what "auto" ships
        assert narrow.dtype == torch.int16 and narrow.shape == (total, 8)
        np.save(os.path.join(out_dir, "rank%d_packed.npy" % rank), narrow.numpy())
        dst = world - 1                                   # the gather to ONE rank (what bench.py times)
        rooted = sr.gather(dst=dst, packed=False)
        assert (rooted is None) == (rank != dst)
        if rank == dst:
            assert torch.equal(rooted, allres)
            assert torch.equal(sr.gather(dst=dst), narrow)
        else:
            assert sr.gather(dst=dst) is None
        if world >= 3:                                    # a sub-group: `dst` names a rank OF THE GROUP (global rank differs)
            sub = dist.new_group([1, 2])
            if rank in (1, 2):
                sg = ShardedRollout(lambda batch, first_env_id: OracleTronStepper(batch, first_env_id), 20, group=sub)
                assert (sg.world, sg.rank) == (2, rank - 1)
                sg.rollout(9, seed=2, chunk=9)
                got = sg.gather(dst=1, packed=False)      # group rank 1 = global rank 2
                assert (got is None) == (rank != 2)
                if rank == 2:
                    one = OracleTronStepper(20, 0)
                    one.rollout(9, 2)
                    assert torch.equal(got, one.results())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 64), (3, 50)])
def test_sharded_rollout_matches_single_process(tmp_path, world, total):
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    single = OracleTronStepper(total, 0)
    single.rollout(32, 5)
    single.rollout(32, 5)
    single.rollout(6, 5)
    want, want_packed = single.results().numpy(), single.results_packed().numpy()
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got, want), "rank %d gathered a different global result" % r
        got = np.load(os.path.join(str(tmp_path), "rank%d_packed.npy" % r))
        assert np.array_equal(got, want_packed), "rank %d gathered different 16-bit rows" % r
    assert want[:, 0].sum() > 0 and np.array_equal(want_packed[:, :2], want[:, :2])


def test_shard_bounds_cover_everything():
    from colosseumrl_amd.parallel import shard_bounds
    for total in (1, 7, 64, 65536, 524288):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_bounds(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)
