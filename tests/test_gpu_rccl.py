"""The RCCL path of the end-of-rollout gather, executed on the hardware a 1-GPU box has: a world of ONE rank.

colosseumrl_amd.parallel runs its single collective (all_gather_into_tensor, backend "nccl" = RCCL) whenever a process
group is initialised, so a one-rank group exercises exactly the code the 8-GPU run uses (communicator creation, the
device-buffer collective on the stepper's stream, the preallocated receive buffer).  The child process creates the group
before it touches the GPU in any other way.  Multi-rank logic (shard bounds, ragged shards, global RNG ids) is covered
on CPU by tests/test_parallel_gloo.py (gloo, world 2 and 3)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))     # RCCL communicator first, before any kernel
from colosseumrl_amd.batched import TronBatch, TTTBatch
from colosseumrl_amd.parallel import ShardedRollout, gather_results
assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
B = 4096 + 7
sr = ShardedRollout(lambda batch, first_env_id: TronBatch(20, 4, batch, device="cuda:0", first_env_id=first_env_id), B)
sr.rollout(48, seed=3, chunk=20)
got = sr.gather(packed=False)                                             # all_gather_into_tensor over RCCL, int32 rows
want = sr.stepper.results_from_columns()
assert got.data_ptr() != sr.stepper.results(copy=False).data_ptr(), "gather() must have gone through the collective's receive buffer"
assert torch.equal(got, want), "gathered rows differ from the local rows"
assert int(got[:, 0].sum()) > B
ref = TronBatch(20, 4, B, device="cuda:0")
ref.rollout(48, seed=3)
assert torch.equal(ref.results(), got)
narrow = sr.gather()                                                      # 48 steps: the 16-byte rows are exact and shipped
assert narrow.dtype == torch.int16 and tuple(narrow.shape) == (B, 8)
assert torch.equal(narrow, sr.stepper.results_packed_from_columns()) and torch.equal(narrow, ref.results_packed())
assert torch.equal(narrow[:, :2].to(torch.int32), got[:, :2]) and torch.equal(narrow[:, 4:8].to(torch.int32), got[:, 7:11])
live = sr.gather(copy=False)
assert live.data_ptr() == sr.gather(copy=False).data_ptr() and live.data_ptr() != narrow.data_ptr()   # reused buffer vs snapshot
rooted = sr.gather(dst=0, packed=False)                                   # torch.distributed.gather: what bench.py times
assert rooted is not None and torch.equal(rooted, want)
assert torch.equal(sr.gather(dst=0), narrow)
sr.rollout(3300, seed=3, chunk=3300)                                      # beyond 3,276 steps the 16-bit totals may wrap: int32 rows
assert not sr.stepper.packed_rows_exact() and sr.gather().dtype == torch.int32
assert torch.equal(sr.gather(), sr.stepper.results_from_columns())
st = ShardedRollout(lambda batch, first_env_id: TTTBatch((3, 5), 3, 3, batch, device="cuda:0", first_env_id=first_env_id), 1000)
st.rollout(40, seed=1, chunk=40)
assert torch.equal(st.gather(), st.stepper.results_from_columns())
# The stream-scoped wait (crl_stream_wait_mapped) covers the collective as well: a synchronous c10d collective makes the
# launch stream wait for the communicator's stream, so what is queued behind it -- here an async copy of the received rows
# to pinned memory, then the signal kernel -- runs after the rows have arrived.  Checked on data, no synchronise in between.
sw = ShardedRollout(lambda batch, first_env_id: TronBatch(20, 4, batch, device="cuda:0", first_env_id=first_env_id), 65536)
twin = TronBatch(20, 4, 65536, device="cuda:0")
host = torch.zeros((65536, 8), dtype=torch.int16).pin_memory()
sw.warm_collective(0)
for i in range(6):
    twin.rollout(20, 5)
    torch.cuda.synchronize()
    want_rows = twin.results_packed().cpu()
    host.fill_(-1)
    sw.rollout(20, 5, 20)
    rows = sw.gather(dst=0, copy=False) if i %% 2 == 0 else sw.gather(dst=None, copy=False)
    host.copy_(rows, non_blocking=True)
    sw.wait()
    assert torch.equal(host, want_rows), "rows incomplete when the stream wait returned (round %%d)" %% i
t = torch.ones(8, device="cuda:0")
dist.all_reduce(t)
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl-world1-ok")
"""


def test_gather_through_rccl_world_of_one(run_fresh):
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="@FREE_PORT@",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    rc, out = run_fresh([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, timeout=300)
    assert rc == 0 and "rccl-world1-ok" in out, out[-3000:]


def test_bench_under_torchrun_world_of_one(run_fresh):
    """bench.py launched the way the driver launches the multi-GPU case (torch.distributed.run), with one rank: the
    process group is created, the timed region contains the RCCL gather, and the JSON line says so."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "@FREE_PORT@", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--only-headline", "--no-cpu-baseline"]
    rc, out = run_fresh(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT, timeout=600)
    assert rc == 0, out[-3000:]
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["config"]["gather"].startswith("rccl") and rec["value"] > 1e8


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly ONE line, got %d: %r" % (len(lines), stdout[:2000])
    assert len(lines[0]) < 4096
    import json
    return json.loads(lines[0])


def test_bench_driver_style_line(run_fresh):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's N = 1 command, minus the slow evidence sections):
    stdout is ONE compact line; its episodes / mean_episode_len describe the W + K = 25 steps that were timed -- not the
    hundreds of steps the later timing passes add to the stepper (round 3's live-buffer bug: 3.1e6 episodes, 9.25)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--only-headline",
           "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "LOCAL_RANK")}
    rc, out, err = run_fresh(cmd, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT, timeout=600, split=True)
    assert rc == 0, err[-3000:]
    rec = _one_json_line(out)
    assert rec["n_gpus"] == 1 and rec["steps"] == 20 and rec["warmup"] == 5 and rec["config"]["gather"].startswith("none")
    assert rec["config"]["workload"] == "tron_p4_n20_b65536" and rec["config"]["global_games"] == 65536
    assert 1.5e5 < rec["config"]["episodes"] <= 65536 * 25 // 2
    assert abs(rec["config"]["mean_episode_len"] - 8.45) < 0.1
    assert rec["value"] == pytest.approx(65536 * 20 / (rec["ms_per_step"] * 20 * 1e-3), rel=1e-4) and rec["value"] > 1e9
    r = rec["roofline"]
    assert r["kernel"] == "tron_rollout_quad_kernel" and 0.3 < r["frac"] < 1.2
    # the bound names the nearer roof: physical HBM traffic / time against 8 TB/s, or VALU instructions / time against the mix's peak
    assert r["bound"] == ("hbm" if r["physical_frac"] >= r["valu_frac_of_mix"] else "issue") and 0.3 < r["traffic_over_algorithmic"] < 0.8
    assert "crl_stream_wait_mapped" in rec["config"]["completion"]
    assert r["achieved"] == pytest.approx(r["bytes_per_env_step"] * 65536 * 20 / (r["launch_ms"] * 1e-3) / 1e9, rel=1e-3)
    assert 300 < r["box_clock_mhz"] < 4000 and 0.1 < r["box_issue_vs_calibration"] < 1.5        # the box's own issue probe
    assert "RCCL" not in err and "NCCL version" not in err      # the contract run creates no process group


def test_bench_self_launch_runs_ranks_under_rccl(run_fresh):
    """`--gpus N` without a launcher starts N ranks under torch.distributed.run as a child process (CRL_BENCH_SELF_LAUNCH=1
    forces that path at N = 1 on this one-GPU box): one line, n_gpus = the ranks RCCL saw, the gather inside the region."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--only-headline",
           "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "LOCAL_RANK")}
    rc, out, err = run_fresh(cmd, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0", CRL_BENCH_SELF_LAUNCH="1"), cwd=ROOT, timeout=600, split=True)
    assert rc == 0, err[-3000:]
    rec = _one_json_line(out)
    assert rec["n_gpus"] == 1 and rec["config"]["gather"].startswith("rccl") and rec["config"]["parallelism"] == "dp1"
    assert "torch.distributed.run" in err
    # a record with a process group is attributable: what the collective costs here, the value without it, each rank's time
    c = rec["collective"]
    assert rec["gather_us"] == c["gather_us"] and 0 < c["gather_us"] < 200 and 0 < c["all_gather_us"] < 200
    assert c["region_gather_us"] > c["region_no_gather_us"] > 10 and len(c["elapsed_ranks_us"]) == 1 and c["row_bytes"] == 16
    assert rec["value_without_gather"] == pytest.approx(65536 * 20 / (c["region_no_gather_us"] * 1e-6), rel=1e-3)
    assert "crl_stream_wait_mapped" in rec["config"]["completion"]      # the same completion at every world size


def test_bench_gpus_2_on_a_one_gpu_box_fails(run_fresh):
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with one GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "LOCAL_RANK")}
    rc, out, err = run_fresh([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                             env=env, cwd=ROOT, timeout=300, split=True)
    assert rc != 0
    rec = _one_json_line(out)
    assert "value" not in rec and "--gpus 2" in rec["error"]
