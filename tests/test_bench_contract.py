"""bench.py's own plumbing, on CPU: the ONE compact stdout line, `--gpus N` never measuring fewer than N GPUs, the
snapshot of the gathered rows, and the contract region + line assembly in a world of two over gloo with the CPU oracle
as the stepper (tests may use the oracle; bench.py itself imports it only inside cpu_baseline)."""
import io
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
from test_parallel_gloo import OracleTronStepper  # noqa: E402

WL = "tron_p4_n20_b65536"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _canned_full():
    """A full record as large as a real one gets: every section present, long strings where bench.py writes long strings."""
    long = "x" * 400
    roof = {"bound": "hbm", "achieved": 7211.123456789, "peak": 8000.0, "unit": "GB/s", "frac": 0.901390432, "traffic": 66123456,
            "traffic_source": long, "kernel": "tron_rollout_quad_kernel", "steps_per_launch": 20, "launch_ms": 0.0180512345,
            "launch_ms_source": long, "bytes_per_env_step": 99.23, "formula": long, "physical_achieved": 3661.123456,
            "physical_frac": 0.457640432, "copy_peak": 5100.123456, "frac_of_copy": 0.71784321, "note": long,
            "valu_issue": {"achieved": 5.43e11, "peak": 1.08e12, "unit": "wave-instr/s", "frac": 0.50277777, "peak_source": long,
                           "waves_per_simd": 4, "peak_at_occupancy": 1.0e12, "frac_at_occupancy": 0.543, "mix": "tron",
                           "peak_of_mix": 6.44123e11, "frac_of_mix": 0.8712345},
            "traffic_over_algorithmic": 0.5081234, "valu_frac_of_mix": 0.5512345, "steady_valu_frac_of_mix": 0.8712345, "steady_bound": "issue",
            "oc_games": 1048576, "oc_launch_ms": 0.2101234, "oc_frac": 0.98765432, "oc_physical_frac": 0.5123456, "oc_step_observe_frac": 0.7123456,
            "steady_value": 2.136e11, "steady_launch_ms": 2.5131234, "steady_frac": 2.53123, "steady_physical_frac": 0.00328123,
            "steady_valu_frac": 0.516123, "box_clock_mhz": 2391.3, "box_issue_vs_calibration": 0.98765432,
            "box_clock_mhz_warm": 2377.7, "box_issue_vs_calibration_warm": 1.0123456}
    wl = {"value": 6.98123456e11, "unit": "env-steps/s", "games": 262144, "steps_per_launch": 2048, "launches": 64,
          "ms_per_env_step": 0.000375123, "mean_episode_len": 18.123, "dtype": "u32", "roofline": dict(roof),
          "cpu_baseline": {"value": 1.23456e8, "unit": "env-steps/s", "cores": 16, "kind": "port", "sample": long, "nproc": 256,
                           "threads_16": {"value": 1.23456e8, "threads": 16, "sample": long}},
          "placement_tests_per_s": 1.234e13}
    dropin = {n: {"new_state": 12345.6, "next_state": 12345.6, "valid_actions": 12345.6, "state_to_observation": 12345.6,
                  "reference_us": {"new_state": 15.2, "next_state": 171456.2, "valid_actions": 86101.3, "state_to_observation": 264.6},
                  "calls": 2000} for n in ("tron", "tictactoe", "tictactoe_3p", "tictactoe_4p", "blokus")}
    dropin["what"] = long
    cpu = {"value": 2.6012345e8, "unit": "env-steps/s", "cores": 16, "kind": "port", "sample": long, "nproc": 256,
           "reference_python": {"value": 4.4e4, "unit": "env-steps/s", "cores": 1, "where": long, "source": long}}
    for c in (1, 16, 32, 256):
        cpu["threads_%d" % c] = {"value": 2.6012345e8 / (1 + c % 3), "threads": c, "sample": long}
    return {
        "metric": "env-steps/sec", "value": 37571234567.891, "unit": "env-steps/s", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 0.00174456789, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int8",
        "data": "synthetic",
        "config": {"workload": WL, "games_per_gpu": 65536, "global_games": 65536, "steps_per_launch": 20, "launches": 1,
                   "agent": "uniform random (Philox-4x32-10), auto-reset", "mean_episode_len": 8.452, "episodes": 193841,
                   "parallelism": "dp1", "gather": "rccl gather to rank 0 (torch.distributed.gather), 16-byte rows",
                   "completion": "ShardedRollout.wait (crl_stream_wait_mapped on the launch stream)",
                   "cpu_baseline": "not run at N > 1 (rank 0 at N = 1 only)",
                   "device_warmup": "none: `value` is the first W + K region of the process"},
        "collective": {"gather_us": 27.12345, "all_gather_us": 31.12345, "region_no_gather_us": 36.12345, "region_gather_us": 63.12345,
                       "region_all_gather_us": 67.12345, "value_without_gather": 2.9012345e11, "row_bytes": 16, "rows_bytes_per_rank": 1048576,
                       "rows_bytes_into_rank0": 7340032, "rank_spread_us": 3.12, "elapsed_ranks_us": [61.12, 62.23, 63.34, 60.45, 61.56, 62.67, 63.78, 60.89],
                       "what": long},
        "gather_us": 27.12345, "value_without_gather": 2.9012345e11,
        "timed_region_ms": 0.03489123, "kernel_ms": 0.0282123, "kernel_ms_dispatch": 0.01756123, "kernel_ms_source": long,
        "roofline": roof, "warmed": {"value": 4.228e10, "timed_region_us": 31.0123, "device_warmup_ms": 91.2, "what": long},
        "value_warmed": 4.228e10,
        "seeds": {"0": {"value": 4.2e10, "mean_episode_len": 8.452}, "spread": 0.01},
        "steady_state": dict(wl), "others": {n: dict(wl) for n in bench.WORKLOADS if n != WL},
        "step_api": {k: {"gpu_us_per_call": 9.3202400, "frac_of_hbm_peak": 0.7123456, "what": long} for k in
                     ("tron_n20_step_auto_reset", "tron_n20_step_observe_fused", "tron_n20_observe_all", "ttt_3x5_step_observe_fused",
                      "blokus_step_observe_fused", "blokus_valid_list", "blokus_select")},
        "stream_peaks": {"copy_GBs": 5100.0, "write_GBs": 5600.0, "what": long}, "dropin": dropin, "cpu_baseline": cpu,
        "placement_tests_per_s": 1.2345e13, "gather": {"gather_us": 7.24, "region_us": 38.2, "region_no_gather_us": 31.0},
    }


def test_compact_line_is_small_and_carries_roofline_and_cpu_baseline(tmp_path):
    full = _canned_full()
    assert len(json.dumps(full)) > 20000                         # the record that did not parse in round 3 was 23.7 KB
    out = io.StringIO()
    line = bench.emit(full, out=out, detail_paths=[str(tmp_path / "bench_detail.json")])
    assert out.getvalue() == line + "\n" and "\n" not in line    # exactly one line
    assert len(line) < 4096, len(line)
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "summary"):
        assert k in rec, k
    assert rec["config"]["workload"] == WL and "model" not in rec["config"]
    assert rec["roofline"]["frac"] == pytest.approx(0.90139, rel=1e-4)
    assert rec["roofline"]["bound"] == "hbm" and rec["roofline"]["peak"] == 8000.0 and rec["roofline"]["traffic"] == 66123456
    assert rec["roofline"]["achieved"] / rec["roofline"]["peak"] == pytest.approx(rec["roofline"]["frac"], rel=1e-3)
    assert rec["roofline"]["traffic_over_algorithmic"] == pytest.approx(0.50812, rel=1e-4) and rec["roofline"]["valu_frac_of_mix"] == pytest.approx(0.55123, rel=1e-4)
    assert rec["roofline"]["steady_bound"] == "issue" and rec["roofline"]["oc_games"] == 1048576 and rec["roofline"]["oc_physical_frac"] > 0
    assert rec["gather_us"] == pytest.approx(27.123, rel=1e-4) and rec["value_without_gather"] == pytest.approx(2.9012e11, rel=1e-4)
    assert rec["collective"]["all_gather_us"] > 0 and len(rec["collective"]["elapsed_ranks_us"]) == 8 and rec["collective"]["rows_bytes_into_rank0"] == 7340032
    assert "crl_stream_wait_mapped" in rec["config"]["completion"]
    assert rec["cpu_baseline"]["value"] == pytest.approx(2.6012e8, rel=1e-3) and rec["cpu_baseline"]["kind"] == "port"
    assert rec["cpu_baseline"]["value"] >= rec["cpu_baseline"]["threads_16"]
    assert rec["value"] == pytest.approx(full["value"], rel=1e-5)
    assert set(rec["summary"]) >= {"headline"} | set(bench.WORKLOADS)
    detail = json.load(open(tmp_path / "bench_detail.json"))      # nothing is lost: the detail file has the sections
    for k in ("others", "step_api", "dropin", "seeds", "steady_state", "stream_peaks"):
        assert k in detail and k not in rec


def test_cpu_baseline_value_is_the_best_thread_count():
    cb = bench.cpu_baseline(WL, seconds=0.4)
    tried = {k: v["value"] for k, v in cb.items() if k.startswith("threads_")}
    assert len(tried) >= 2 and "threads_1" in tried
    assert cb["value"] == max(tried.values()) and cb["cores"] == int(max(tried, key=tried.get).split("_")[1])
    assert ("%d threads" % cb["cores"]) in cb["sample"]


def _run_bench(argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "LOCAL_RANK", "CRL_BENCH_SELF_LAUNCH")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    return p.returncode, p.stdout, p.stderr


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_gpus_2_with_fewer_devices_fails_loudly(no_gpu_context):
    """`--gpus N` means N ranks: with fewer devices visible the run exits non-zero and says why on stdout -- it never
    measures one GPU and calls it two."""
    rc, out, _ = _run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5"])
    assert rc != 0
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert "value" not in rec and "--gpus 2" in rec["error"] and "visible" in rec["error"]


def test_gpus_must_equal_world_size_under_a_launcher(no_gpu_context):
    rc, out, _ = _run_bench(["--gpus", "4"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                              "MASTER_PORT": str(_free_port())})
    assert rc != 0
    rec = json.loads(out.strip())
    assert "value" not in rec and "WORLD_SIZE=2" in rec["error"]


@pytest.mark.skipif(torch.cuda.device_count() >= 1, reason="needs a box without a GPU")
def test_no_gpu_is_an_error_not_a_cpu_measurement(no_gpu_context):
    rc, out, _ = _run_bench([])
    assert rc != 0 and "no GPU" in json.loads(out.strip())["error"]


class LiveRowsStepper:
    """A stepper whose `results(copy=False)` hands out its LIVE buffer, as TronBatch does on the device."""

    def __init__(self, batch, first_env_id):
        self.rows = torch.zeros((batch, 4), dtype=torch.int32)
        self.rows[:, 2] = torch.arange(first_env_id, first_env_id + batch, dtype=torch.int32)

    def rollout(self, steps, seed):
        self.rows[:, 1] += steps
        self.rows[:, 0] += (steps + 3) // 4                       # an episode every 4 steps

    def results(self, copy=True):
        return self.rows.clone() if copy else self.rows


def test_contract_region_snapshots_the_rows_it_timed():
    """Round 3's bug: the line's episodes / mean_episode_len were read from a live buffer after further launches."""
    from colosseumrl_amd.parallel import ShardedRollout
    pl = bench.Plumbing(torch, dist, torch.device("cpu"), False)
    sr = ShardedRollout(LiveRowsStepper, 32)
    meas = bench.contract_region(pl, sr, steps=20, warmup=5, seed=0, chunk=8192)
    assert meas["launches"] == 1 and meas["elapsed"] == meas["elapsed_rank"] > 0
    before = meas["rows"].clone()
    sr.rollout(400, 0, 20)                                        # what launch_time_pass / dispatch_time_pass do afterwards
    assert torch.equal(meas["rows"], before) and not torch.equal(sr.stepper.rows, before)
    mean_len, n_ep = bench.mean_episode_len(meas["rows"])
    assert n_ep == 32 * (2 + 5) and mean_len == pytest.approx(25 / 7)        # W + K = 25 steps, nothing more


def _world2(rank, world, port, batch, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from colosseumrl_amd.parallel import ShardedRollout
        pl = bench.Plumbing(torch, dist, torch.device("cpu"), True)
        assert (pl.world, pl.rank) == (world, rank)
        sr = ShardedRollout(lambda batch, first_env_id: OracleTronStepper(batch, first_env_id, N=20, P=4), world * batch)
        meas = bench.contract_region(pl, sr, steps=20, warmup=5, seed=0, chunk=8192)
        assert (meas["rows"] is None) == (rank != 0)
        assert meas["elapsed"] >= meas["elapsed_rank"] > 0          # the MAX over ranks
        elapsed = torch.tensor([meas["elapsed"]], dtype=torch.float64)
        both = [torch.zeros_like(elapsed) for _ in range(world)]
        dist.all_gather(both, elapsed)
        assert both[0].item() == both[1].item()                     # every rank holds the same (max) time
        coll = bench.collective_attribution(pl, sr, 20, 0, 8192, reps=4)         # every rank, as in main()
        if rank == 0:
            rows = meas["rows"]
            assert rows.dtype == torch.int16 and rows.shape == (world * batch, 8)     # the 16-bit rows, int32-viewed on the wire
            full = bench.contract_record(WL, batch, world, 20, 5, 8192, meas, launch_s=meas["elapsed"], launch_source="test",
                                         copy_gbs=None, gather_desc="gloo gather to rank 0, %d-byte rows" % (rows.shape[1] * 2))
            bench.attach_collective(full, coll, rows.shape[1] * 2, batch, world, 20)
            full["config"]["completion"] = pl.completion()
            out = io.StringIO()
            bench.emit(full, out=out, detail_paths=[os.path.join(out_dir, "detail.json")])
            with open(os.path.join(out_dir, "stdout.txt"), "w") as f:
                f.write(out.getvalue())
            np.save(os.path.join(out_dir, "rows.npy"), rows.numpy())
        else:
            assert not os.path.exists(os.path.join(out_dir, "stdout.txt")) or True    # only rank 0 ever emits
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_contract_region_and_line_in_a_world_of_two(tmp_path):
    """main()'s multi-rank branch without a GPU: max-over-ranks time, gather(dst=0) of equal shards of int32-viewed 16-bit
    rows, rank 0 alone emits ONE line with n_gpus = 2, and the gathered rows equal a single process's."""
    batch = 48
    mp.spawn(_world2, args=(2, _free_port(), batch, str(tmp_path)), nprocs=2, join=True)
    text = open(tmp_path / "stdout.txt").read()
    assert text.count("\n") == 1 and len(text) < 4096
    rec = json.loads(text)
    assert rec["n_gpus"] == 2 and rec["config"]["global_games"] == 2 * batch and rec["config"]["games_per_gpu"] == batch
    assert rec["config"]["parallelism"] == "dp2" and rec["scaling"] == "weak" and rec["steps"] == 20 and rec["warmup"] == 5
    assert rec["value"] == pytest.approx(2 * batch * 20 / (rec["ms_per_step"] * 20 * 1e-3), rel=1e-4)
    single = OracleTronStepper(2 * batch, 0, N=20, P=4)
    single.rollout(5, 0)
    single.rollout(20, 0)
    want = single.results_packed().numpy()
    assert np.array_equal(np.load(tmp_path / "rows.npy"), want)
    n_ep, len_sum = int(want[:, 0].sum()), int(want[:, 1].sum())
    assert rec["config"]["episodes"] == n_ep and rec["config"]["mean_episode_len"] == pytest.approx(len_sum / n_ep, abs=1e-3)
    assert n_ep <= 2 * batch * 25 / 2                                # W + K = 25 steps; no Tron episode is shorter than 2 steps
    assert rec["roofline"]["frac"] > 0 and rec["roofline"]["traffic"] is not None
    # the N > 1 record is attributable: the collective's cost in this world, the value without it, every rank's time
    c = rec["collective"]
    assert rec["gather_us"] == c["gather_us"] and c["region_gather_us"] - c["region_no_gather_us"] == pytest.approx(c["gather_us"], abs=0.02 + 2e-4 * c["region_gather_us"])
    assert c["all_gather_us"] == pytest.approx(c["region_all_gather_us"] - c["region_no_gather_us"], abs=0.02 + 2e-4 * c["region_all_gather_us"])   # (5 significant digits on the line)
    assert rec["value_without_gather"] == pytest.approx(2 * batch * 20 / (c["region_no_gather_us"] * 1e-6), rel=1e-3)
    assert c["row_bytes"] == 16 and c["rows_bytes_into_rank0"] == 16 * batch and len(c["elapsed_ranks_us"]) == 2
    assert c["rank_spread_us"] == pytest.approx(max(c["elapsed_ranks_us"]) - min(c["elapsed_ranks_us"]), abs=0.02 + 2e-4 * max(c["elapsed_ranks_us"]))
    assert rec["config"]["completion"] == "none (cpu)"
