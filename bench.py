#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched random-agent rollout (BASELINE.json metric) + the evidence around it.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

HEADLINE (the contract line).  One "step" = one env-step of EVERY game of the batch (Tron: all live players move).
The workload is BASELINE.json configs[1]: 4-player 20x20 Tron, 65,536 games per GPU, uniform random agent drawn from
the counter-based RNG, auto-reset on terminal.  State is resident in HBM before the timed region; the timed region is W
untimed + exactly K timed steps, issued as fused launches of min(--chunk, remaining) steps, bracketed by barrier +
synchronize; the time is the max over ranks.  `value` is the FIRST such region of the process: nothing runs between
set-up and W (the device-warmed figure is `value_warmed` in the detail record).  With N > 1 each rank owns the games
[rank*B, (rank+1)*B) (weak scaling, no data-path collective) and the per-game result rows (written by the rollout
kernel itself) are gathered once to rank 0 with a single RCCL gather at the end, inside the timed region
(colosseumrl_amd.parallel.ShardedRollout; the collective also runs in a world of one rank when a process group exists,
e.g. under `torchrun --nproc-per-node 1`).  The clock of a rank stops when its own work incl. the collective is
complete; the value uses the MAX over ranks.

`--gpus N` ALWAYS means N ranks: under a launcher WORLD_SIZE must equal N (else exit 2); without one and N > 1 this
script starts `python -m torch.distributed.run --nproc-per-node N` on itself as a CHILD process before anything has
touched the GPU, passes the child's stdout through and exits with its code; with fewer than N devices visible it exits
2 and says so on stdout.  It never measures fewer GPUs than it was asked for.

OUTPUT.  stdout carries exactly ONE line: the compact contract object (< 4 KB: the contract fields, `roofline`,
`cpu_baseline`, `summary`).  Everything else (N = 1, unless --only-headline) goes to bench_detail.json next to this
script (and gpurun_out/bench_detail.json when that directory exists) and, as one line, to stderr:
  roofline      the contract's formula: `achieved` = SURVEY 8(d)'s algorithmic bytes per env-step x the env-steps of one
                launch / the launch's duration (HIP events attached to the dispatch; the marker-event figure beside it),
                `frac` = / 8 TB/s; `traffic` = the HBM bytes the launch really moved (rocprofv3 PMC of the SAME launch
                shape, profiles/traffic_*.json; a documented model when no PMC pass exists for the shape) with
                `physical_achieved` / `physical_frac` / `frac_of_copy` (vs the device-copy bandwidth measured in this
                run) from it.  The fused kernels keep state in LDS across a launch, so on LONG launches the algorithmic
                figure exceeds the physical one and 1.0 (noted in the object): their roof is instruction issue --
                `valu_issue` = PMC SQ_INSTS_VALU of that launch shape / launch time against the MEASURED issue peak
                (tools/ubench/valu_rate.hip -> profiles/r*_issue_calibration.json)
  warmed        the same W + K region again after ~90 ms of device warm-up on a scratch stepper (`value_warmed`)
  steady_state  the same workload in long launches (the regime a rollout worker lives in)
  seeds         the headline region repeated for seeds {0, 1, 2}
  others        the other BASELINE workloads (TicTacToe 5x5 / 3x5 / 3x3x3, Blokus, the Tron 40x40 shard of config 5)
  step_api      the per-step batched API: sample / step(auto_reset) / observe_all / fused step+observe, TicTacToe and
                Blokus step / valid / observe -- what replaces next_state + state_to_observation in a learner loop
  cpu_baseline  the C oracle on this box's host cores at several thread counts (`value` = the best of them, `cores` =
                its thread count), nproc stated, plus the reference's own Python+Cython path as timed in the build
                container (it cannot run here)
  gather        (--gather-probe) the cost of the end-of-rollout collective in a one-rank RCCL group
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s HBM3E spec (6.29 TB/s measured float4 copy)
# fallback issue peak when no calibration file is present: one wave64 VALU instruction per 2 cycles per SIMD-32
# (MI355X_MICROARCH.md "v_fma_f32 (wave64) 2 cyc"), 256 CUs x 4 SIMDs at 2.4 GHz
VALU_PEAK_MODEL = 256 * 4 * 2.4e9 / 2


def _load_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def newest_calibration():
    """profiles/r<N>_*issue_calibration.json of the highest round (tools/ubench/valu_rate.hip), or None."""
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_*issue_calibration.json")):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    return best[1] if best else None


CALIBRATION = newest_calibration()

WORKLOADS = {
    # name: (game, kwargs, per-GPU batch, env-steps fused into one launch by default, resident waves per SIMD)
    "tron_p4_n20_b65536": ("tron", dict(board_size=20, num_players=4), 65536, 8192, 4),   # one lane per player: 4 waves/SIMD
    "tron_p4_n40_b65536": ("tron", dict(board_size=40, num_players=4), 65536, 8192, 4),   # one lane per player, bitboards
    "ttt_p3_5x5_k4_b262144": ("ttt", dict(dims=(5, 5), k=4, num_players=3), 262144, 2048, 4),
    "ttt_p3_3x5_k3_b262144": ("ttt", dict(dims=(3, 5), k=3, num_players=3), 262144, 2048, 4),
    "ttt_p4_3x3x3_b262144": ("ttt", dict(dims=(3, 3, 3), k=3, num_players=4), 262144, 2048, 4),
    "blokus_p4_b16384": ("blokus", dict(), 16384, 2048, 8),
}
HEADLINE = "tron_p4_n20_b65536"
# the end-of-rollout collective of the contract region: "gather" (to rank 0: seven shards over seven xGMI links at once) or
# "all_gather" (every rank gets everything: a ring).  CRL_BENCH_COLLECTIVE overrides; both are timed in every N > 1 record
# (`collective` section) and in a one-rank RCCL group by --gather-probe (profiles/r5_gather_probe.json).
COLLECTIVE = os.environ.get("CRL_BENCH_COLLECTIVE", "gather")


def reference_python(workload):
    """The reference's own CPU path (Python + Cython / scipy, one core) on this workload's game, as timed in the build
    container by tools/time_reference_rollout.py -> profiles/reference_python.json (it cannot travel to the GPU box).  The
    record of that file for the workload + where / jit / versions / script, or None when the file or the workload is
    absent -- there are no constants to fall back to."""
    rj = _load_json(os.path.join(ROOT, "profiles", "reference_python.json"))
    rec = ((rj or {}).get("workloads") or {}).get(workload)
    if not rec:
        return None
    return {"value": rec["value"], "unit": "env-steps/s", "cores": rj.get("cores_used", 1), "nproc": rj.get("nproc"),
            "mean_episode_len": rec.get("mean_episode_len"), "sample": "%d env-steps in %.1f s" % (rec["steps"], rec["seconds"]),
            "where": rj.get("where"), "jit": rj.get("jit"), "versions": rj.get("versions"),
            "source": "profiles/reference_python.json (%s)" % rj.get("script", "tools/time_reference_rollout.py")}


def algorithmic_bytes_per_step(game, kw, mean_len):
    """bytes one env-step has to move with the SoA layout, reset amortised."""
    if game == "tron":
        P, N = kw["num_players"], kw["board_size"]
        return (12 * P + 2) + (N * N + 4 * P) / max(mean_len, 1.0)
    if game == "ttt":
        cells = 1
        for d in kw["dims"]:
            cells *= d
        return cells + 1 + 1 + 4 + 3 + (cells + 7) // 8 + cells / max(mean_len, 1.0)
    if game == "blokus":
        return 400 + 5 + 16 + 8 + 4 + 4 + 6 + (400 + 32) / max(mean_len, 1.0)
    raise ValueError(game)


def launch_traffic_model(game, kw):
    """HBM bytes per game one fused rollout LAUNCH has to move (state in + out once, statistics read-modify-write,
    the packed result row), whatever the number of steps: the state lives in LDS / registers in between."""
    if game == "tron":
        P, N = kw["num_players"], kw["board_size"]
        state = N * N + 4 * P
        stats = 2 * (4 + 4 + 4 + 4 + 4 * P + 4 * P) + 3 + 4 * (3 + 2 * P)
        return 2 * state + stats
    if game == "ttt":
        P = kw["num_players"]
        return 2 * (4 * P + 2) + 2 * (4 * 4 + 4 * P + 4) + 4 * (3 + P)
    if game == "blokus":
        return 2 * 360 + 2 * 8 + 4 * 10
    raise ValueError(game)


def kernel_name(game, kw, steps_per_launch):
    """The dominant kernel of the workload (what the rocprof summaries under profiles/ list)."""
    if game == "tron":
        n = kw["board_size"]                                     # csrc/tron.hip, crl_tron_rollout's choice
        if n <= 20:
            return "tron_rollout_quad_kernel" if kw["num_players"] <= 4 else "tron_rollout_lds_kernel"
        if n <= 40 and kw["num_players"] <= 4:
            return "tron_rollout_qbits_kernel"
        if n <= 40:
            return "tron_rollout_bits_kernel" if steps_per_launch >= 256 else "tron_rollout_lds_kernel"
        return "tron_rollout_kernel"
    return "%s_rollout_kernel" % game


def make_stepper(game, kw, batch, device, first_env_id):
    from colosseumrl_amd import batched
    if game == "tron":
        return batched.TronBatch(batch=batch, device=device, first_env_id=first_env_id, **kw)
    if game == "ttt":
        return batched.TTTBatch(batch=batch, device=device, first_env_id=first_env_id, **kw)
    if game == "blokus":
        return batched.BlokusBatch(batch=batch, device=device, first_env_id=first_env_id, **kw)
    raise ValueError(game)


# ---------------------------------------------------------------------------------------------- CPU baseline
def cpu_thread_counts(nproc, quick=False):
    """Thread counts the CPU baseline is timed at: the 16 cores that are a 1-GPU box's share (CRL_CPU_THREADS overrides),
    every hardware thread the box shows, 32 in between, and one."""
    share = max(1, min(nproc, int(os.environ.get("CRL_CPU_THREADS", "16"))))
    counts = [share, nproc] if quick else [share, nproc, min(nproc, 2 * share), 1]
    out = []
    for c in counts:
        if c not in out:
            out.append(c)
    return out


def cpu_baseline(workload, seconds=6.0, quick=False):
    """Time the CPU oracle (bit-exact C restatement, oracle/) on this box's host cores on a bounded sample at several
    thread counts (`cpu_thread_counts`).  `value` = the BEST of them and `cores` = the threads that run used; every
    count tried is listed as `threads_<n>`; plus the reference's own Python figure."""
    from oracle import oracle as O
    game, kw = WORKLOADS[workload][:2]
    nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    extra = {}

    def runner(B, cores):
        if game == "tron":
            N, P = kw["board_size"], kw["num_players"]
            sh, sd = O.tron_start_positions(N, P)

            def run(T):
                st = O.TronState(N, P, B)
                O.tron_reset(st, sh, sd)
                t0 = time.perf_counter()
                O.tron_rollout(st, 1, 0, T, sh, sd, n_threads=cores)
                return time.perf_counter() - t0
        elif game == "ttt":
            def run(T):
                st = O.TTTState(kw["dims"], kw["k"], kw["num_players"], B)
                t0 = time.perf_counter()
                O.ttt_rollout(st, 1, 0, T, n_threads=cores)
                return time.perf_counter() - t0
        else:
            def run(T):
                st = O.BlokusState(B)
                O.blokus_placement_tests(reset=True)
                t0 = time.perf_counter()
                O.blokus_rollout(st, 1, 0, T, n_threads=cores)
                dt = time.perf_counter() - t0
                extra["placement_tests_per_env_step"] = O.blokus_placement_tests() / float(B * T)
                return dt
        return run

    def timed(cores, seconds):
        B = (64 * cores) if game == "blokus" else (max(65536, 512 * cores) if cores > 1 else 8192)
        run = runner(B, cores)
        T = 4
        dt = run(T)                                # warms the thread pool
        while dt < 0.3 and T < (1 << 18):           # calibrate on a run long enough to be meaningful
            T *= 4
            dt = run(T)
        rate = B * T / max(dt, 1e-9)
        T = int(max(4, min(1 << 20, rate * seconds / B)))
        dt = run(T)
        return {"value": B * T / dt, "threads": cores,
                "sample": "%d games x %d steps, oracle/liboracle.so (C, OpenMP over games), %d threads, %.1f s" % (B, T, cores, dt)}

    counts = cpu_thread_counts(nproc, quick)
    runs = {c: timed(c, seconds if i == 0 else max(2.0, seconds / 2)) for i, c in enumerate(counts)}
    best = max(runs.values(), key=lambda r: r["value"])
    out = {"value": best["value"], "unit": "env-steps/s", "cores": best["threads"], "kind": "port", "sample": best["sample"],
           "nproc": nproc}
    for c, r in runs.items():
        out["threads_%d" % c] = r
    out["reference_python"] = reference_python(workload) or "absent (profiles/reference_python.json has no record for this workload)"
    out.update(extra)
    return out


# ---------------------------------------------------------------------------------------------- roofline
def pmc_for(workload, steps_per_launch):
    """PMC record of this workload's dominant kernel for EXACTLY this launch shape, or None."""
    tj = _load_json(os.path.join(ROOT, "profiles", "traffic_%s.json" % workload))
    if not tj:
        return None
    shapes = tj.get("launch_shapes")
    if shapes is None:                                   # round-1 file format: one shape per file
        shapes = {str(tj.get("steps_per_launch")): tj}
    return shapes.get(str(int(steps_per_launch)))


def valu_peaks(waves_per_simd, mix="valu"):
    """(chip VALU issue peak, peak at this occupancy, source) in wave64 VALU instructions / s from the calibration ubench;
    `mix` = "valu" (independent integer VALU only: the pure peak) or a game's instruction mix ("tron" / "ttt" / "blokus":
    VALU with that game's share of SALU, LDS and quarter-rate multiplies in the stream -- what THAT stream can reach)."""
    cal = _load_json(CALIBRATION) if CALIBRATION else None
    if not cal or mix not in cal.get("mixes", {}):
        if mix != "valu":
            return None, None, None
        return VALU_PEAK_MODEL, None, "model: 1 wave64 VALU / 2 cycles / SIMD at 2.4 GHz (no calibration file)"
    rows = {int(r["waves_per_simd"]): r for r in cal["mixes"][mix]}
    peak = max(r["valu_wave_insts_per_s"] for r in rows.values())
    w = max(k for k in rows if k <= max(1, waves_per_simd))
    return peak, rows[w]["valu_wave_insts_per_s"], "measured: profiles/%s (tools/ubench/valu_rate.hip)" % os.path.basename(CALIBRATION)


MIX_OF_GAME = {"tron": "tron", "ttt": "ttt", "blokus": "blokus"}


def roofline(workload, batch, steps_per_launch, launch_s, mean_len, copy_gbs, launch_source=None):
    """The contract's roofline object.  `achieved` / `frac` follow the task's formula: ALGORITHMIC bytes of one launch
    (SURVEY 8(d)'s per-env-step figure x the env-steps the launch processes) / the launch's duration, against 8 TB/s;
    `traffic` = the HBM bytes the launch really moved (PMC); `physical_*` = that traffic / the same duration.  A fused
    launch keeps its state in LDS / registers across its steps, so for long launches the algorithmic figure exceeds the
    physical one by about the number of fused steps (and 1.0): HBM is then not the kernel's roof, `valu_issue` is."""
    game, kw = WORKLOADS[workload][:2]
    wps = WORKLOADS[workload][4]
    alg_bps = algorithmic_bytes_per_step(game, kw, mean_len)
    alg_gbs = alg_bps * batch * steps_per_launch / launch_s / 1e9
    pmc = pmc_for(workload, steps_per_launch)
    if pmc and pmc.get("hbm_bytes_per_launch"):
        traffic, source = int(pmc["hbm_bytes_per_launch"]) * batch // int(pmc.get("games", batch)), "rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE), same launch shape: profiles/traffic_%s.json" % workload
    else:
        traffic, source = int(launch_traffic_model(game, kw) * batch), "model: state in + out once per launch + statistics (no PMC pass for %d-step launches)" % steps_per_launch
    phys = traffic / launch_s / 1e9
    r = {"bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbs / HBM_PEAK_GBS,
         "traffic": traffic, "traffic_source": source, "kernel": kernel_name(game, kw, steps_per_launch),
         "steps_per_launch": int(steps_per_launch), "launch_ms": launch_s * 1e3,
         "launch_ms_source": launch_source or "HIP events around back-to-back launches on the launch stream",
         "bytes_per_env_step": round(alg_bps, 2),
         "formula": "achieved = bytes_per_env_step (SURVEY 8(d): 12P+2 + (N*N+4P)/mean_episode_len for Tron) x games x steps_per_launch / launch_ms",
         "physical_achieved": phys, "physical_frac": phys / HBM_PEAK_GBS}
    if r["frac"] > 1.0:
        r["note"] = ("frac > 1: the fused launch keeps boards in LDS / registers across its %d steps, so the algorithmic bytes of a "
                     "step-at-a-time stepper are never moved; HBM is not this launch shape's roof -- see valu_issue and physical_frac" % steps_per_launch)
    if copy_gbs:
        r["copy_peak"] = copy_gbs
        r["frac_of_copy"] = phys / copy_gbs               # physical traffic against the device-copy ceiling of this run
    if pmc and pmc.get("valu_insts_per_launch"):
        peak, peak_occ, src = valu_peaks(wps)
        v = pmc["valu_insts_per_launch"] / launch_s
        vi = {"achieved": v, "peak": peak, "unit": "wave-instr/s", "frac": v / peak, "peak_source": src,
              "waves_per_simd": wps}
        if peak_occ:
            vi["peak_at_occupancy"] = peak_occ
            vi["frac_at_occupancy"] = v / peak_occ
        _, mix_occ, _ = valu_peaks(wps, MIX_OF_GAME[game])
        if mix_occ:                                        # against what this game's instruction MIX issues at this occupancy
            vi["mix"] = MIX_OF_GAME[game]
            vi["peak_of_mix"] = mix_occ
            vi["frac_of_mix"] = v / mix_occ
        r["valu_issue"] = vi
    # Which roof is nearer: the launch's PHYSICAL HBM traffic / time against 8 TB/s, or its VALU instructions / time against
    # what its instruction mix can issue.  "hbm" only where the memory fraction is the larger one (`frac`, the contract's
    # algorithmic-bytes formula, stays as it is: it credits bytes a fused launch never moves).
    issue = (r.get("valu_issue") or {}).get("frac_of_mix") or (r.get("valu_issue") or {}).get("frac")
    r["bound"] = "hbm" if issue is None or r["physical_frac"] >= issue else "issue"
    r["traffic_over_algorithmic"] = traffic / max(alg_bps * batch * steps_per_launch, 1.0)
    return r


def measure_copy_bandwidth(torch, device, nbytes=1 << 30, reps=5):
    """On-box streaming ceilings over a 1 GiB buffer (HBM bytes moved / time; SURVEY 8d's second peak):
    `copy` = the best of the runtime's device-to-device copy and two vectorised read+write kernels (16 bytes per lane);
    `write` = a pure write stream (the per-step kernels write four times what they read, so a balanced copy alone is not
    their ceiling).  Returns (copy GB/s, write GB/s)."""
    src = torch.empty(nbytes // 4, dtype=torch.int32, device=device).fill_(1)
    dst = torch.empty_like(src)
    best = [0.0, 0.0]
    for which, op, moved in ((0, lambda: dst.copy_(src), 2.0 * nbytes), (0, lambda: torch.add(src, 1, out=dst), 2.0 * nbytes),
                             (0, lambda: torch.bitwise_xor(src, 1, out=dst), 2.0 * nbytes), (1, lambda: dst.fill_(3), 1.0 * nbytes)):
        op()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            op()
        e1.record()
        torch.cuda.synchronize()
        best[which] = max(best[which], moved / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9)
    del src, dst
    return best[0], best[1]


# ---------------------------------------------------------------------------------------------- measurements
def box_issue_probe(torch, device, waves_per_simd=4, target_s=3e-4):
    """What THIS box issues right now (crl_diag_issue_probe: independent integer VALU instructions on every SIMD, 4 waves
    each) and at which shader clock: {valu_wave_insts_per_s, clock_mhz, vs_calibration}.  The rollout kernels are bound by
    instruction issue, so a box whose clocks are capped runs them proportionally slower; `vs_calibration` = this box's rate /
    the committed calibration's at the same occupancy says so in the record."""
    from colosseumrl_amd import _native
    lib = _native.lib()
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    blocks = cus * waves_per_simd
    out = torch.empty((blocks * 256,), dtype=torch.int32, device=device)
    clk = torch.zeros((2,), dtype=torch.int64, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    iters = 256
    rate = None
    for _ in range(3):                                     # first pass sizes the run, the last one is reported
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.crl_diag_issue_probe(out.data_ptr(), clk.data_ptr(), blocks, iters, stream)
        e1.record()
        torch.cuda.synchronize()
        if rc:
            _native.check(rc, "crl_diag_issue_probe")
        secs = e0.elapsed_time(e1) * 1e-3
        rate = blocks * 4 * iters * 64 / max(secs, 1e-9)
        iters = int(max(64, min(1 << 20, iters * target_s / max(secs, 1e-6))))
    ticks = clk.cpu().tolist()
    rec = {"valu_wave_insts_per_s": rate, "clock_mhz": round(100.0 * ticks[0] / max(ticks[1], 1), 1), "waves_per_simd": waves_per_simd}
    _, peak_occ, _ = valu_peaks(waves_per_simd)
    if peak_occ:
        rec["vs_calibration"] = rate / peak_occ
    return rec


class Plumbing:
    """Device + process-group plumbing of the contract region, so that the same region runs on `cuda` over RCCL (the
    product) and on `cpu` over gloo (tests/test_bench_contract.py drives it with a CPU stepper in a world of two)."""

    def __init__(self, torch, dist, device, use_dist):
        self.torch, self.dist, self.device, self.use_dist = torch, dist, device, bool(use_dist)
        self.cuda = device.type == "cuda"
        self.world = dist.get_world_size() if self.use_dist else 1
        self.rank = dist.get_rank() if self.use_dist else 0

    def sync(self):
        if self.cuda:
            self.torch.cuda.synchronize()

    def complete(self, sr):
        """End of this rank's timed work: the product's own stream-scoped wait (`ShardedRollout.wait` -> `TronBatch.wait` ->
        crl_stream_wait_mapped: the host spins on a word in mapped memory that a one-thread kernel behind the queued work
        publishes; ~3.5 us cheaper than a device synchronise -- the same call a user of the batched steppers makes).  The
        same at every world size: a synchronous torch.distributed collective makes the launch stream wait for the
        communicator's stream, so the signal kernel behind it runs after the rows have arrived
        (tests/test_gpu_rccl.py checks that on data).  CRL_BENCH_SYNC=device falls back to torch.cuda.synchronize()."""
        if self.cuda and hasattr(sr, "wait") and os.environ.get("CRL_BENCH_SYNC") != "device":
            sr.wait()
        else:
            self.sync()

    def completion(self):
        return ("ShardedRollout.wait (crl_stream_wait_mapped on the launch stream)" if self.cuda and os.environ.get("CRL_BENCH_SYNC") != "device"
                else "torch.cuda.synchronize" if self.cuda else "none (cpu)")

    def barrier(self):
        self.sync()
        if self.use_dist:
            self.dist.barrier()
            self.sync()

    def max_over_ranks(self, values):
        if not self.use_dist:
            return [float(v) for v in values]
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]


def timed_rollout(pl, sr, steps, seed, chunk, events=None, dst=None, gather=True):
    """The contract's timed region: exactly `steps` env-steps + the gather of the per-game results, bracketed by
    barrier + synchronize on both sides.  The clock stops when THIS rank's work (incl. the collective, which itself waits
    for the other ranks' shards) has completed on the device; the closing barrier follows, and the caller takes the MAX of
    the per-rank times -- the time of the slowest rank, without the latency of the closing barrier itself.
    Returns (elapsed s, marker-event s or None, launches, SNAPSHOT of the gathered rows or None): `gather(copy=False)`
    hands out a live buffer (without a process group the stepper's own rows), so the rows are cloned right after the
    clock stops -- what the caller reads describes this region even if the stepper is advanced afterwards.
    `events` = (ev0, ev1) brackets the launches with HIP events on the launch stream (two hipEventRecord calls = 3.6 us
    of the region; ~10 us next to a collective: tools/debug/gather_latency.py); None leaves them out, and the launches
    are timed in a separate pass (`launch_time_pass`, `dispatch_time_pass`)."""
    pl.barrier()
    gc_on = gc.isenabled()
    gc.disable()                                           # as timeit does: a young-generation collection inside a ~35 us region is ~10 us of noise
    try:
        t0 = time.perf_counter()
        if events:
            events[0].record()                             # same stream the kernels are launched on
        launches = sr.rollout(steps, seed, chunk)
        if events:
            events[1].record()
        # the one collective: per-game results to rank `dst` (no host sync between the last launch and it; the reused
        # receive buffer).  gather=False (the attribution regions only, never `value`): the same region without it
        gathered = sr.gather(dst=dst, copy=False) if gather else None
        pl.complete(sr)
        elapsed = time.perf_counter() - t0
    finally:
        if gc_on:
            gc.enable()
    if gathered is not None:
        gathered = gathered.clone()                        # outside the clock
    pl.barrier()
    return elapsed, (events[0].elapsed_time(events[1]) * 1e-3 if events else None), launches, gathered


def contract_region(pl, sr, steps, warmup, seed, chunk, events=None):
    """W untimed + exactly K timed steps, as the contract says: the warm-up goes through the very function that is timed
    (same launch path, gather and barriers), so nothing in the timed region runs for the first time.  Every rank calls
    this; returns {elapsed (MAX over ranks), elapsed_rank, marker_s, launches, rows (rank 0: snapshot of all games' result
    rows, else None)}."""
    dst = (0 if pl.use_dist else None) if COLLECTIVE == "gather" else None    # the episode-end gather goes to rank 0
    if pl.use_dist and hasattr(sr, "warm_collective"):
        sr.warm_collective(dst)                            # set-up: communicator, channels, receive buffer (no stepping)
    if warmup > 0:
        timed_rollout(pl, sr, warmup, seed, chunk, events, dst)
    else:
        sr.gather(dst=dst, copy=False)                     # brings the communicator / receive buffer up outside the clock
    e, k, n, g = timed_rollout(pl, sr, steps, seed, chunk, events, dst)
    return {"elapsed": pl.max_over_ranks([e])[0], "elapsed_rank": e, "marker_s": k, "launches": n, "rows": g}


def collective_attribution(pl, sr, steps, seed, chunk, reps=9):
    """What the end-of-rollout collective costs in THIS world, so that an N > 1 line separates kernel scaling from
    collective latency: further regions of the contract's shape (barrier + synchronise on both sides, max over ranks per
    region, median over `reps`) without a collective, with `gather` to rank 0 and with `all_gather_into_tensor`.
    Every rank calls this.  {region_no_gather_us, region_gather_us, region_all_gather_us, gather_us, all_gather_us,
    elapsed_ranks_us (every rank's own time of the last gather region), rank_spread_us}."""
    def regions(dst, gather):
        ts = []
        for i in range(reps):
            e = timed_rollout(pl, sr, steps, seed, chunk, None, dst, gather)[0]
            ts.append(pl.max_over_ranks([e])[0])
            if hasattr(sr.stepper, "reset_stats") and (i + 1) * steps % 2000 < steps:
                sr.stepper.reset_stats()                  # keep the 16-bit rows exact (what a short region ships)
        ts = sorted(ts[1:])
        return ts[len(ts) // 2] * 1e6, e
    none_us, _ = regions(0, False)
    gather_us, e_rank = regions(0, True)
    all_us, _ = regions(None, True)
    if pl.use_dist:
        t = pl.torch.zeros((pl.world,), dtype=pl.torch.float64, device=pl.device)
        t[pl.rank] = e_rank * 1e6
        pl.dist.all_reduce(t, op=pl.dist.ReduceOp.SUM)
        ranks = [round(float(v), 2) for v in t.tolist()]
    else:
        ranks = [round(e_rank * 1e6, 2)]
    return {"region_no_gather_us": round(none_us, 2), "region_gather_us": round(gather_us, 2), "region_all_gather_us": round(all_us, 2),
            "gather_us": round(gather_us - none_us, 2), "all_gather_us": round(all_us - none_us, 2),
            "elapsed_ranks_us": ranks, "rank_spread_us": round(max(ranks) - min(ranks), 2),
            "what": "median of %d further regions of the timed shape each (max over ranks per region): no collective / gather to rank 0 / all_gather_into_tensor" % (reps - 1)}


def attach_collective(out, coll, row_bytes, batch, world, steps):
    """Put `collective_attribution`'s result into the record of an N-rank run: the `collective` section + the two scalars
    a reader of the line wants first, `gather_us` and `value_without_gather`."""
    if not coll:
        return
    coll.update({"row_bytes": row_bytes, "rows_bytes_per_rank": row_bytes * batch, "rows_bytes_into_rank0": row_bytes * batch * (world - 1),
                 "value_without_gather": world * batch * steps / (coll["region_no_gather_us"] * 1e-6),
                 "value_in_attribution_regions": world * batch * steps / (coll["region_%s_us" % COLLECTIVE] * 1e-6)})
    out["collective"] = coll
    out["gather_us"], out["value_without_gather"] = coll["%s_us" % COLLECTIVE], coll["value_without_gather"]


def dispatch_time_pass(torch, sr, steps, seed, chunk, events, reps=20):
    """The launches' own duration: HIP events attached to the first / last dispatch (crl_tron_rollout_timed; Tron only),
    median over `reps` isolated regions of the timed shape.  None for steppers without that entry point."""
    if not getattr(sr.stepper, "rollout_takes_events", False):
        return None
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        sr.rollout(steps, seed, chunk, events=events)
        torch.cuda.synchronize()
        ts.append(events[0].elapsed_time(events[1]) * 1e-3)
    return sorted(ts)[len(ts) // 2]


def launch_time_pass(torch, sr, steps, seed, chunk, events):
    """HIP-event time of the launches of one more region of the same shape (no gather), for runs whose contract region
    carries no events."""
    torch.cuda.synchronize()
    events[0].record()
    sr.rollout(steps, seed, chunk)
    events[1].record()
    torch.cuda.synchronize()
    return events[0].elapsed_time(events[1]) * 1e-3


def mean_episode_len(gathered):
    n_ep = int(gathered[..., 0].sum().item())
    return int(gathered[..., 1].sum().item()) / max(n_ep, 1), n_ep


def contract_record(workload, batch, world, steps, warmup, chunk, meas, launch_s, launch_source, copy_gbs, gather_desc):
    """The contract fields + `roofline` of one measured region (`meas` = contract_region's result on rank 0)."""
    game = WORKLOADS[workload][0]
    mean_len, n_ep = mean_episode_len(meas["rows"])
    steps_per_launch = min(chunk, steps)
    elapsed = meas["elapsed"]
    rec = {
        "metric": "env-steps/sec", "value": world * batch * steps / elapsed, "unit": "env-steps/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": elapsed * 1e3 / steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int8" if game != "ttt" else "u32", "data": "synthetic",
        "config": {"workload": workload, "games_per_gpu": batch, "global_games": world * batch,
                   "steps_per_launch": steps_per_launch, "launches": meas["launches"],
                   "agent": "uniform random (Philox-4x32-10), auto-reset",
                   "mean_episode_len": round(mean_len, 3), "episodes": n_ep, "parallelism": "dp%d" % world,
                   "gather": gather_desc,
                   "device_warmup": "none: `value` is the first W + K region of the process"},
        "timed_region_ms": elapsed * 1e3,
        "roofline": roofline(workload, batch, steps_per_launch, launch_s, mean_len, copy_gbs, launch_source),
    }
    if steps % steps_per_launch:
        # the roofline describes the dominant (first) launch shape; the mean launch time is an approximation then
        rec["roofline"]["note"] = "launches of the timed region are not equal-sized; launch_ms is their mean"
    return rec


def steady_state(torch, workload, device, seed, copy_gbs, target_s=0.25):
    """env-steps/s of a workload in its default long launches: one warm-up launch, then n equal launches (n chosen
    for ~target_s of GPU time), wall clock between synchronisations; roofline from the HIP-event launch time."""
    game, kw, batch, chunk = WORKLOADS[workload][:4]
    st = make_stepper(game, kw, batch, device, 0)
    st.rollout(chunk, seed)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    st.rollout(chunk, seed)
    e1.record()
    torch.cuda.synchronize()
    one = max(e0.elapsed_time(e1) * 1e-3, 1e-6)
    n = int(max(2, min(64, target_s / one)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        st.rollout(chunk, seed)
    e1.record()
    res = st.results(copy=False)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    mean_len = mean_episode_len(res)[0]
    launch_s = e0.elapsed_time(e1) * 1e-3 / n
    out = {"value": batch * chunk * n / elapsed, "unit": "env-steps/s", "games": batch, "steps_per_launch": chunk,
           "launches": n, "ms_per_env_step": elapsed * 1e3 / (chunk * n), "mean_episode_len": round(mean_len, 3),
           "dtype": "u32" if game == "ttt" else "int8",
           "roofline": roofline(workload, batch, chunk, launch_s, mean_len, copy_gbs)}
    del st
    return out


OUT_OF_CACHE_BATCH = 1 << 20       # Tron 20 x 20: 1,048,576 games x 416 B = 436 MB of state > the 256 MiB Infinity Cache


def out_of_cache(torch, device, seed, steps=20):
    """The Tron kernels on a batch whose state does NOT fit the Infinity Cache (B = 1,048,576: 436 MB; at BASELINE's
    65,536 games the 27 MB of state live in the 256 MiB MALL between launches, and the PMC's FETCH_SIZE counts those hits):
    the `steps`-step rollout launch (events attached to the dispatch, median of 9 isolated launches) and the fused
    step_observe call (events around 20 calls back to back), each with its algorithmic bytes, the PMC traffic of the
    same launch shape where a pass exists (profiles/traffic_tron_p4_n20_b1048576.json) and the fractions of 8 TB/s."""
    from colosseumrl_amd.batched import TronBatch
    B, N, P = OUT_OF_CACHE_BATCH, 20, 4
    tb = TronBatch(N, P, B, device=device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()
    torch.cuda.synchronize()
    tb.rollout(steps, seed)
    ts = []
    for i in range(9):
        torch.cuda.synchronize()
        tb.rollout(steps, seed, events=(e0, e1))
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    launch_s = sorted(ts)[len(ts) // 2]
    mean_len = mean_episode_len(tb.results(copy=False))[0]
    alg = algorithmic_bytes_per_step("tron", dict(num_players=P, board_size=N), mean_len) * B * steps
    kp = ((_load_json(os.path.join(ROOT, "profiles", "traffic_out_of_cache.json")) or {}).get("kernels") or {})
    pmc = kp.get("tron_rollout_quad_kernel<24>") or {}
    if pmc.get("steps_per_launch") not in (None, steps) or pmc.get("games") not in (None, B):
        pmc = {}                                            # a PMC record describes exactly one launch shape
    out = {"games": B, "state_bytes": B * (N * N + 4 * P), "infinity_cache_bytes": 256 << 20}
    r = {"kernel": "tron_rollout_quad_kernel", "steps_per_launch": steps, "launch_ms": launch_s * 1e3, "mean_episode_len": round(mean_len, 3),
         "value": B * steps / launch_s, "algorithmic_bytes": alg, "frac": alg / launch_s / 1e9 / HBM_PEAK_GBS,
         "traffic_model": int(launch_traffic_model("tron", dict(num_players=P, board_size=N)) * B)}
    r["traffic"] = int(pmc["hbm_bytes_per_call"]) if pmc.get("hbm_bytes_per_call") else None
    phys = r["traffic"] or r["traffic_model"]
    r["physical_frac"] = phys / launch_s / 1e9 / HBM_PEAK_GBS
    r["traffic_source"] = ("rocprofv3 PMC (2 x FETCH_SIZE + WRITE_SIZE), same launch shape: profiles/traffic_out_of_cache.json" if r["traffic"]
                           else "model: state in + out once + statistics (no PMC pass)")
    r["traffic_over_algorithmic"] = phys / alg
    if pmc.get("valu_insts_per_call"):
        _, mix_occ, _ = valu_peaks(4, "tron")
        if mix_occ:
            r["valu_frac_of_mix"] = pmc["valu_insts_per_call"] / launch_s / mix_occ
    r["bound"] = "hbm" if r["physical_frac"] >= (r.get("valu_frac_of_mix") or 0.0) else "issue"
    out["rollout"] = r
    fo = tb.step_observe(None, seed=7, out=None)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        tb.step_observe(None, seed=7, out=fo)
    e1.record()
    torch.cuda.synchronize()
    call_s = e0.elapsed_time(e1) * 1e-3 / 20
    nbytes = (1 + P) * N * N * B
    so = {"kernel": "tron_step_observe_kernel<4, 64>", "gpu_us_per_call": call_s * 1e6, "algorithmic_bytes": nbytes,
          "frac": nbytes / call_s / 1e9 / HBM_PEAK_GBS, "value": B / call_s,
          "traffic": (kp.get("tron_step_observe_kernel<4, 64>") or {}).get("hbm_bytes_per_call")}
    if so["traffic"]:
        so["physical_frac"] = so["traffic"] / call_s / 1e9 / HBM_PEAK_GBS
    out["step_observe"] = so
    acts = torch.randint(-1, 2, (P, B), dtype=torch.int8, device=device)
    tb.step(acts, auto_reset=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        tb.step(acts, auto_reset=True)
    e1.record()
    torch.cuda.synchronize()
    call_s = e0.elapsed_time(e1) * 1e-3 / 20
    st = {"kernel": "tron_step_kernel<4>", "gpu_us_per_call": call_s * 1e6, "algorithmic_bytes": (12 * P + 2) * B,
          "frac": (12 * P + 2) * B / call_s / 1e9 / HBM_PEAK_GBS, "value": B / call_s,
          "traffic": (kp.get("tron_step_kernel<4>") or {}).get("hbm_bytes_per_call")}
    if st["traffic"]:
        st["physical_frac"] = st["traffic"] / call_s / 1e9 / HBM_PEAK_GBS
    out["step"] = st
    del tb, fo, acts
    torch.cuda.empty_cache()
    # the other two games at batches whose state exceeds the Infinity Cache: short rollout launches (the state goes in and
    # out once per launch whatever its length), events around back-to-back launches
    from colosseumrl_amd.batched import BlokusBatch, TTTBatch
    for name, make, plies, kernel, game, kw, wps in (
            ("ttt_p3_3x5_k3", lambda: TTTBatch((3, 5), 3, 3, 20 * (1 << 20), device=device), 8, "ttt_rollout_kernel<3, 4, true>", "ttt",
             dict(dims=(3, 5), k=3, num_players=3), 4),
            ("blokus_p4", lambda: BlokusBatch(3 * (1 << 18), device=device), 8, "blokus_rollout_kernel", "blokus", dict(), 8)):
        sb = make()
        sb.rollout(plies, seed)
        torch.cuda.synchronize()
        n = 5
        e0.record()
        for _ in range(n):
            sb.rollout(plies, seed)
        e1.record()
        torch.cuda.synchronize()
        launch_s = e0.elapsed_time(e1) * 1e-3 / n
        mean_len = mean_episode_len(sb.results(copy=False))[0]
        state_b = sb.B * (4 * 3 + 2 if game == "ttt" else 360)
        alg = algorithmic_bytes_per_step(game, kw, mean_len) * sb.B * plies
        pmc = kp.get(kernel) or {}
        rec = {"kernel": kernel, "games": sb.B, "state_bytes": state_b, "steps_per_launch": plies, "launch_ms": launch_s * 1e3,
               "value": sb.B * plies / launch_s, "algorithmic_bytes": alg, "frac": alg / launch_s / 1e9 / HBM_PEAK_GBS,
               "traffic_model": int(launch_traffic_model(game, kw) * sb.B),
               "traffic": int(pmc["hbm_bytes_per_call"]) if pmc.get("hbm_bytes_per_call") and pmc.get("games") == sb.B else None}
        phys = rec["traffic"] or rec["traffic_model"]
        rec["physical_frac"] = phys / launch_s / 1e9 / HBM_PEAK_GBS
        if pmc.get("valu_insts_per_call") and pmc.get("games") == sb.B:
            _, mix_occ, _ = valu_peaks(wps, MIX_OF_GAME[game])
            if mix_occ:
                rec["valu_frac_of_mix"] = pmc["valu_insts_per_call"] / launch_s / mix_occ
        rec["bound"] = "hbm" if rec["physical_frac"] >= (rec.get("valu_frac_of_mix") or 0.0) else "issue"
        out[name] = rec
        del sb
        torch.cuda.empty_cache()
    return out


def step_api_rates(torch, device, copy_gbs, write_gbs=None):
    """The per-step batched API (external or sampled actions, every output written every step)."""
    stream_gbs = max(copy_gbs or 0.0, write_gbs or 0.0) or None
    from colosseumrl_amd.batched import BlokusBatch, TronBatch, TTTBatch
    out = {}

    def rate(fn, calls):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(calls):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / calls, e0.elapsed_time(e1) * 1e-3 / calls

    B, N, P = 65536, 20, 4
    tb = TronBatch(N, P, B, device=device)
    acts = [torch.randint(-1, 2, (P, B), dtype=torch.int8, device=device) for _ in range(16)]
    i = [0]

    def eager():
        tb.step(acts[i[0] & 15], auto_reset=True)
        i[0] += 1
    wall, gpu = rate(eager, 1000)
    out["tron_n20_step_auto_reset"] = {"env_steps_per_s": B / wall, "us_per_call": wall * 1e6, "gpu_us_per_call": gpu * 1e6,
                                       "algorithmic_GBs": (12 * P + 2) * B / gpu / 1e9, "games": B}
    # what the call really moves: rocprofv3 PMC 2 x FETCH_SIZE + WRITE_SIZE per dispatch (profiles/traffic_step_api.json).
    # crl_tron_step probes and writes single bytes, each a 64-byte sector: ~10x its algorithmic 50 B / game -- and still LESS
    # than streaming the boards through LDS would (N*N in + the touched sectors out: (N*N + 50) x B = 29.5 MB one way)
    tj = (_load_json(os.path.join(ROOT, "profiles", "traffic_step_api.json")) or {}).get("kernels", {})

    def pmc_bytes(kernel):
        return (tj.get(kernel) or {}).get("hbm_bytes_per_call")
    if pmc_bytes("tron_step_kernel<4>"):
        nb = pmc_bytes("tron_step_kernel<4>")
        out["tron_n20_step_auto_reset"].update({"hbm_bytes_per_call_pmc": nb, "sector_GBs": nb / gpu / 1e9,
                                                "sector_frac_of_hbm_peak": nb / gpu / 1e9 / HBM_PEAK_GBS,
                                                "board_streaming_bytes": (N * N + 12 * P + 2) * B,
                                                "note": "byte probes / trail writes move whole 64-byte sectors; for a step that also "
                                                        "needs observations use step_observe (one coalesced read of every board)"})
    # the same 16 calls recorded once into a HIP graph and replayed: what a caller that steps in a loop should do about the
    # launch boundary (the call is launch- / latency-bound: 3.3 MB of traffic)
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for k in range(16):
                    tb.step(acts[k], auto_reset=True)
        torch.cuda.current_stream().wait_stream(side)
        wall, gpu = rate(graph.replay, 60)
        out["tron_n20_step_auto_reset_graph16"] = {"env_steps_per_s": 16 * B / wall, "us_per_call": wall * 1e6 / 16,
                                                   "gpu_us_per_call": gpu * 1e6 / 16, "games": B,
                                                   "what": "16 step calls captured in one HIP graph, per call"}
        del graph
    except Exception as exc:                                    # never fatal for the bench line
        out["tron_n20_step_auto_reset_graph16"] = {"error": repr(exc)[:200]}
    buf = tb.observe_all()
    wall, gpu = rate(lambda: tb.observe_all(buf), 500)
    nbytes = (1 + P) * N * N * B
    out["tron_n20_observe_all"] = {"games_per_s": B / wall, "us_per_call": wall * 1e6, "gpu_us_per_call": gpu * 1e6,
                                   "GBs": nbytes / gpu / 1e9, "frac_of_hbm_peak": nbytes / gpu / 1e9 / HBM_PEAK_GBS,
                                   "frac_of_copy": nbytes / gpu / 1e9 / copy_gbs if copy_gbs else None,
                                   "frac_of_stream": nbytes / gpu / 1e9 / stream_gbs if stream_gbs else None,
                                   "bytes_per_call": nbytes, "what": "N*N in + P*N*N out per game (all P observers)"}

    def unfused():
        tb.step(tb.sample(7), auto_reset=True)
        tb.observe_all(buf)
    wall, gpu = rate(unfused, 500)
    out["tron_n20_sample_step_observe_all_3calls"] = {"env_steps_per_s": B / wall, "us_per_call": wall * 1e6,
                                                      "gpu_us_per_call": gpu * 1e6, "GBs": nbytes / gpu / 1e9,
                                                      "frac_of_copy": nbytes / gpu / 1e9 / copy_gbs if copy_gbs else None}
    if hasattr(tb, "step_observe"):
        fo = tb.step_observe(None, seed=7, out=None)

        def fused():
            tb.step_observe(None, seed=7, out=fo)
        wall, gpu = rate(fused, 500)
        out["tron_n20_step_observe_fused"] = {"env_steps_per_s": B / wall, "us_per_call": wall * 1e6, "gpu_us_per_call": gpu * 1e6,
                                              "GBs": nbytes / gpu / 1e9, "frac_of_hbm_peak": nbytes / gpu / 1e9 / HBM_PEAK_GBS,
                                              "frac_of_copy": nbytes / gpu / 1e9 / copy_gbs if copy_gbs else None,
                                              "frac_of_stream": nbytes / gpu / 1e9 / stream_gbs if stream_gbs else None,
                                              "bytes_per_call": nbytes, "hbm_bytes_per_call_pmc": pmc_bytes("tron_step_observe_kernel<4, 64>"),
                                              "what": "ONE launch: sample -> next_state (auto-reset) -> state_to_observation of all P observers; "
                                                      "bytes counted = N*N in + P*N*N out per game"}
    wall, gpu = rate(lambda: tb.ranking(), 200)
    out["tron_n20_ranking"] = {"games_per_s": B / wall, "gpu_us_per_call": gpu * 1e6}
    del tb, buf, acts
    # the reference's DEFAULT board: 19 x 19 = 361 cells, not a whole number of 16-byte chunks (the flat-stream kernels)
    t19 = TronBatch(19, P, B, device=device)
    f19 = t19.step_observe(None, seed=7, out=None)
    wall, gpu = rate(lambda: t19.step_observe(None, seed=7, out=f19), 300)
    nb19 = (1 + P) * 361 * B
    out["tron_n19_step_observe_fused"] = {"env_steps_per_s": B / wall, "gpu_us_per_call": gpu * 1e6, "GBs": nb19 / gpu / 1e9,
                                          "frac_of_hbm_peak": nb19 / gpu / 1e9 / HBM_PEAK_GBS,
                                          "what": "the same ONE launch on the reference's default 19x19 board (361 cells per game: chunks straddle games)"}
    wall, gpu = rate(lambda: t19.rollout(20, 3), 300)
    out["tron_n19_rollout20"] = {"env_steps_per_s": 20 * B / wall, "gpu_us_per_call": gpu * 1e6, "what": "a 20-step fused rollout launch at 19x19"}
    del t19, f19
    Bt = 262144
    tt = TTTBatch((3, 5), 3, 3, Bt, device=device)
    a = torch.randint(0, 15, (Bt,), dtype=torch.int8, device=device)
    wall, gpu = rate(lambda: tt.step(a, auto_reset=True), 1000)
    out["ttt_3x5_step_auto_reset"] = {"env_steps_per_s": Bt / wall, "us_per_call": wall * 1e6, "gpu_us_per_call": gpu * 1e6,
                                      "algorithmic_GBs": 26 * Bt / gpu / 1e9, "games": Bt}

    def ttt_unfused():
        tt.step(tt.sample(3), auto_reset=True)
        tt.valid_mask()
        tt.board(tt.to_move)
    wall, gpu = rate(ttt_unfused, 500)
    out["ttt_3x5_sample_step_valid_board_4calls"] = {"env_steps_per_s": Bt / wall, "us_per_call": wall * 1e6, "gpu_us_per_call": gpu * 1e6}
    if hasattr(tt, "step_observe"):
        to = tt.step_observe(None, seed=3)
        wall, gpu = rate(lambda: tt.step_observe(None, seed=3, out=to), 500)
        out["ttt_3x5_step_observe_fused"] = {"env_steps_per_s": Bt / wall, "us_per_call": wall * 1e6, "gpu_us_per_call": gpu * 1e6,
                                             "what": "ONE launch: sample -> next_state (auto-reset) -> valid mask + observation of the next mover"}
    del tt
    Bb = 16384
    bb = BlokusBatch(Bb, device=device)
    bb.rollout(24, 5)                                      # mid-game positions
    wall, gpu = rate(lambda: bb.valid(), 50)
    out["blokus_valid_count"] = {"games_per_s": Bb / wall, "gpu_us_per_call": gpu * 1e6, "games": Bb}
    count, ids = bb.valid_list(2048)
    wall, gpu = rate(lambda: bb.valid_list(2048, out=ids), 50)
    out["blokus_valid_list"] = {"games_per_s": Bb / wall, "gpu_us_per_call": gpu * 1e6, "games": Bb,
                                "mean_legal_actions": float(count.float().mean().item()), "max_legal_actions": int(count.max().item()),
                                "what": "the ordered legal-action ids of every game, compacted [B][2048] (crl_blokus_valid_list)"}
    rk = torch.zeros((Bb,), dtype=torch.int32, device=device)
    wall, gpu = rate(lambda: bb.select(rk), 50)
    out["blokus_select"] = {"games_per_s": Bb / wall, "gpu_us_per_call": gpu * 1e6, "what": "the r-th legal action for caller-chosen ranks"}
    act = bb.sample(5, advance=False)
    if hasattr(bb, "is_valid"):
        wall, gpu = rate(lambda: bb.is_valid(act), 50)
        out["blokus_is_valid"] = {"games_per_s": Bb / wall, "gpu_us_per_call": gpu * 1e6, "what": "is_valid_action of one dense id per game, no enumeration"}
    occ0, inv0, sc0, rd0, tm0 = bb.occ.clone(), bb.inv.clone(), bb.score.clone(), bb.round.clone(), bb.to_move.clone()

    def bstep():
        bb.occ.copy_(occ0); bb.inv.copy_(inv0); bb.score.copy_(sc0); bb.round.copy_(rd0); bb.to_move.copy_(tm0)
        bb.step(act)
    wall, gpu = rate(bstep, 50)
    out["blokus_step_legal_action"] = {"env_steps_per_s": Bb / wall, "gpu_us_per_call": gpu * 1e6,
                                       "what": "next_state with a sampled legal action (incl. restoring the position: 5 small copies)"}
    pl = torch.zeros((Bb,), dtype=torch.int8, device=device)
    wall, gpu = rate(lambda: bb.observe(pl), 100)
    out["blokus_observe"] = {"obs_per_s": Bb / wall, "gpu_us_per_call": gpu * 1e6}

    def blokus_unfused():
        bb.step(bb.sample(5), auto_reset=True)
        bb.valid()
        bb.observe(bb.to_move.to(torch.int8))
    wall, gpu = rate(blokus_unfused, 100)
    out["blokus_sample_step_valid_observe_4calls"] = {"env_steps_per_s": Bb / wall, "us_per_call": wall * 1e6, "gpu_us_per_call": gpu * 1e6}
    if hasattr(bb, "step_observe"):
        bo = bb.step_observe(None, seed=5)
        wall, gpu = rate(lambda: bb.step_observe(None, seed=5, out=bo), 100)
        out["blokus_step_observe_fused"] = {"env_steps_per_s": Bb / wall, "us_per_call": wall * 1e6, "gpu_us_per_call": gpu * 1e6,
                                            "what": "ONE launch: sample -> next_state (auto-reset) -> legal-action count + observation of the next mover"}
    return out


def reference_dropin_us():
    """The reference's own per-call latency (us), one core, build container: tools/time_reference_dropin.py ->
    profiles/reference_dropin.json ({} when absent: no constants)."""
    return _load_json(os.path.join(ROOT, "profiles", "reference_dropin.json")) or {}


def dropin_latencies(names=("tron", "tictactoe", "tictactoe_3p", "tictactoe_4p", "blokus")):
    """The single-state drop-in API (what match_server.py:193,201-203,218 and ClientEnvironment.py:327-328 call): mean
    microseconds per new_state / next_state / valid_actions / state_to_observation on ONE state, random play with
    restarts at terminal, wall clock around each call (every call ends synchronised: it returns numpy / Python
    objects).  Beside each figure the reference's own, timed in the build container (tools/time_reference_dropin.py)."""
    import random
    from colosseumrl_amd.config import get_environment
    rng = random.Random(0)
    out = {}
    ref_us = reference_dropin_us()

    def one_pass(name):
        env = get_environment(name)() if name != "tron" else get_environment(name)("20;4")
        n_next, n_other = (300, 300) if name == "blokus" else (2000, 1000)
        state, players = env.new_state()
        for _ in range(3):                                  # lazy set-up (context, staging, code objects) outside the timing
            va = env.valid_actions(state, players[0])
            env.state_to_observation(state, players[0])
            acts = [rng.choice(va) for _ in players]
            state, players, _, term, _ = env.next_state(state, players, acts)
            if term:
                state, players = env.new_state()
        t = {"new_state": 0.0, "next_state": 0.0, "valid_actions": 0.0, "state_to_observation": 0.0}
        k = dict.fromkeys(t, 0)
        for i in range(n_next):
            t0 = time.perf_counter()
            va = env.valid_actions(state, players[0])
            t1 = time.perf_counter()
            if i < n_other:
                env.state_to_observation(state, players[0])
                t2 = time.perf_counter()
                t["state_to_observation"] += t2 - t1
                k["state_to_observation"] += 1
            t["valid_actions"] += t1 - t0
            k["valid_actions"] += 1
            acts = [rng.choice(va) for _ in players]
            t0 = time.perf_counter()
            state, players, _, term, _ = env.next_state(state, players, acts)
            t["next_state"] += time.perf_counter() - t0
            k["next_state"] += 1
            if term:
                t0 = time.perf_counter()
                state, players = env.new_state()
                t["new_state"] += time.perf_counter() - t0
                k["new_state"] += 1
        return {m: t[m] / max(k[m], 1) * 1e6 for m in t}, k["next_state"]

    for name in names:
        # two passes, the lower figure per method: whole passes of these ~20-us calls come out at twice the time now and
        # then (seen for whichever env runs first after a different one; per-stream latencies are uniform,
        # tools/debug/stream_latency.py), and a latency is what the call CAN do
        (a, calls), (b, _) = one_pass(name), one_pass(name)
        rec = {m: round(min(a[m], b[m]), 1) for m in a}
        rec["reference_us"] = ref_us.get(name)
        rec["calls"] = calls
        out[name] = rec
    out["what"] = ("us per call of the BaseEnvironment single-state API on one state (B = 1), random play, mean over a pass, the lower "
                   "of two passes; reference_us = the reference's own Python path, one core, build container (it cannot "
                   "travel to the GPU box)")
    return out


def world_of_one_gather_us(torch, dist, make, args, steps_per_launch):
    """Cost of the end-of-rollout collective in a world of ONE rank, for runs that have no process group (the driver's
    plain N = 1 run): a one-rank RCCL group is created here, AFTER the contract measurement, the timed region's launch
    shape is run with and without `gather`, and the group is destroyed again.  {gather_us, region_us, region_no_gather_us}."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    from colosseumrl_amd.parallel import ShardedRollout
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        sr = ShardedRollout(make, WORKLOADS[args.workload][2] if args.batch <= 0 else args.batch)
        def region(mode):
            ts = []
            for i in range(60):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sr.rollout(steps_per_launch, args.seed, steps_per_launch)
                if mode == "gather":
                    sr.gather(dst=0, copy=False)
                elif mode == "all_gather":
                    sr.gather(dst=None, copy=False)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
                if i % 20 == 19:
                    sr.stepper.reset_stats()                # keep the 16-bit rows exact (what a 20-step region ships)
            ts = sorted(ts[10:])
            return ts[len(ts) // 2] * 1e6
        sr.warm_collective(0)
        sr.warm_collective(None)
        region("gather")
        a, b, c = region(None), region("gather"), region("all_gather")
        a2, b2, c2 = region(None), region("gather"), region("all_gather")      # a second round: the order must not decide
        row = sr.gather(dst=0, copy=False)
        return {"gather_us": round(min(b, b2) - min(a, a2), 2), "all_gather_us": round(min(c, c2) - min(a, a2), 2),
                "region_us": round(min(b, b2), 2), "region_all_gather_us": round(min(c, c2), 2), "region_no_gather_us": round(min(a, a2), 2),
                "rounds": [[round(a, 2), round(b, 2), round(c, 2)], [round(a2, 2), round(b2, 2), round(c2, 2)]],
                "row_bytes": int(row.shape[1] * row.element_size()), "what": "median of 50 launch [+ collective] + synchronise "
                "regions of the timed launch shape in a one-rank RCCL group created for this measurement, two rounds (the lower "
                "median of each): no collective / torch.distributed.gather with the per-rank view list / all_gather_into_tensor"}
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _sig(x, n=4):
    return None if x is None else float("%.*g" % (n, x))


def compact_summary(out):
    """~1.5 KB of scalars: per workload env-steps/s (v), ms per launch (ms), HBM fraction of 8 TB/s by the contract's
    formula (alg: algorithmic bytes / launch time; above 1 for long fused launches) and physical (hbm: measured PMC
    traffic / launch time), VALU issue fraction of the measured peak (valu), the CPU oracle's best rate (cpu)."""
    sig = _sig

    def wl(rec):
        r = rec.get("roofline", {})
        return {"v": sig(rec.get("value")), "ms": sig(r.get("launch_ms")), "alg": sig(r.get("frac"), 3), "hbm": sig(r.get("physical_frac"), 3),
                "valu": sig(r.get("valu_issue", {}).get("frac"), 3), "mix": sig(r.get("valu_issue", {}).get("frac_of_mix"), 3),
                "bound": r.get("bound"), "cpu": sig(rec.get("cpu_baseline", {}).get("value"), 3)}
    warmed = out.get("warmed") or {}
    sm = {"headline": {"v": sig(out["value"]), "us": sig(out["timed_region_ms"] * 1e3), "kernel_us": sig((out.get("kernel_ms") or 0) * 1e3) or None,
                       "kernel_dispatch_us": sig((out.get("kernel_ms_dispatch") or 0) * 1e3) or None,
                       "alg": sig(out["roofline"].get("frac"), 3), "hbm": sig(out["roofline"].get("physical_frac"), 3),
                       "of_copy": sig(out["roofline"].get("frac_of_copy"), 3), "warmed_v": sig(warmed.get("value")),
                       "warmed_us": sig(warmed.get("timed_region_us")), "gather_us": (out.get("gather") or {}).get("gather_us")}}
    if "steady_state" in out:
        sm[out["config"]["workload"]] = wl(dict(out["steady_state"], cpu_baseline=out.get("cpu_baseline", {})))
    for name, rec in out.get("others", {}).items():
        sm[name] = wl(rec)
    sa = out.get("step_api", {})
    pick = {"tron_step": "tron_n20_step_auto_reset", "tron_step_observe": "tron_n20_step_observe_fused", "tron19_step_observe": "tron_n19_step_observe_fused",
            "tron_observe_all": "tron_n20_observe_all", "ttt_step_observe": "ttt_3x5_step_observe_fused",
            "blokus_step_observe": "blokus_step_observe_fused", "blokus_valid_list": "blokus_valid_list"}
    if sa:
        sm["step_api_us"] = {k: sig(sa[v].get("gpu_us_per_call"), 3) for k, v in pick.items() if v in sa}
        if "tron_n20_step_observe_fused" in sa:
            sm["step_api_us"]["tron_step_observe_hbm"] = sig(sa["tron_n20_step_observe_fused"].get("frac_of_hbm_peak"), 3)
    if "dropin" in out and "error" not in out["dropin"]:
        sm["dropin_us"] = {k: [v["next_state"], v["valid_actions"], v["state_to_observation"], v["new_state"],
                               (v.get("reference_us") or {}).get("next_state")]
                           for k, v in out["dropin"].items() if isinstance(v, dict)}
        sm["dropin_cols"] = "next_state, valid_actions, state_to_observation, new_state, reference next_state"
    return sm


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data")
CONFIG_KEYS = ("workload", "games_per_gpu", "global_games", "steps_per_launch", "launches", "mean_episode_len", "episodes",
               "parallelism", "gather", "completion", "device_warmup", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel", "launch_ms",
                 "bytes_per_env_step", "physical_frac", "frac_of_copy", "valu_frac", "valu_frac_of_mix", "steady_value",
                 "steady_bound", "steady_frac", "steady_valu_frac", "steady_valu_frac_of_mix",
                 "oc_games", "oc_launch_ms", "oc_frac", "oc_physical_frac", "oc_step_observe_frac",
                 "box_clock_mhz", "box_issue_vs_calibration", "box_clock_mhz_warm", "box_issue_vs_calibration_warm", "note")
LINE_LIMIT = 4096


def compact_line(full):
    """The ONE stdout line: the contract fields, `config`, `roofline` and `cpu_baseline` as scalars, `summary` -- cut down
    from the full record `full` (which goes to bench_detail.json).  Always below LINE_LIMIT bytes: if a summary ever
    outgrows it, the summary's optional sections are dropped, never a contract field."""
    line = {k: full[k] for k in CONTRACT_KEYS}
    line["value"] = _sig(full["value"], 6)
    line["ms_per_step"] = _sig(full["ms_per_step"], 6)
    line["config"] = {k: full["config"][k] for k in CONFIG_KEYS if k in full["config"]}
    r = dict(full["roofline"])
    r["valu_frac"] = (r.get("valu_issue") or {}).get("frac")
    if r.get("note"):
        r["note"] = r["note"][:120]
    line["roofline"] = {k: (_sig(r[k], 5) if isinstance(r[k], float) else r[k]) for k in ROOFLINE_KEYS if r.get(k) is not None}
    line["roofline"].setdefault("traffic", None)
    cb = full.get("cpu_baseline")
    if cb:
        c = {"value": _sig(cb["value"], 5), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:140],
             "nproc": cb.get("nproc")}
        for k, v in cb.items():
            if k.startswith("threads_") and isinstance(v, dict):
                c[k] = _sig(v["value"], 4)
        if isinstance(cb.get("reference_python"), dict):
            c["reference_python"] = _sig(cb["reference_python"]["value"], 4)
        line["cpu_baseline"] = c
    coll = full.get("collective")
    if coll:                                               # the N > 1 record: kernel scaling apart from collective latency
        line["collective"] = {k: (_sig(coll[k], 5) if isinstance(coll[k], float) else coll[k])
                              for k in ("gather_us", "all_gather_us", "region_no_gather_us", "region_gather_us", "region_all_gather_us",
                                        "value_without_gather", "row_bytes", "rows_bytes_into_rank0", "rank_spread_us") if k in coll}
        ranks = coll.get("elapsed_ranks_us") or []
        line["collective"]["elapsed_ranks_us"] = ranks if len(ranks) <= 8 else [min(ranks), max(ranks)]
    for k in ("gather_us", "value_without_gather", "value_warmed", "placement_tests_per_s", "detail"):
        if full.get(k) is not None:
            line[k] = _sig(full[k], 5) if isinstance(full[k], float) else full[k]
    line["summary"] = compact_summary(full)
    for drop in ("dropin_cols", "dropin_us", "step_api_us"):
        if len(json.dumps(line)) < LINE_LIMIT - 96:
            break
        line["summary"].pop(drop, None)
    return line


def emit(full, out=None, detail_paths=None, name="bench_detail.json"):
    """Write the full record to the detail files and to stderr (one line), then the compact line -- the only thing that
    ever goes to stdout -- to `out`."""
    out = out or sys.stdout
    written = []
    for path in (detail_paths if detail_paths is not None else default_detail_paths(name)):
        try:
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            written.append(os.path.relpath(path, ROOT))
        except OSError:
            pass
    full["detail"] = written[0] if written else "stderr"
    line = json.dumps(compact_line(full))
    assert len(line) < LINE_LIMIT, len(line)
    sys.stderr.write("bench detail: " + json.dumps(full) + "\n")
    sys.stderr.flush()
    out.write(line + "\n")
    out.flush()
    return line


def default_detail_paths(name="bench_detail.json"):
    paths = [os.path.join(ROOT, name)]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", name))
    return paths


def fail(msg, code=2):
    """A run that cannot measure what it was asked to: one JSON line on stdout naming the cause, non-zero exit."""
    sys.stdout.write(json.dumps({"error": msg}) + "\n")
    sys.stdout.flush()
    sys.exit(code)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=65536)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="games per GPU (default: the workload's)")
    ap.add_argument("--chunk", type=int, default=0, help="env-steps fused into one kernel launch (default: the workload's)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--device-warmup", action="store_true",
                    help="with --only-headline: also run the W + K region again after ~90 ms of device warm-up (`value_warmed`; "
                         "the full run always does).  `value` is the first W + K region of the process either way")
    ap.add_argument("--gather-probe", action="store_true",
                    help="single process only: afterwards, create a one-rank RCCL group and time the region with / without the gather")
    ap.add_argument("--only-headline", action="store_true", help="skip warmed / steady_state / seeds / others / step_api / dropin (profiling runs)")
    ap.add_argument("--only-step-api", action="store_true", help="run just the per-step API section (profiling runs)")
    ap.add_argument("--only-dropin", action="store_true", help="run just the single-state drop-in latency section")
    ap.add_argument("--only-out-of-cache", action="store_true",
                    help="run just the out-of-Infinity-Cache section (Tron 20x20, 1,048,576 games: 20-step rollout, step_observe, step)")
    return ap.parse_args(argv)


def launcher_env():
    """(under a launcher?, world, rank, local rank) from the environment torch.distributed.run sets."""
    under = all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR"))
    return under, int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def gpus_visible_without_hip():
    """GPUs this process would see, counted WITHOUT bringing up the HIP runtime (torch.cuda.device_count() stays clear of it
    only while torch's amdsmi path works; its fallback is hipGetDeviceCount, and a fork + exec from a process that has
    initialised the GPU is refused on this pool): KFD topology nodes with SIMDs, narrowed by ROCR_VISIBLE_DEVICES /
    HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.  None when the topology cannot be read -- the ranks then check for themselves."""
    import glob
    total = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for path in nodes:
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            total += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var in os.environ:
            ids = [x for x in os.environ[var].split(",") if x.strip()]
            total = min(total, len([x for x in ids if not x.strip().lstrip("-").isdigit() or 0 <= int(x) < total]))
    return total


def self_launch(args, argv):
    """`--gpus N` with N > 1 and no launcher: start N ranks of this script under torch.distributed.run as a CHILD process
    (this process never touches the GPU: devices are counted from the KFD topology, not through HIP), pass its stdout /
    stderr through and return its exit code.  Fewer than N visible devices is an error, never a smaller measurement: said
    here when the topology is readable, else by rank 0 of the child (`main`: exit 2 with the cause on stdout)."""
    import socket
    import subprocess
    have = gpus_visible_without_hip()
    if have is not None and have < args.gpus:
        fail("--gpus %d but only %d GPU(s) visible to this process: refusing to measure fewer GPUs than asked for" % (args.gpus, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    import signal
    child = subprocess.Popen(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), start_new_session=True)

    def forward(signum, _frame):                           # a killed bench must not leave N rank processes on the GPUs
        try:
            os.killpg(child.pid, signum)
        except ProcessLookupError:
            pass
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, forward)
    try:
        return child.wait()
    finally:
        if child.poll() is None:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass


def device_warmup(torch, make_scratch, steps_per_launch, seed):
    """~60 ms of large device copies, then ~30 ms of launches of the timed region's shape on a SCRATCH stepper (other
    memory, other games): the memory clocks follow sustained HBM traffic, which 20-us launches with a synchronise between
    them are not.  Returns the milliseconds spent.  Never part of `value`."""
    t_dev = time.perf_counter()
    blob = torch.empty((256 << 20,), dtype=torch.uint8, device="cuda")
    while time.perf_counter() - t_dev < 0.06:
        blob[: 128 << 20].copy_(blob[128 << 20:])
        torch.cuda.synchronize()
    del blob
    scratch = make_scratch()
    t_launch = time.perf_counter()
    while time.perf_counter() - t_launch < 0.03:
        scratch.rollout(steps_per_launch, seed)
        torch.cuda.synchronize()
    del scratch
    return (time.perf_counter() - t_dev) * 1e3


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    under_launcher, world, rank, local_rank = launcher_env()
    if under_launcher and args.gpus != world:
        if rank == 0:
            fail("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
        sys.exit(2)
    if not under_launcher and (args.gpus > 1 or os.environ.get("CRL_BENCH_SELF_LAUNCH") == "1"):
        sys.exit(self_launch(args, argv))           # (the variable forces the N > 1 start-up path at N = 1: tests on a 1-GPU box)
    if args.gpus < 1:
        fail("--gpus must be >= 1")

    import torch
    import torch.distributed as dist
    if under_launcher and torch.cuda.device_count() < max(world, local_rank + 1) and world > 1:
        if rank == 0:                                   # every rank sees the same count: all leave, rank 0 says why
            fail("--gpus %d but only %d GPU(s) visible to this process: refusing to measure fewer GPUs than asked for" % (args.gpus, torch.cuda.device_count()))
        sys.exit(2)
    if torch.cuda.device_count() < 1:
        fail("no GPU visible to this process (torch.cuda.device_count() == 0): bench.py measures the HIP path only, there is no CPU fallback")
    torch.cuda.set_device(local_rank if under_launcher else 0)
    if under_launcher:                                     # also for a world of ONE rank: the RCCL path is the same code
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", torch.cuda.current_device())
    pl = Plumbing(torch, dist, device, dist.is_initialized())
    assert pl.world == world and pl.rank == rank

    if args.only_dropin:
        print(json.dumps({"dropin": dropin_latencies()}))
        return
    if args.only_out_of_cache:
        print(json.dumps({"out_of_cache": out_of_cache(torch, device, args.seed)}))
        return
    if args.only_step_api:
        copy_gbs, write_gbs = measure_copy_bandwidth(torch, device)
        print(json.dumps({"copy_peak_GBs": copy_gbs, "write_stream_peak_GBs": write_gbs,
                          "step_api": step_api_rates(torch, device, copy_gbs, write_gbs)}))
        return

    game, kw, batch, default_chunk = WORKLOADS[args.workload][:4]
    if args.batch > 0:
        batch = args.batch
    if args.chunk <= 0:
        args.chunk = default_chunk
    from colosseumrl_amd.parallel import ShardedRollout

    def make(batch, first_env_id):
        return make_stepper(game, kw, batch, device, first_env_id)
    # weak scaling: every rank owns `batch` games; global ids rank*batch .. (rank+1)*batch - 1
    sr = ShardedRollout(make, world * batch)
    steps_per_launch = min(args.chunk, args.steps)
    events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    for e in events:
        e.record()                                         # creates the HIP events (torch does so lazily)
    torch.cuda.synchronize()
    # The contract region carries NO timing events at any world size: two hipEventRecord calls are 3.6 us of a 36-us region
    # in a single process and ~10 us next to a collective (see timed_rollout), and a per-N value that is instrumented at
    # N = 1 only would skew the scaling curve the driver derives from these lines.  The launches are timed in further
    # regions of the same shape right after (`launch_time_pass`: recorded markers; `dispatch_time_pass`: events attached
    # to the dispatch).  CRL_BENCH_REGION_EVENTS=1 puts the markers back into the region (single process only).
    region_events = events if (not pl.use_dist and os.environ.get("CRL_BENCH_REGION_EVENTS") == "1") else None

    # ---- THE measurement: the first W + K region of the process.  Everything below it is evidence around it.
    meas = contract_region(pl, sr, args.steps, args.warmup, args.seed, args.chunk, region_events)
    kernel_s = meas["marker_s"]
    if kernel_s is None:
        kernel_s = launch_time_pass(torch, sr, args.steps, args.seed, args.chunk, events)
    kernel_dispatch_s = dispatch_time_pass(torch, sr, args.steps, args.seed, args.chunk, events) if args.steps <= 4096 else None
    # every rank: what the collective costs in this world (further regions of the same shape; never `value`)
    coll = collective_attribution(pl, sr, args.steps, args.seed, args.chunk) if pl.use_dist else None
    if rank == 0:
        copy_gbs, write_gbs = measure_copy_bandwidth(torch, device) if world == 1 else (None, None)
        # The launch's duration for the roofline: HIP events ATTACHED to the dispatch (the kernel's own begin -> end, median
        # of 20 isolated regions of the timed shape) where the stepper offers them -- that is the figure rocprofv3's
        # --stats average for the kernel agrees with; two RECORDED marker events around the launch also see the gap
        # between the first marker and the kernel's start (~3-5 us on a 20-us launch).  Both are in the detail record.
        launches = meas["launches"]
        launch_s = kernel_s / launches
        launch_source = "HIP marker events recorded around the launches of " + ("the timed region" if region_events else "one more region of the timed shape")
        if kernel_dispatch_s and launches == 1:
            launch_s = kernel_dispatch_s
            launch_source = ("HIP events attached to the dispatch (hipExtLaunchKernel start / stop), median of 20 isolated regions of the "
                             "timed shape; marker events recorded around such a launch: %.2f us" % (kernel_s * 1e6))
        box_cold = box_issue_probe(torch, device)          # the box as it is right behind the contract region
        row_bytes = int(meas["rows"].shape[-1] * meas["rows"].element_size())
        gather_desc = ("rccl %s, %d-byte rows" % ("gather to rank 0 (torch.distributed.gather)" if COLLECTIVE == "gather"
                                                 else "all_gather_into_tensor", row_bytes)) if pl.use_dist \
            else "none (single process, no process group)"
        out = contract_record(args.workload, batch, world, args.steps, args.warmup, args.chunk, meas, launch_s, launch_source,
                              copy_gbs, gather_desc)
        out.update({"kernel_ms": kernel_s * 1e3, "kernel_ms_dispatch": kernel_dispatch_s * 1e3 if kernel_dispatch_s else None,
                    "kernel_ms_source": ("HIP marker events around the launches of the timed region" if region_events else
                                         "HIP marker events around the launches of one more region of the same shape, right after the timed "
                                         "one (two event records cost 3.6 us of the region, ~10 us next to a collective: not recorded inside it)")})
        out["config"]["completion"] = pl.completion()
        attach_collective(out, coll, row_bytes, batch, world, args.steps)
        if world > 1:
            out["config"]["cpu_baseline"] = "not run at N > 1 (rank 0 at N = 1 only)"
        out["box"] = {"after_contract_region": box_cold,
                      "what": "crl_diag_issue_probe: independent integer VALU instructions at 4 waves per SIMD on every CU; clock_mhz = "
                              "shader clock under that load; vs_calibration = rate / profiles/%s at the same occupancy"
                              % (os.path.basename(CALIBRATION) if CALIBRATION else "(no calibration file)")}
        out["roofline"]["valu_frac_of_mix"] = (out["roofline"].get("valu_issue") or {}).get("frac_of_mix")
        out["roofline"]["box_clock_mhz"] = box_cold["clock_mhz"]
        out["roofline"]["box_issue_vs_calibration"] = box_cold.get("vs_calibration")
        full_run = world == 1 and not args.only_headline
        if world == 1 and (full_run or args.device_warmup):
            ms = device_warmup(torch, lambda: make_stepper(game, kw, batch, device, 0), steps_per_launch, args.seed)
            sr.stepper.reset()
            sr.stepper.reset_stats()
            w = contract_region(pl, sr, args.steps, args.warmup, args.seed, args.chunk, region_events)
            out["warmed"] = {"value": batch * args.steps / w["elapsed"], "timed_region_us": w["elapsed"] * 1e6,
                             "device_warmup_ms": round(ms, 1), "mean_episode_len": round(mean_episode_len(w["rows"])[0], 3),
                             "what": "the same W + K region again after device copies + launches on a scratch stepper; NOT `value`"}
            out["value_warmed"] = out["warmed"]["value"]
        if full_run:
            seeds = {}
            for sd in (0, 1, 2):                            # seeds {0, 1, 2}, same timed region
                sr.stepper.reset()
                sr.stepper.reset_stats()
                m = contract_region(pl, sr, args.steps, args.warmup, sd, args.chunk, region_events)
                ml, ne = mean_episode_len(m["rows"])
                seeds[str(sd)] = {"value": batch * args.steps / m["elapsed"], "mean_episode_len": round(ml, 3), "episodes": ne}
            vals = [v["value"] for v in seeds.values()]
            seeds["spread"] = (max(vals) - min(vals)) / (sum(vals) / len(vals))
            out["seeds"] = seeds
            out["steady_state"] = steady_state(torch, args.workload, device, args.seed, copy_gbs)
            ssr = out["steady_state"]["roofline"]
            out["roofline"].update({"steady_value": out["steady_state"]["value"], "steady_launch_ms": ssr["launch_ms"],
                                    "steady_frac": ssr["frac"], "steady_physical_frac": ssr["physical_frac"],
                                    "steady_valu_frac": ssr.get("valu_issue", {}).get("frac"),
                                    "steady_valu_frac_of_mix": ssr.get("valu_issue", {}).get("frac_of_mix"),
                                    "steady_bound": ssr.get("bound")})
            others = {}
            for wl in WORKLOADS:
                if wl != args.workload:
                    others[wl] = steady_state(torch, wl, device, args.seed, copy_gbs)
            out["others"] = others
            out["box"]["after_steady_state"] = warm = box_issue_probe(torch, device)      # the box behind ~2 s of rollouts
            out["roofline"]["box_clock_mhz_warm"] = warm["clock_mhz"]
            out["roofline"]["box_issue_vs_calibration_warm"] = warm.get("vs_calibration")
            out["step_api"] = step_api_rates(torch, device, copy_gbs, write_gbs)
            if args.workload == HEADLINE:
                try:                                        # the first real-HBM rows: state that does not fit the Infinity Cache
                    out["out_of_cache"] = oc = out_of_cache(torch, device, args.seed)
                    out["roofline"].update({"oc_games": oc["games"], "oc_launch_ms": oc["rollout"]["launch_ms"],
                                            "oc_frac": oc["rollout"]["frac"], "oc_physical_frac": oc["rollout"]["physical_frac"],
                                            "oc_step_observe_frac": oc["step_observe"]["frac"]})
                except Exception as exc:                    # never fatal for the bench line
                    out["out_of_cache"] = {"error": repr(exc)[:300]}
            out["stream_peaks"] = {"copy_GBs": copy_gbs, "write_GBs": write_gbs,
                                   "what": "measured in this run over 1 GiB: best read+write copy, pure write stream"}
            try:
                out["dropin"] = dropin_latencies()
            except Exception as exc:                        # never fatal for the bench line
                out["dropin"] = {"error": repr(exc)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
            if "placement_tests_per_env_step" in out["cpu_baseline"]:
                out["placement_tests_per_s"] = out["value"] * out["cpu_baseline"]["placement_tests_per_env_step"]
            for wl, rec in out.get("others", {}).items():
                rec["cpu_baseline"] = cpu_baseline(wl, seconds=3.0, quick=True)
                if "placement_tests_per_env_step" in rec["cpu_baseline"]:
                    # reference-equivalent work: placements the reference's loops test per env-step (counted by the
                    # oracle on its sample) x the GPU's env-steps/s; the HIP kernel fits whole shapes instead
                    rec["placement_tests_per_s"] = rec["value"] * rec["cpu_baseline"]["placement_tests_per_env_step"]
        if world == 1 and not pl.use_dist and args.gather_probe and game == "tron":
            try:                                            # creates (and destroys) a one-rank RCCL group: behind a flag
                out["gather"] = world_of_one_gather_us(torch, dist, make, args, steps_per_launch)
            except Exception as exc:
                out["gather"] = {"error": repr(exc)[:300]}
        # the full single-process run owns bench_detail.json; --only-headline and launcher runs write their own file
        emit(out, name="bench_detail.json" if full_run else "bench_detail_n%d%s.json" % (world, "_rccl" if pl.use_dist else "_headline"))
    if pl.use_dist and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    # stdout carries the ONE JSON line and nothing else: libraries that write to file descriptor 1 behind Python's back
    # (RCCL prints a version banner when its communicator comes up) are sent to stderr for the duration.  A self-launch
    # (--gpus N > 1 without a launcher) happens inside main() before anything is imported that could write there, and its
    # child does this redirection itself, so the parent's descriptor 1 is passed through untouched.
    _under, _, _, _ = launcher_env()
    if _under or (parse_args().gpus <= 1 and os.environ.get("CRL_BENCH_SELF_LAUNCH") != "1"):
        sys.stdout.flush()
        _real_stdout = os.dup(1)
        os.dup2(2, 1)
        sys.stdout = os.fdopen(_real_stdout, "w")
    main()
    sys.stdout.flush()
