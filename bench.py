#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched random-agent rollout (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one env-step of EVERY game of the batch (Tron: all live players move).  The default
workload is BASELINE.json configs[1]: 4-player 20x20 Tron, 65,536 games per GPU, uniform random
agent drawn from the counter-based RNG, auto-reset on terminal.  State is resident in HBM before the
timed region; the timed region is W untimed + exactly K timed steps, issued as fused launches of
--chunk steps, bracketed by barrier + synchronize; the time is the max over ranks.  With N > 1 each
rank owns the games [rank*B, (rank+1)*B) (weak scaling, no data-path collective) and the per-game
results are gathered once with a single RCCL all_gather at the end (inside the timed region);
the sharding / gather code is colosseumrl_amd.parallel.ShardedRollout (gloo-tested on CPU).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 4   # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at 2.4 GHz

WORKLOADS = {
    # name: (game, kwargs, per-GPU batch[, env-steps fused into one launch when --chunk is not given (default 2048)])
    "tron_p4_n20_b65536": ("tron", dict(board_size=20, num_players=4), 65536, 8192),   # ~20 us per launch of copies
    "tron_p4_n40_b65536": ("tron", dict(board_size=40, num_players=4), 65536, 8192),   # longer launches: replay epilogue
    "ttt_p3_5x5_k4_b262144": ("ttt", dict(dims=(5, 5), k=4, num_players=3), 262144),
    "ttt_p3_3x5_k3_b262144": ("ttt", dict(dims=(3, 5), k=3, num_players=3), 262144),
    "ttt_p4_3x3x3_b262144": ("ttt", dict(dims=(3, 3, 3), k=3, num_players=4), 262144),
    "blokus_p4_b16384": ("blokus", dict(), 16384),
}


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


def kernel_name(game, kw):
    """The dominant kernel of the workload (what the rocprof summaries under profiles/ list)."""
    if game == "tron":
        n = kw["board_size"]                                     # csrc/tron.hip, crl_tron_rollout's choice
        return "tron_rollout_lds_kernel" if n <= 20 else "tron_rollout_bits_kernel" if n <= 40 else "tron_rollout_kernel"
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


def cpu_baseline(game, kw, seconds=12.0):
    """Time the CPU oracle (bit-exact C restatement, oracle/) on this box's host cores on a bounded sample."""
    from oracle import oracle as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("CRL_CPU_THREADS", "16"))))    # a 1-GPU box's CPU share is 16 cores
    if game == "tron":
        N, P = kw["board_size"], kw["num_players"]
        sh, sd = O.tron_start_positions(N, P)
        B = 65536

        def run(T):
            st = O.TronState(N, P, B)
            O.tron_reset(st, sh, sd)
            t0 = time.perf_counter()
            O.tron_rollout(st, 1, 0, T, sh, sd, n_threads=cores)
            return time.perf_counter() - t0
    elif game == "ttt":
        B = 65536

        def run(T):
            st = O.TTTState(kw["dims"], kw["k"], kw["num_players"], B)
            t0 = time.perf_counter()
            O.ttt_rollout(st, 1, 0, T, n_threads=cores)
            return time.perf_counter() - t0
    elif game == "blokus":
        B = 64 * cores

        def run(T):
            st = O.BlokusState(B)
            t0 = time.perf_counter()
            O.blokus_rollout(st, 1, 0, T, n_threads=cores)
            return time.perf_counter() - t0
    else:
        raise ValueError(game)
    T = 4
    dt = run(T)                       # warms the thread pool
    while dt < 0.5 and T < (1 << 18):  # calibrate on a run long enough to be meaningful
        T *= 4
        dt = run(T)
    rate = B * T / max(dt, 1e-9)
    T = int(max(4, min(1 << 20, rate * seconds / B)))
    dt = run(T)
    return {"value": B * T / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d games x %d steps, oracle/liboracle.so (C, OpenMP over games), %.1f s" % (B, T, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=65536)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="tron_p4_n20_b65536", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="games per GPU (default: the workload's)")
    ap.add_argument("--chunk", type=int, default=0, help="env-steps fused into one kernel launch (default: the workload's, 2048)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    device = torch.device("cuda", torch.cuda.current_device())

    game, kw, batch = WORKLOADS[args.workload][:3]
    if args.batch > 0:
        batch = args.batch
    if args.chunk <= 0:
        args.chunk = WORKLOADS[args.workload][3] if len(WORKLOADS[args.workload]) > 3 else 2048
    from colosseumrl_amd.parallel import ShardedRollout
    # weak scaling: every rank owns `batch` games; global ids rank*batch .. (rank+1)*batch - 1
    sr = ShardedRollout(lambda batch, first_env_id: make_stepper(game, kw, batch, device, first_env_id), world * batch)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sr.rollout(args.warmup, args.seed, args.chunk)
    sr.gather()   # the rollout epilogue (result packing + the gather) is warmed up too: torch loads kernels lazily
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                                           # same stream the kernels are launched on
    launches = sr.rollout(args.steps, args.seed, args.chunk)
    ev1.record()
    gathered = sr.gather()                                 # the one collective: per-game results to every rank
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    n_ep = int(gathered[..., 0].sum().item())
    len_sum = int(gathered[..., 1].sum().item())
    mean_len = len_sum / max(n_ep, 1)

    if rank == 0:
        total_steps = world * batch * args.steps
        value = total_steps / elapsed
        bytes_per_step = algorithmic_bytes_per_step(game, kw, mean_len)
        per_launch_steps = batch * args.steps / launches
        launch_s = kernel_ms * 1e-3 / launches
        achieved = bytes_per_step * per_launch_steps / launch_s / 1e9
        traffic = valu_insts = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # PMC bytes are per launch: only comparable when this run fuses the same number of steps per launch
                same = tj.get("steps_per_launch") == args.chunk
                traffic = tj.get("hbm_bytes_per_launch") if same else None
                valu_insts = tj.get("valu_insts_per_launch") if same else None
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8" if game != "ttt" else "u32", "data": "synthetic",
            "config": {"workload": args.workload, "games_per_gpu": batch, "global_games": world * batch,
                       "steps_per_launch": args.chunk, "agent": "uniform random (Philox-4x32-10), auto-reset",
                       "mean_episode_len": round(mean_len, 3), "episodes": n_ep, "parallelism": "dp%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel_name(game, kw), "launch_ms": launch_s * 1e3,
                         "algorithmic_bytes_per_env_step": round(bytes_per_step, 2)},
        }
        if valu_insts:
            # second view for these integer kernels: wave-instructions the VALUs issued (PMC SQ_INSTS_VALU of the same
            # launch shape, profiles/) over this run's launch time, against one wave64 VALU instruction per SIMD per 4 cycles
            out["roofline"]["valu_issue"] = {"achieved": valu_insts / launch_s, "peak": VALU_PEAK_WAVE_INSTS,
                                             "unit": "wave-instr/s", "frac": valu_insts / launch_s / VALU_PEAK_WAVE_INSTS}
        if out["roofline"]["frac"] > 1.0:
            # the fused rollout keeps boards in LDS for all steps of a launch: the per-step state traffic the
            # algorithmic figure counts (a step-at-a-time stepper) never reaches HBM, so the HBM roofline of
            # that formulation is exceeded; what binds the kernel is instruction issue ("Tron rollout").
            out["roofline"]["note"] = ("state is LDS-resident across the %d fused steps of a launch; measured HBM bytes "
                                       "per launch are in 'traffic'; the kernel is instruction-issue bound" % args.chunk)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(game, kw)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
