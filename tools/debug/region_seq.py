#!/usr/bin/env python3
"""A sequence of contract-style regions (bench.timed_rollout: barrier, launch [+ collective], completion) in ONE process: how
the first regions of a process decay, with or without a one-rank RCCL group, ending on the stream wait or on a device
synchronise -- per region.
    [USE_PG=0] [SEQ=wait,device,...] [WARM=n] [PREWAIT=n] python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \\
        --master-addr 127.0.0.1 --master-port 29555 tools/debug/region_seq.py        (USE_PG=0: plain `python`, no group)
The first region runs 5 steps (the contract's W), the others 20.  Prints `SEQ [(mode, us), ...]`.  Round 5 found with it that a
completion flag allocated lazily inside the first region (hipHostMalloc: 1.3 ms) made the SECOND region -- the one the contract
times -- 6 us slower without a process group and 30 us slower with one; the steppers now open the flag at construction."""
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402


def main():
    torch.cuda.set_device(0)
    use = os.environ.get("USE_PG", "1") == "1"
    if use:
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    device = torch.device("cuda", 0)
    pl = bench.Plumbing(torch, dist, device, use)
    from colosseumrl_amd.parallel import ShardedRollout
    game, kw, batch, chunk = bench.WORKLOADS[bench.HEADLINE][:4]
    sr = ShardedRollout(lambda batch, first_env_id: bench.make_stepper(game, kw, batch, device, first_env_id), batch)
    dst = 0 if use else None
    if use:
        for _ in range(int(os.environ.get("WARM", "3"))):
            sr.warm_collective(dst, 1)
    for _ in range(int(os.environ.get("PREWAIT", "0"))):
        sr.wait()
    out = []
    for mode in os.environ.get("SEQ", "wait,wait,wait,wait,wait,wait").split(","):
        if mode == "device":
            os.environ["CRL_BENCH_SYNC"] = "device"
        else:
            os.environ.pop("CRL_BENCH_SYNC", None)
        e = bench.timed_rollout(pl, sr, 5 if not out else 20, 0, 8192, None, dst)[0]
        out.append((mode, round(e * 1e6, 1)))
    print("SEQ", out)
    if use:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
