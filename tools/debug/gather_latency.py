#!/usr/bin/env python3
"""Host-visible cost of the end-of-rollout collective in a world of ONE rank (bring-up aid for bench.py's `gather_us`):
    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 tools/debug/gather_latency.py
Times launch -> collective -> synchronise for a 20-step Tron launch with (a) no collective, (b) dist.gather to rank 0,
(c) all_gather_into_tensor, for the wide int32 rows (44 B / game) and 16-byte rows."""
import os
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
from colosseumrl_amd.batched import TronBatch  # noqa: E402

B = 65536
tb = TronBatch(20, 4, B)
wide = tb.results(copy=False) if "copy" in TronBatch.results.__code__.co_varnames else tb.results()
narrow = torch.zeros((B, 4), dtype=torch.int32, device=wide.device)      # 16-byte rows (NCCL has no int16)
for name, payload in (("wide 44 B/game", wide), ("packed 16 B/game", narrow)):
    recv = torch.empty_like(payload)

    def none():
        pass

    def gather():
        dist.gather(payload, [recv], dst=0)

    def allgather():
        dist.all_gather_into_tensor(recv, payload)

    def copy():
        recv.copy_(payload)

    for label, fn in (("no collective", none), ("dist.gather", gather), ("all_gather_into_tensor", allgather), ("plain copy", copy)):
        for _ in range(20):
            tb.rollout(20, 0)
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(300):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tb.rollout(20, 0)
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print("%-18s %-24s median %.1f us  p10 %.1f  min %.1f" % (name, label, ts[150] * 1e6, ts[30] * 1e6, ts[0] * 1e6), flush=True)

# what precedes the region matters: bench.py's contract region comes right after barrier + synchronise
from colosseumrl_amd.parallel import ShardedRollout  # noqa: E402
sr = ShardedRollout(lambda batch, first_env_id: TronBatch(20, 4, batch, first_env_id=first_env_id), B)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def region(before, events):
    ts, ks = [], []
    for i in range(120):
        torch.cuda.synchronize()
        before()
        t0 = time.perf_counter()
        if events:
            e0.record()
        sr.rollout(20, 0, 20)
        if events:
            e1.record()
        sr.gather(dst=0, copy=False)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        if events:
            ks.append(e0.elapsed_time(e1))
        if i % 40 == 39:
            sr.stepper.reset_stats()
    ts.sort(); ks.sort()
    return ts[60] * 1e6, (ks[60] * 1e3 if ks else 0.0)


def nothing():
    pass


def barrier():
    dist.barrier()
    torch.cuda.synchronize()


def sleep200():
    time.sleep(200e-6)


def sleep1ms():
    time.sleep(1e-3)


for label, before in (("nothing", nothing), ("dist.barrier + sync", barrier), ("sleep 200 us", sleep200), ("sleep 1 ms", sleep1ms)):
    for ev in (False, True):
        r, k = region(before, ev)
        print("before the region: %-20s events %-5s region median %.1f us  kernel (events) %.1f us" % (label, ev, r, k), flush=True)
dist.destroy_process_group()
