import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
import bench
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
    n = int(os.environ.get("WARM", "3"))
    for _ in range(n):
        sr.warm_collective(dst, 1)
        if os.environ.get("WARM_SYNC", "0") == "1":
            torch.cuda.synchronize()
for _ in range(int(os.environ.get("PREWAIT", "0"))):
    sr.wait()
seq = os.environ.get("SEQ", "wait,wait,wait,wait,wait,wait").split(",")
out = []
for mode in seq:
    if mode == "device": os.environ["CRL_BENCH_SYNC"] = "device"
    else: os.environ.pop("CRL_BENCH_SYNC", None)
    e = bench.timed_rollout(pl, sr, 5 if not out else 20, 0, 8192, None, dst)[0]
    out.append((mode, round(e * 1e6, 1)))
print("SEQ", out)
if use:
    dist.barrier(); dist.destroy_process_group()
