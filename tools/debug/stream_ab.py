#!/usr/bin/env python3
"""GPU time of the streaming per-step calls of the LOADED library (CRL_LIB_PATH selects a variant: tools/lib_variant.sh) at a
batch inside the Infinity Cache (65,536 games) and outside it (1,048,576): Tron 20x20 step_observe (N*N in, P*N*N out),
observe_all, and the 20-step rollout launch.  One JSON object; run it once per variant.
    CRL_LIB_PATH=build/ab_nt/libcolosseum_hip.so python tools/debug/stream_ab.py [games,games,...]"""
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
import json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from colosseumrl_amd.batched import TronBatch  # noqa: E402


def gpu_us(fn, calls, rounds=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(calls):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / calls)
    ts.sort()
    return [round(ts[len(ts) // 2], 2), round(ts[0], 2)]


def main():
    out = {"lib": os.environ.get("CRL_LIB_PATH", "shipped")}
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [65536, 1 << 20]
    for B in sizes:
        tb = TronBatch(20, 4, B)
        calls = 10 if B > 200000 else 100
        fo = tb.step_observe(None, seed=7, out=None)
        rec = {"step_observe_us": gpu_us(lambda: tb.step_observe(None, seed=7, out=fo), calls)}
        buf = tb.observe_all()
        rec["observe_all_us"] = gpu_us(lambda: tb.observe_all(buf), calls)
        rec["rollout20_us"] = gpu_us(lambda: tb.rollout(20, 3), calls)
        pl = torch.randint(0, 4, (B,), dtype=torch.int8, device="cuda")
        rec["observe_us"] = gpu_us(lambda: tb.observe(pl), calls)
        nbytes = 5 * 400 * B
        rec["step_observe_TBs"] = round(nbytes / rec["step_observe_us"][0] / 1e6, 3)
        out["b%d" % B] = rec
        del tb, fo, buf
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
