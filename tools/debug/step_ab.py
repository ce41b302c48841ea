#!/usr/bin/env python3
"""A/B of crl_tron_step's two interchangeable kernels (one library, pinned by flag): byte probes in HBM
(tron_step_kernel: one lane per game, every probe / trail store a 64-byte sector) against boards staged through LDS
(tron_step_staged_kernel: one coalesced read of every board).  Tron 20x20 P4 at BASELINE's batch (65,536 games: 27 MB of
state, resident in the 256 MiB Infinity Cache between calls) and at 1,048,576 games (436 MB: out of it); also 40x40.
    python tools/debug/step_ab.py            # one JSON object: GPU us per call (median / min of interleaved rounds)
Random actions, auto-reset on; HIP events around `calls` back-to-back calls per round."""
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
import json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from colosseumrl_amd.batched import TronBatch  # noqa: E402


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    out = {}
    for N, B in ((20, 4096),) if quick else ((20, 65536), (20, 1 << 20), (40, 65536), (40, 1 << 18)):
        P = 4
        tb = TronBatch(N, P, B)
        acts = [torch.randint(-1, 2, (P, B), dtype=torch.int8, device="cuda") for _ in range(8)]
        calls = 20 if B > 200000 else 200
        times = {"bytes": [], "staged": []}
        for rnd in range(7):
            for k in ("bytes", "staged"):
                tb.step(acts[0], auto_reset=True, kernel=k)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(calls):
                    tb.step(acts[i & 7], auto_reset=True, kernel=k)
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) * 1e3 / calls)
        rec = {"games": B, "state_bytes": B * (N * N + 4 * P), "algorithmic_bytes": (12 * P + 2) * B}
        for k, ts in times.items():
            ts = sorted(ts[1:])
            rec[k + "_us"] = [round(ts[len(ts) // 2], 2), round(ts[0], 2)]
        rec["staged_over_bytes"] = round(rec["staged_us"][0] / rec["bytes_us"][0], 3)
        out["tron_n%d_b%d" % (N, B)] = rec
        del tb, acts
        torch.cuda.empty_cache()
    out["what"] = "GPU us per crl_tron_step call (auto-reset, random actions): [median, min] over 6 interleaved rounds"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
