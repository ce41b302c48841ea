#!/usr/bin/env python3
"""Every batched Tron call over a grid of board sizes, player counts and batch sizes: where does a shape fall off the fast
kernels?  Prints, per (N, P, B), microseconds per call (median of back-to-back launches) and nanoseconds per game for the
20-step rollout, a 512-step rollout (per step), step, step_observe, observe_all, observe, ranking and reset.
    python tools/debug/shape_sweep.py [N,N,...] [P,P,...] [B,B,...] [json path]
    python tools/debug/shape_sweep.py kernels [N,N,...] [T,T,...] [B [P]]
    python tools/debug/shape_sweep.py others [quick]
`kernels`: the interchangeable rollout kernels ("auto" = the library's choice, "qbits", "bits", "bytes", "global", "gquad") against the
launch length at P = 4 -- where the fixed cost of an LDS-resident launch (copy in, replay, copy out) is worth it.
`others`: TicTacToe over ten board shapes and Blokus over batch sizes, whole and ragged batches (no cliffs: +10-15 % for a
ragged batch's extra workgroup)."""
import json
import os
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from colosseumrl_amd.batched import TronBatch

def timed(fn, reps=10, rounds=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    ts.sort()
    return ts[len(ts) // 2]


arg = sys.argv[1:]
if arg and arg[0] == "others":
    from colosseumrl_amd.batched import TTTBatch, BlokusBatch
    dev = torch.device("cuda", 0)
    QUICK = len(arg) > 1 and arg[1] == "quick"
    print("TTT: dims k P B | roll8 roll512/step step_observe sample+step")
    SHAPES = (((3, 3), 3, 2), ((3, 5), 3, 3), ((3, 3, 3), 3, 4), ((5, 5), 4, 3), ((4, 4), 3, 2), ((4, 4), 4, 2), ((6, 5), 4, 3),
              ((4, 8), 4, 4), ((2, 4, 4), 3, 3), ((5, 6), 5, 2))
    for dims, k, P in (SHAPES[:2] if QUICK else SHAPES):
        for B in ((4096, 4097) if QUICK else (262144, 262145, 1000)):
            st = TTTBatch(dims, k, P, B, device=dev)
            r8 = timed(lambda: st.rollout(8, 1))
            r512 = timed(lambda: st.rollout(512, 1), reps=2, rounds=3) / 512
            so = timed(lambda: st.step_observe(None, 1, True))
            ss = timed(lambda: st.step(st.sample(1), auto_reset=True))
            print("%-10s k%d P%d %7d | %8.1f %8.4f %8.1f %8.1f" % (dims, k, P, B, r8, r512, so, ss), flush=True)
            del st
    print("Blokus: B | roll8 roll64/step step_observe valid_list select")
    for B in ((256, 257) if QUICK else (16384, 16385, 16387, 1000, 4096, 65536)):
        st = BlokusBatch(B, device=dev)
        st.rollout(24, 3)
        r8 = timed(lambda: st.rollout(8, 1), reps=3, rounds=3)
        r64 = timed(lambda: st.rollout(64, 1), reps=1, rounds=3) / 64
        so = timed(lambda: st.step_observe(None, 1, True), reps=5, rounds=3)
        out = torch.empty((B, 2048), dtype=torch.int32, device=dev)
        vl = timed(lambda: st.valid_list(2048, None, out), reps=5, rounds=3)
        rank = torch.zeros((B,), dtype=torch.int32, device=dev)
        se = timed(lambda: st.select(rank), reps=5, rounds=3)
        print("%7d | %8.1f %8.2f %8.1f %8.1f %8.1f" % (B, r8, r64, so, vl, se), flush=True)
        del st, out
    sys.exit(0)
if arg and arg[0] == "kernels":
    dev = torch.device("cuda", 0)
    Ns = [int(x) for x in (arg[1] if len(arg) > 1 else "20,21,24,28,32,39,40").split(",")]
    Ts = [int(x) for x in (arg[2] if len(arg) > 2 else "1,4,20,64,256").split(",")]
    Bk = int(arg[3]) if len(arg) > 3 else 65536
    Pk = int(arg[4]) if len(arg) > 4 else 4
    KERNS = ("auto", "quad", "pair", "qbits", "bits", "bytes", "global", "gquad") if Pk <= 2 else ("auto", "qbits", "bits", "bytes", "global", "gquad")
    for N in Ns:
        st = TronBatch(N, Pk, Bk, device=dev)
        for kern in KERNS:
            row = []
            for T in Ts:
                for _ in range(2):
                    st.rollout(T, 1, kernel=kern)
                torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        st.rollout(T, 1, kernel=kern)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / 5)
                row.append("%8.1f" % sorted(ts)[1])
            print("N=%d %-6s us per launch at T=%s: %s" % (N, kern, ",".join(map(str, Ts)), " ".join(row)), flush=True)
        del st
    sys.exit(0)
NS = [int(x) for x in (arg[0] if len(arg) > 0 else "10,15,16,19,20,21,24,30,39,40,41,50").split(",")]
PS = [int(x) for x in (arg[1] if len(arg) > 1 else "2,3,4,6,8").split(",")]
BS = [int(x) for x in (arg[2] if len(arg) > 2 else "65536,65553").split(",")]
OUT = arg[3] if len(arg) > 3 else None
dev = torch.device("cuda", 0)


rows = []
print("%3s %2s %7s | %9s %9s %8s %9s %9s %8s %8s %8s   (us per call; roll512 = us per step)" %
      ("N", "P", "B", "roll20", "roll512", "step", "step_obs", "obs_all", "observe", "ranking", "reset"), flush=True)
for N in NS:
    for P in PS:
        if 4 * (N // 2 - 1) < P:        # the spawn ring must hold every player
            continue
        for B in BS:
            st = TronBatch(N, P, B, device=dev)
            act = torch.zeros((P, B), dtype=torch.int8, device=dev)
            who = torch.zeros((B,), dtype=torch.int8, device=dev)
            bufs = st.observe_all_buffers()
            r = {"N": N, "P": P, "B": B}
            r["roll20"] = timed(lambda: st.rollout(20, 1))
            r["roll512"] = timed(lambda: st.rollout(512, 1), reps=2, rounds=3) / 512
            r["step"] = timed(lambda: st.step(act, auto_reset=True))
            r["step_obs"] = timed(lambda: st.step_observe(None, 1, True, bufs))
            r["obs_all"] = timed(lambda: st.observe_all(bufs))
            r["observe"] = timed(lambda: st.observe(who))
            r["ranking"] = timed(lambda: st.ranking())
            r["reset"] = timed(lambda: st.reset())
            rows.append(r)
            print("%3d %2d %7d | %9.1f %9.3f %8.1f %9.1f %9.1f %8.1f %8.1f %8.1f" %
                  (N, P, B, r["roll20"], r["roll512"], r["step"], r["step_obs"], r["obs_all"], r["observe"], r["ranking"],
                   r["reset"]), flush=True)
            del st, bufs, act, who
            torch.cuda.empty_cache()
if OUT:
    with open(OUT, "w") as f:
        json.dump(rows, f, indent=0)
