#!/usr/bin/env python3
"""A/B of library VARIANTS (tools/lib_variant.sh) on one device: every variant runs in its own process (the library is
loaded once per process), rounds interleaved so that clock drift hits all alike.

    python tools/lib_ab.py [--isolated] <workload> <steps per launch>[,<steps>...] <tag>=<path to .so> [<tag>=<path> ...]

Default: launches back to back (steady state).  --isolated: one launch between two synchronisations, the way bench.py's
contract region times a short launch (HIP-event time of the launch and wall clock of launch + events + synchronise).

workload: a bench.py WORKLOADS name.  Prints per variant and launch length the median / min kernel time over the rounds
(HIP events around a batch of launches) and env-steps/s."""
import json
import os
import subprocess
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import torch
import bench
game, kw, batch, chunk = bench.WORKLOADS[%(wl)r][:4]
st = bench.make_stepper(game, kw, batch, torch.device("cuda", 0), 0)
out = {}
for T in %(Ts)r:
    reps = max(3, min(200, int(0.05 / max(1e-5, T * 4e-7))))
    for _ in range(3):
        st.rollout(T, 0)
    torch.cuda.synchronize()
    best = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            st.rollout(T, 0)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps)
    best.sort()
    out[str(T)] = [best[len(best) // 2], best[0]]
print("AB " + json.dumps(out))
"""

# isolated launches, the way bench.py's contract region times one: synchronise, event, launch, event, synchronise
CHILD_ISOLATED = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r)
import torch
import bench
game, kw, batch, chunk = bench.WORKLOADS[%(wl)r][:4]
st = bench.make_stepper(game, kw, batch, torch.device("cuda", 0), 0)
out = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for T in %(Ts)r:
    for _ in range(20):
        st.rollout(T, 0)
    torch.cuda.synchronize()
    wall, ev = [], []
    for i in range(300):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        st.rollout(T, 0)
        e1.record()
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
        ev.append(e0.elapsed_time(e1))
    wall.sort(); ev.sort()
    out[str(T)] = [ev[150], ev[15], wall[150]]
print("AB " + json.dumps(out))
"""


def main():
    isolated = "--isolated" in sys.argv
    if isolated:
        sys.argv.remove("--isolated")
    wl, Ts = sys.argv[1], [int(t) for t in sys.argv[2].split(",")]
    variants = [a.split("=", 1) for a in sys.argv[3:]]
    import bench
    batch = bench.WORKLOADS[wl][2]
    res = {tag: {str(T): [] for T in Ts} for tag, _ in variants}
    for rnd in range(3):
        for tag, path in variants:
            env = dict(os.environ, CRL_LIB_PATH=path)
            p = subprocess.run([sys.executable, "-c", (CHILD_ISOLATED if isolated else CHILD) % {"root": ROOT, "wl": wl, "Ts": Ts}], env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("AB ")]
            if p.returncode != 0 or not line:
                print("variant %s failed: %s" % (tag, p.stderr[-2000:]))
                return 1
            for T, v in json.loads(line[-1][3:]).items():
                res[tag][T].append(v)
    for T in Ts:
        for tag, _ in variants:
            med = sorted(v[0] for v in res[tag][str(T)])[1]
            mn = min(v[1] for v in res[tag][str(T)])
            if isolated:
                wall = sorted(v[2] for v in res[tag][str(T)])[1]
                print("%s T=%-5d %-12s isolated launch: events median %7.2f us  p5 %7.2f us   launch+2 events+sync wall %7.2f us" % (wl, T, tag, med * 1e3, mn * 1e3, wall * 1e3), flush=True)
                continue
            print("%s T=%-5d %-12s median %9.4f ms  min %9.4f ms  -> %.4g env-steps/s" % (wl, T, tag, med, mn, batch * T / (med * 1e-3)), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
