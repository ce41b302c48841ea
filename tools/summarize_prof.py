#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into a small text/JSON summary."""
import csv
import glob
import json
import os
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


summary = {}
# kernel trace
rows = []
for f in find("trace/**/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
by = defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("== kernel trace (us) ==")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-90s n=%5d total=%10.1f avg=%9.2f med=%9.2f min=%9.2f max=%9.2f" % (k[:90], len(v), sum(v), sum(v) / len(v), v2[len(v2) // 2], v2[0], v2[-1]))
    summary.setdefault("kernels", {})[k] = {"calls": len(v), "total_us": sum(v), "avg_us": sum(v) / len(v), "median_us": v2[len(v2) // 2]}
for f in find("trace/**/*kernel_stats.csv"):
    print("== rocprofv3 --stats ==")
    print(open(f).read())
# pmc
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in find(d + "/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        print("== %s (per-dispatch averages) ==" % d)
    for k, cs in acc.items():
        if not any(w in k for w in ("rollout", "replay", "step", "observe", "valid", "sample", "ranking", "reset", "list", "select")):
            continue
        line = {c: sum(v) / len(v) for c, v in cs.items()}
        print(k[:80], json.dumps({c: round(x, 1) for c, x in line.items()}))
        summary.setdefault("pmc", {}).setdefault(k, {}).update(line)
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
