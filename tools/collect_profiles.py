#!/usr/bin/env python3
"""Copy the summaries tools/profile_bench.sh left under gpurun_out/prof_<tag>/ into profiles/<prefix>_* and, for
bench workloads, write profiles/traffic_<workload>.json (HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes,
FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950).
usage: tools/collect_profiles.py <tag> <prefix> [<workload> <kernel substring> <steps_per_launch> [<games> [<companion substring>]]]
`companion`: a second kernel that belongs to the same launch (tron_replay_kernel behind tron_rollout_qbits_kernel): its bytes and
instruction counts are added to the launch's, and it is named in the record."""
import glob
import json
import os
import shutil
import sys
if "-h" in sys.argv[1:] or "--help" in sys.argv[1:]:     # usage without touching the GPU (tests/test_tools_smoke.py)
    print(__doc__)
    sys.exit(0)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, prefix + "_rocprofv3_summary.txt"))
shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, prefix + "_rocprofv3_summary.json"))
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:                                       # gpurun merges runs into the same directory: take the newest
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, prefix + "_kernel_stats.csv"))
if len(sys.argv) == 3 and prefix.endswith("step_api"):
    # the per-step API pass: HBM bytes per CALL of every kernel that has both PMC passes -> profiles/traffic_step_api.json
    summ = json.load(open(os.path.join(src, "summary.json")))
    rec = {}
    for name, vals in summ.get("pmc", {}).items():
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals and "namespace" in name:
            short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip()
            rec[short] = {"FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
                          "hbm_bytes_per_call": int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)}
    json.dump({"note": "bench.py --only-step-api under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes): per-dispatch averages; "
                       "hbm_bytes_per_call = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction of MI355X_MICROARCH.md)",
               "source": "profiles/%s_rocprofv3_summary.json" % prefix, "kernels": rec},
              open(os.path.join(dst, "traffic_step_api.json"), "w"), indent=1)
    print(json.dumps({k: v["hbm_bytes_per_call"] for k, v in rec.items()}))
if len(sys.argv) == 3 and prefix.endswith("out_of_cache"):
    # bench.py --only-out-of-cache: Tron 20x20 at 1,048,576 games -> profiles/traffic_out_of_cache.json (per call of the three kernels)
    summ = json.load(open(os.path.join(src, "summary.json")))
    rec = {}
    for name, vals in summ.get("pmc", {}).items():
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip()
        games = {"tron_rollout_quad_kernel<24>": 1 << 20, "tron_step_observe_kernel<4, 64>": 1 << 20, "tron_step_kernel<4>": 1 << 20,
                 "ttt_rollout_kernel<3, 4, true>": 20 * (1 << 20), "blokus_rollout_kernel": 3 * (1 << 18)}
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals and short in games:
            rec[short] = {"FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
                          "hbm_bytes_per_call": int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024),
                          "valu_insts_per_call": vals.get("SQ_INSTS_VALU"), "tcc_hit": vals.get("TCC_HIT_sum"), "tcc_miss": vals.get("TCC_MISS_sum"),
                          "games": games[short]}
            if short.startswith("tron_rollout"):
                rec[short]["steps_per_launch"] = 20
    json.dump({"note": "bench.py --only-out-of-cache under rocprofv3 --pmc (separate passes): Tron 20x20, 1,048,576 games = 436 MB of state, "
                       "TicTacToe 3x5 at 20,971,520 games = 294 MB, Blokus at 786,432 games = 283 MB: each more than the 256 MiB Infinity Cache; per-dispatch averages; hbm_bytes_per_call = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                       "(gfx950 correction of MI355X_MICROARCH.md)",
               "source": "profiles/%s_rocprofv3_summary.json" % prefix, "kernels": rec},
              open(os.path.join(dst, "traffic_out_of_cache.json"), "w"), indent=1)
    print(json.dumps({k: v["hbm_bytes_per_call"] for k, v in rec.items()}))
if len(sys.argv) > 3:
    workload, kern, spl = sys.argv[3], sys.argv[4], int(sys.argv[5])
    summ = json.load(open(os.path.join(src, "summary.json")))
    def find(counter):
        for name, vals in summ.get("pmc", {}).items():
            if kern in name and counter in vals:
                return name, vals[counter]
        raise SystemExit("no %s for %s" % (counter, kern))
    name, fetch = find("FETCH_SIZE")
    _, write = find("WRITE_SIZE")
    def opt(counter):
        try:
            return find(counter)[1]
        except SystemExit:
            return None
    games = int(sys.argv[6]) if len(sys.argv) > 6 else None
    rec = {"kernel": name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip(),
           "steps_per_launch": spl, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
           "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
           "valu_insts_per_launch": opt("SQ_INSTS_VALU"), "salu_insts_per_launch": opt("SQ_INSTS_SALU"),
           "lds_insts_per_launch": opt("SQ_INSTS_LDS"), "waves_per_launch": opt("SQ_WAVES"),
           "source": "profiles/%s_rocprofv3_summary.json" % prefix}
    if games:
        rec["games"] = games
    if len(sys.argv) > 7:                                # the launch's second kernel: one dispatch of it per dispatch of the first
        kern = sys.argv[7]
        cname, cfetch = find("FETCH_SIZE")
        _, cwrite = find("WRITE_SIZE")
        rec["companion"] = {"kernel": cname.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip(),
                            "FETCH_SIZE_KB": cfetch, "WRITE_SIZE_KB": cwrite}
        rec["hbm_bytes_per_launch"] += int((2 * cfetch + cwrite) * 1024)
        for key, counter in (("valu_insts_per_launch", "SQ_INSTS_VALU"), ("salu_insts_per_launch", "SQ_INSTS_SALU"), ("lds_insts_per_launch", "SQ_INSTS_LDS")):
            extra = opt(counter)
            if extra is not None and rec[key] is not None:
                rec[key] += extra
    path = os.path.join(dst, "traffic_%s.json" % workload)
    try:
        out = json.load(open(path))
    except Exception:
        out = {}
    if "launch_shapes" not in out:                       # round-1 format (one launch shape per file) -> keep it as a shape
        old = {k: v for k, v in out.items() if k not in ("workload", "note")}
        out = {"workload": workload, "launch_shapes": ({str(old["steps_per_launch"]): old} if old.get("steps_per_launch") else {})}
    out["workload"] = workload
    out["note"] = ("per launch shape (env-steps fused into one launch): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) and SQ "
                   "instruction counters, per-dispatch averages of the workload's dominant kernel; hbm_bytes_per_launch = "
                   "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 tallies 128-B requests as 64 B); "
                   "bench.py attaches a shape's record only to launches of exactly that many steps")
    out["launch_shapes"][str(spl)] = rec
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(rec))
