#!/bin/bash
# issue-rate calibration incl. the 32x32->64 multiply mixes (v_mul_lo + v_mul_hi pairs against v_mad_u64_u32)
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rate.hip -o /tmp/valu_rate && timeout -k 10 300 /tmp/valu_rate > gpurun_out/valu_issue_calibration.json; echo "rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/valu_issue_calibration.json"))
for k, v in d["mixes"].items():
    print(k, [(r["waves_per_simd"], round(r["cycles_per_inst_per_simd"], 2), "%.3g" % r["all_wave_insts_per_s"]) for r in v])
PY
