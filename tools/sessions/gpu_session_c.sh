#!/bin/bash
# Round-2 GPU session C: tests, half-EXEC issue test, Blokus occupancy sweep, profiles of the other workloads, bench.
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; date
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
echo "== valu calibration (with half-EXEC mix)"; date
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rate.hip -o /tmp/valu_rate && timeout -k 10 300 /tmp/valu_rate > gpurun_out/valu_calibration.json; echo "rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/valu_calibration.json'))
for m,rows in d['mixes'].items():
    print(m, [(r['waves_per_simd'], round(r['cycles_per_inst_per_simd'],2), '%.3g'%r['all_wave_insts_per_s']) for r in rows])"
echo "== headline x3"
for i in 1 2 3; do timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['timed_region_ms'], d['kernel_ms'])"; done
echo "== blokus occupancy sweep"; date
for occ in 6 7 8; do
  LIB=$(bash tools/diag_build.sh occ$occ -DBLK_WAVES_PER_SIMD=$occ 2>/dev/null | tail -1)
  for rep in 1 2; do
  CRL_LIB_PATH=$LIB timeout -k 10 300 python3 bench.py --workload blokus_p4_b16384 --steps 4096 --warmup 2048 --only-headline --no-cpu-baseline 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('occ $occ', round(d['value']/1e6,1), 'M env-steps/s')"
  done
done
echo "== profiles"; date
timeout -k 10 300 bash tools/profile_bench.sh r2_tron_n40 --workload tron_p4_n40_b65536 --steps 16384 --warmup 8192 > gpurun_out/prof_r2_tron_n40.log 2>&1; echo "n40 rc=$?"
timeout -k 10 300 bash tools/profile_bench.sh r2_ttt_5x5 --workload ttt_p3_5x5_k4_b262144 --steps 8192 --warmup 2048 > gpurun_out/prof_r2_ttt_5x5.log 2>&1; echo "ttt5 rc=$?"
timeout -k 10 300 bash tools/profile_bench.sh r2_ttt_3x5 --workload ttt_p3_3x5_k3_b262144 --steps 8192 --warmup 2048 > gpurun_out/prof_r2_ttt_3x5.log 2>&1; echo "ttt3 rc=$?"
timeout -k 10 300 bash tools/profile_bench.sh r2_ttt_3x3x3 --workload ttt_p4_3x3x3_b262144 --steps 8192 --warmup 2048 > gpurun_out/prof_r2_ttt_3x3x3.log 2>&1; echo "ttt333 rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_blokus --workload blokus_p4_b16384 --steps 4096 --warmup 2048 > gpurun_out/prof_r2_blokus.log 2>&1; echo "blokus rc=$?"
date
