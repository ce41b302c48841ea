#!/bin/bash
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_blokus.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
for occ in 8 7 6; do
  LIB=$(bash tools/diag_build.sh occ$occ -DBLK_WAVES_PER_SIMD=$occ 2>/dev/null | tail -1)
  for rep in 1 2; do
  CRL_LIB_PATH=$LIB timeout -k 10 300 python3 bench.py --workload blokus_p4_b16384 --steps 4096 --warmup 2048 --only-headline --no-cpu-baseline 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('occ $occ', round(d['value']/1e6,1), 'M env-steps/s')"
  done
done
timeout -k 10 400 bash tools/profile_bench.sh r2_blokus --workload blokus_p4_b16384 --steps 4096 --warmup 2048 > gpurun_out/prof_r2_blokus.log 2>&1; echo "blokus rc=$?"
grep "blokus_rollout" gpurun_out/prof_r2_blokus/summary.txt | cut -c1-600
