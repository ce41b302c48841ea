#!/bin/bash
# final profiles of the lane-per-player kernels (both launch shapes of the headline, the 40x40 shard)
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 bash tools/profile_bench.sh r2_tron_n40 --workload tron_p4_n40_b65536 --steps 16384 --warmup 8192 > gpurun_out/prof_r2_tron_n40.log 2>&1; echo "n40 rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_tron_n20 --steps 16384 --warmup 8192 > gpurun_out/prof_r2_tron_n20.log 2>&1; echo "n20 rc=$?"
timeout -k 10 400 bash tools/profile_bench.sh r2_tron_n20_t20 --steps 20 --warmup 20 > gpurun_out/prof_r2_tron_n20_t20.log 2>&1; echo "t20 rc=$?"
for f in tron_n40 tron_n20 tron_n20_t20; do grep "rollout" gpurun_out/prof_r2_$f/summary.txt | head -3 | cut -c1-420; done
find gpurun_out -name "*_kernel_trace.csv" -delete; find gpurun_out -name "*_counter_collection.csv" -delete; du -sh gpurun_out
