#!/bin/bash
# GPU box (via gpurun): rocprofv3 kernel trace + PMC passes for a SET of bench workloads, one call.
#   gpurun --timeout 1200 -- 'bash tools/gpu_profile.sh r3 tron_n20_t20 tron_n20 tron_n40 blokus step_api'
# <round prefix> then any of: tron_n20_t20 (the driver's 20-step launch) tron_n20 tron_n40 ttt_5x5 ttt_3x5 ttt_3x3x3 blokus step_api out_of_cache
# Leaves gpurun_out/prof_<prefix>_<name>/summary.{txt,json}; back in the build container `tools/collect_profiles.py`
# copies them into profiles/ (see profiles/README.md).  A pass that fails or times out ends the script (no further GPU step).
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PFX=$1; shift
for NAME in "$@"; do
  case $NAME in
    tron_n20_t20) ARGS="--steps 20 --warmup 20" ;;
    tron_n20)     ARGS="--steps 16384 --warmup 8192" ;;
    tron_n40)     ARGS="--workload tron_p4_n40_b65536 --steps 16384 --warmup 8192" ;;
    ttt_5x5)      ARGS="--workload ttt_p3_5x5_k4_b262144 --steps 8192 --warmup 2048" ;;
    ttt_3x5)      ARGS="--workload ttt_p3_3x5_k3_b262144 --steps 8192 --warmup 2048" ;;
    ttt_3x3x3)    ARGS="--workload ttt_p4_3x3x3_b262144 --steps 8192 --warmup 2048" ;;
    blokus)       ARGS="--workload blokus_p4_b16384 --steps 4096 --warmup 2048" ;;
    step_api)     ARGS="--only-step-api" ;;
    out_of_cache) ARGS="--only-out-of-cache" ;;     # Tron 20x20 at 1,048,576 games (436 MB of state > the 256 MiB Infinity Cache)
    *) echo "unknown profile set $NAME"; exit 2 ;;
  esac
  timeout -k 10 420 bash tools/profile_bench.sh ${PFX}_$NAME $ARGS > gpurun_out/prof_${PFX}_$NAME.log 2>&1
  rc=$?; echo "$NAME rc=$rc"
  [ $rc -ne 0 ] && { tail -5 gpurun_out/prof_${PFX}_$NAME.log; exit 1; }
  grep -i "rollout\|step_kernel\|observe" gpurun_out/prof_${PFX}_$NAME/summary.txt | head -4 | cut -c1-300
done
# the raw per-dispatch tables are large; the summaries are what travels back
find gpurun_out -name "*_kernel_trace.csv" -delete; find gpurun_out -name "*_counter_collection.csv" -delete; du -sh gpurun_out
