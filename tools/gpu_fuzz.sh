#!/bin/bash
# GPU box (via gpurun): randomised differential soak of every kernel family against the CPU oracle.
#   gpurun --timeout 1200 -- 'bash tools/gpu_fuzz.sh [n_tron] [n_ttt] [n_blokus] [n_step_api] [seed]'
# Defaults are the end-of-round soak (9000 / 6000 / 60 / 6000).  The logs stay under gpurun_out/; a run that FAULTS keeps
# its full log there too -- copy it to profiles/ together with the commit that fixes the cause.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
NT=${1:-9000}; NX=${2:-6000}; NB=${3:-60}; NS=${4:-6000}; SEED=${5:-31415}
timeout -k 10 500 python3 tools/debug/tron_fuzz.py $NT $SEED > gpurun_out/fuzz_tron.log 2>&1; rc=$?; tail -2 gpurun_out/fuzz_tron.log; [ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 tools/debug/ttt_blokus_fuzz.py $NX $NB $((SEED + 1)) > gpurun_out/fuzz_ttt_blokus.log 2>&1; rc=$?; tail -2 gpurun_out/fuzz_ttt_blokus.log; [ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 tools/debug/step_api_fuzz.py $NS $((SEED + 2)) > gpurun_out/fuzz_step_api.log 2>&1; rc=$?; tail -2 gpurun_out/fuzz_step_api.log; [ $rc -ne 0 ] && exit 1
echo "fuzz: clean"
