#!/bin/bash
# GPU box (via gpurun): the GPU suite's kernel tests and a differential fuzz on the BOUNDS-ASSERT build (tools/lib_bounds.sh,
# built in the container beforehand: `bash tools/lib_bounds.sh`), then the assert counters.
#   bash tools/lib_bounds.sh && gpurun --timeout 1200 -- 'bash tools/gpu_bounds.sh [n_tron n_ttt n_blokus n_step_api seed]'
# Passes when every test passes AND crl_diag_bounds() reports no failed check (its self-test proves per call that a failing
# check would have been recorded).  The counters live in the process that ran the kernels, so each step prints its own
# (tests/conftest.py at session end, the fuzzers at exit: CRL_EXPECT_BOUNDS_BUILD=1 makes a shipped build an error there).
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export CRL_LIB_PATH=$PWD/build/bounds/libcolosseum_hip.so
[ -f "$CRL_LIB_PATH" ] || { echo "build/bounds/libcolosseum_hip.so is missing: run tools/lib_bounds.sh in the container first"; exit 2; }
export CRL_EXPECT_BOUNDS_BUILD=1
timeout -k 10 900 python3 -m pytest tests/test_gpu_tron.py tests/test_gpu_ttt.py tests/test_gpu_blokus.py tests/test_gpu_dropin.py tests/test_gpu_soak.py \
    tests/test_gpu_abi_properties.py -x -q -m gpu > gpurun_out/bounds_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/bounds_pytest.log; grep "bounds asserts" gpurun_out/bounds_pytest.log
[ $rc -ne 0 ] && exit 1
NT=${1:-1500}; NX=${2:-1500}; NB=${3:-20}; NS=${4:-1500}; SEED=${5:-5}
for spec in "tron_fuzz.py $NT $SEED" "ttt_blokus_fuzz.py $NX $NB $((SEED + 1))" "step_api_fuzz.py $NS $((SEED + 2))"; do
  set -- $spec
  timeout -k 10 400 python3 tools/debug/$@ > gpurun_out/bounds_$1.log 2>&1; rc=$?
  tail -3 gpurun_out/bounds_$1.log
  [ $rc -ne 0 ] && exit 1
done
echo "bounds: clean"
