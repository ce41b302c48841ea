set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_blokus.py -m gpu -x -q > gpurun_out/pytest_blk.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_blk.log
[ $rc -ne 0 ] && exit 1
python3 tools/debug/blokus_count_shape.py | tail -6
timeout -k 10 400 python3 tools/lib_ab.py blokus_p4_b16384 512 base=build/ab_tttrot/libcolosseum_hip.so rows=colosseumrl_amd/libcolosseum_hip.so || exit 1
