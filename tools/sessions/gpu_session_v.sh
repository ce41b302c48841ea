#!/bin/bash
# A/B on ONE box: Philox products as v_mad_u64_u32 (shipped) against the compiler's v_mul_lo/v_mul_hi pair (-DCRL_MUL_PAIR)
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
PAIR=$(bash tools/diag_build.sh mulpair -DCRL_MUL_PAIR) || exit 1
for rep in 1 2; do
  for lib in shipped "$PAIR"; do
    [ "$lib" = shipped ] && unset CRL_LIB_PATH || export CRL_LIB_PATH=$lib
    echo "== lib=$lib"
    timeout -k 10 300 python3 tools/kernel_ab.py 20 8192 || exit 1
    timeout -k 10 300 python3 tools/kernel_ab.py 40 8192 || exit 1
    for w in ttt_p3_5x5_k4_b262144 ttt_p4_3x3x3_b262144 blokus_p4_b16384; do
      timeout -k 10 300 python3 bench.py --workload $w --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '%.4g'%d['value'])"
    done
  done
done
