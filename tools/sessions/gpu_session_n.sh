#!/bin/bash
# A/B of the RNG-advance code layouts (TRON_RNG_VARIANT 0/1/2, burst-unlikely) on both Tron kernels.
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for v in "0" "1" "2" "0 -DTRON_BURST_UNLIKELY" "2 -DTRON_BURST_UNLIKELY"; do   # (the switches existed only for this experiment)
  tag=$(echo "v$v" | tr -d ' -' )
  LIB=$(bash tools/diag_build.sh $tag -DTRON_RNG_VARIANT=$v 2>/dev/null | tail -1)
  echo "== variant $v"
  CRL_LIB_PATH=$LIB timeout -k 10 300 python3 tools/kernel_ab.py 20 8192 | grep -v "^$"
  CRL_LIB_PATH=$LIB timeout -k 10 300 python3 tools/kernel_ab.py 40 8192 | grep bits
done
