#!/bin/bash
# A/B the Blokus kernels' register budget on the GPU box: one scratch build per waves/SIMD (tools/diag_build.sh), loaded
# through CRL_LIB_PATH; the in-tree objects and the shipped library are never touched.
set -euo pipefail
for occ in 4 5 6 8; do
  LIB=$(tools/diag_build.sh occ$occ -DBLK_WAVES_PER_SIMD=$occ)
  echo "== waves/SIMD $occ"
  CRL_LIB_PATH=$LIB python3 bench.py --workload blokus_p4_b16384 --steps 256 --warmup 32 --chunk 64 --only-headline --no-cpu-baseline 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), 'M env-steps/s', round(d['ms_per_step'],4), 'ms/step')"
done
