#!/bin/bash
# Build a DIAGNOSTIC variant of libcolosseum_hip.so into a scratch directory (never over the in-tree objects) and print
# its path; callers export it as CRL_LIB_PATH (colosseumrl_amd/_native.py loads that instead of the shipped library).
# usage: LIB=$(tools/diag_build.sh <tag> [extra hipcc flags, e.g. -DBLK_STAMPS]); CRL_LIB_PATH=$LIB python ...
set -euo pipefail
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/crl_diag_$TAG
mkdir -p "$OUT"
for f in capi tron ttt blokus; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" \
      -c "$ROOT/colosseumrl_amd/csrc/$f.hip" -o "$OUT/$f.o" &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libcolosseum_hip.so" "$OUT"/capi.o "$OUT"/tron.o "$OUT"/ttt.o "$OUT"/blokus.o
echo "$OUT/libcolosseum_hip.so"
