#!/bin/bash
# Build a VARIANT of libcolosseum_hip.so into build/ab_<tag>/ inside the tree (git-ignored, but it travels to the GPU box
# with the snapshot, unlike /tmp) and print its path.  Only the named source is recompiled with the extra flags; the other
# objects are taken from the shipped build.
# usage: tools/lib_variant.sh <tag> <tron|ttt|blokus|capi> [extra hipcc flags, e.g. -DCRL_QUAD_SKEW=0]
set -euo pipefail
TAG=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/ab_$TAG
mkdir -p "$OUT"
make -s -C "$ROOT/colosseumrl_amd/csrc" -j4 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" \
    -c "$ROOT/colosseumrl_amd/csrc/$SRC.hip" -o "$OUT/$SRC.o"
OBJS=""
for f in capi tron ttt blokus; do
  if [ "$f" = "$SRC" ]; then OBJS="$OBJS $OUT/$f.o"; else OBJS="$OBJS $ROOT/colosseumrl_amd/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libcolosseum_hip.so" $OBJS
echo "$OUT/libcolosseum_hip.so"
