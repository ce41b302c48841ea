#!/bin/bash
# Build the BOUNDS-ASSERT variant of libcolosseum_hip.so (every source with -DCRL_BOUNDS) into build/bounds/ inside the tree
# (git-ignored, travels to the GPU box with the snapshot) and print its path.  GPU AddressSanitizer is not available on the
# pool: this build carries explicit range checks on the kernels' data-dependent LDS / table accesses instead
# (crl_common.hpp; codes 1xx tron.hip, 2xx ttt.hip, 3xx blokus.hip), read back with crl_diag_bounds().
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/bounds
mkdir -p "$OUT"
rm -f "$OUT"/*.o "$OUT/libcolosseum_hip.so"        # a failed compile must not link against an object of an earlier run
pids=()
for f in capi tron ttt blokus; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DCRL_BOUNDS \
      -c "$ROOT/colosseumrl_amd/csrc/$f.hip" -o "$OUT/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done           # a bare `wait` returns 0 whatever the jobs returned
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libcolosseum_hip.so" "$OUT"/capi.o "$OUT"/tron.o "$OUT"/ttt.o "$OUT"/blokus.o
echo "$OUT/libcolosseum_hip.so"
