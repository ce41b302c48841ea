#!/bin/bash
# A/B the Blokus kernels' register budget on the GPU box: rebuild blokus.o with different waves/SIMD and bench.
set -e
cd colosseumrl_amd/csrc
for occ in 4 5 6 8; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DBLK_WAVES_PER_SIMD=$occ -c blokus.hip -o blokus.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcolosseum_hip.so capi.o tron.o ttt.o blokus.o
  echo "== waves/SIMD $occ"
  (cd ../.. && python bench.py --workload blokus_p4_b16384 --steps 256 --warmup 32 --chunk 64 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), 'M env-steps/s', round(d['ms_per_step'],4), 'ms/step')")
done
