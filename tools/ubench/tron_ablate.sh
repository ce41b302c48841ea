#!/bin/bash
# Diagnostic (GPU box): how much of the Tron LDS rollout is LDS probe latency / LDS write traffic?
# Builds tron.hip variants with the probes (1) and also the trail writes (2) replaced by register no-ops
# (results are wrong by construction; only the kernel time matters) and times the headline workload.
set -e
cd colosseumrl_amd/csrc
cp tron.o /tmp/tron.o.keep
for ab in 0 1 2; do
  sed 's#"/root/repo/colosseumrl_amd/csrc/crl_common.hpp"#"crl_common.hpp"#' ../../tools/ubench/tron_ablate.hip.txt > /tmp/tron_ab.hip
  cp /tmp/tron_ab.hip ./tron_ab_tmp.hip
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTRON_ABLATE=$ab -c tron_ab_tmp.hip -o tron.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcolosseum_hip.so capi.o tron.o ttt.o blokus.o
  echo "== TRON_ABLATE=$ab"
  (cd ../.. && python bench.py --steps 4096 --warmup 512 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,2), 'G/s  launch_ms', round(d['roofline']['launch_ms'],4))")
done
rm -f tron_ab_tmp.hip
cp /tmp/tron.o.keep tron.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcolosseum_hip.so capi.o tron.o ttt.o blokus.o
