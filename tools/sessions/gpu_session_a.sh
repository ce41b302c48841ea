#!/bin/bash
# Round-2 GPU session A (run through gpurun): GPU tests, issue-rate calibration, driver-style bench, T=20 launch profile.
set -uo pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; date
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
echo "== valu calibration"; date
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rate.hip -o /tmp/valu_rate && timeout -k 10 300 /tmp/valu_rate > gpurun_out/valu_calibration.json; echo "rc=$?"; cat gpurun_out/valu_calibration.json
echo "== driver-style bench"; date
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err; echo "bench rc=$?"; head -c 3000 gpurun_out/bench_driver.json; tail -3 gpurun_out/bench_driver.err
echo "== headline only, repeated"; 
for i in 1 2 3; do timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --only-headline --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['timed_region_ms'], d['kernel_ms'])"; done
echo "== T=20 launch trace"; date
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t20/trace -- python3 bench.py --steps 20 --warmup 20 --only-headline --no-cpu-baseline > gpurun_out/prof_t20_trace.log 2>&1; echo "rc=$?"
python3 tools/summarize_prof.py gpurun_out/prof_t20 > gpurun_out/prof_t20/summary.txt 2>&1; head -30 gpurun_out/prof_t20/summary.txt
date
