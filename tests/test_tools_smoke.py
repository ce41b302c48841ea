"""tools/ stays runnable: every script compiles, documents itself and answers --help without a GPU; the CPU-only model
runs; the shell scripts parse.  On the GPU box (`-m gpu`) the diagnostic scripts that need no special build run once with
their smallest arguments, so that a change of the library or of bench.py that breaks one is seen the round it happens."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "debug", "*.py"))
                 + glob.glob(os.path.join(ROOT, "examples", "*.py")))


def test_there_are_scripts_to_check():
    assert len(SCRIPTS) >= 12


@pytest.mark.parametrize("path", SCRIPTS, ids=[os.path.relpath(p, ROOT) for p in SCRIPTS])
def test_script_compiles_and_prints_its_usage(path, no_gpu_context):
    p = subprocess.run([sys.executable, path, "--help"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(p.stdout.strip()) > 40                        # the module docstring: what it measures and how to call it


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh"))), ids=os.path.basename)
def test_shell_script_parses(path, no_gpu_context):
    assert subprocess.run(["bash", "-n", path], capture_output=True).returncode == 0
    assert open(path).read().startswith("#!/bin/bash")


def test_readme_names_only_scripts_that_exist():
    import re
    text = open(os.path.join(ROOT, "tools", "README.md")).read()
    for name in set(re.findall(r"`(?:tools/)?((?:debug/|ubench/)?[a-z_0-9]+\.(?:py|sh))`", text)):
        assert os.path.exists(os.path.join(ROOT, "tools", name)) or os.path.exists(os.path.join(ROOT, name)), name


def test_bank_conflict_model_runs(no_gpu_context):
    """The CPU-only LDS bank model that chose the Blokus count pass's table layout (row-major, 9 entries per row)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "blokus_bank_model.py"), "20"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = dict(l.split()[1:3] for l in p.stdout.splitlines() if l.startswith("rs "))      # row-major, entries per row -> cost
    assert float(rows["9"]) < float(rows["10"]) < float(rows["11"])                          # 9 entries per row: what ships


GPU_RUNS = [
    (["tools/debug/tron_fuzz.py", "12", "7"], "12 cases, 0 mismatches"),
    (["tools/debug/ttt_blokus_fuzz.py", "12", "1", "8"], "12 + 1 cases, 0 mismatches"),
    (["tools/debug/step_api_fuzz.py", "12", "9"], "12 cases, 0 mismatches"),
    (["tools/debug/batch_sweep.py", "ttt_p3_3x5_k3_b262144", "64", "4096,8192"], "8192"),
    (["tools/debug/list_ab.py", "colosseumrl_amd/libcolosseum_hip.so"], "mean legal"),
    (["tools/kernel_ab.py", "20", "64"], "quad"),
    (["tools/debug/dropin_stress.py", "3", "1500"], "0 errors"),
    (["tools/debug/host_latency.py"], "region raw"),
    (["tools/debug/step_ab.py", "quick"], "staged_over_bytes"),
    (["tools/debug/dropin_breakdown.py", "200"], "env_next_state"),
    (["tools/debug/shape_sweep.py", "19,24", "4", "4096,4099"], " 24  4    4099"),
    (["tools/debug/shape_sweep.py", "kernels", "24", "1,8", "4096"], "N=24 global"),
    (["tools/debug/shape_sweep.py", "others", "quick"], "    257 |"),
    (["examples/dropin_game.py", "tron", "1"], "ranking"),
    (["examples/dropin_game.py", "blokus", "2"], "ranking"),
    (["examples/dropin_game.py", "tictactoe_4p", "3"], "ranking"),
    (["examples/batched_rollout.py", "8192"], "BlokusVectorEnv: 20 plies"),
    (["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "@FREE_PORT@",
      "tools/debug/gather_latency.py"], "region median"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("argv,expect", GPU_RUNS, ids=[a[-1].split("/")[-1] if a[0] == "-m" else a[0].split("/")[-1] for a, _ in GPU_RUNS])
def test_diagnostic_script_runs_on_the_gpu(run_fresh, argv, expect):
    rc, out = run_fresh([sys.executable] + argv, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT, timeout=600)
    assert rc == 0, out[-3000:]
    assert expect in out, out[-2000:]


UBENCH = sorted(glob.glob(os.path.join(ROOT, "tools", "ubench", "*.hip")))


@pytest.mark.parametrize("path", UBENCH, ids=os.path.basename)
def test_ubench_source_compiles_for_gfx950(path, tmp_path, no_gpu_context):
    """The calibration / latency micro-benchmarks are stand-alone HIP programs built on the GPU box; here: they still compile."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    p = subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", "-c", path, "-o", str(tmp_path / "ubench.o")], capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(open(path).read().split("Build + run on the GPU box")) == 2        # the usage line every ubench carries
