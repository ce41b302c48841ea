"""The CPU restatement under AddressSanitizer + UBSan.  The oracle is the checker of every GPU test, so
an out-of-bounds read in it would be a silent hole in all of them; the reference itself compiles its Cython kernel with
every check off (CyTronGrid.pyx:1).  oracle/Makefile's `liboracle_asan.so` is loaded through ORACLE_LIB in a CHILD process
started with LD_PRELOAD=libasan; the child replays the golden Tron / TicTacToe / Blokus fixtures (the oracle test files
themselves) and ragged random-agent rollouts of each game.  A canary first shows the sanitizer really sees the oracle's
accesses.  CPU only: GPU sanitizers are not available on the pool."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_env():
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    lib = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(lib) or not os.path.exists(lib):
        pytest.skip("libasan.so not found")
    return dict(os.environ, LD_PRELOAD=lib, ORACLE_LIB="liboracle_asan.so", PYTHONPATH=ROOT,
                ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97",       # CPython itself 'leaks' at exit
                UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98")


CANARY = r"""
import numpy as np
from oracle import oracle as O
assert O.lib()._name.endswith("liboracle_asan.so"), O.lib()._name
c, k, out = np.zeros(4, np.uint32), np.zeros(2, np.uint32), np.zeros(2, np.uint32)     # `out` needs 4 words
O.lib().orc_philox4x32(O._p(c), O._p(k), O._p(out))
print("canary-survived")
"""

RAGGED = r"""
import numpy as np
from oracle import oracle as O
assert O.lib()._name.endswith("liboracle_asan.so"), O.lib()._name
# ragged / odd sizes: batches that are no multiple of anything, boards at both ends of the supported range, every player count
for N, P, B, T in ((5, 2, 1, 40), (7, 8, 3, 33), (20, 4, 129, 64), (21, 3, 17, 50), (40, 4, 5, 120), (64, 8, 2, 70)):
    sh, sd = O.tron_start_positions(N, P)
    st = O.TronState(N, P, B)
    O.tron_reset(st, sh, sd)
    O.tron_rollout(st, 3, 1000003, T, sh, sd)
    act = np.random.default_rng(N).integers(-1, 2, size=(P, B)).astype(np.int8)
    O.tron_step(st, act)
    O.tron_observe(st, np.zeros(B, np.int8))
    assert int(st.n_episodes.sum()) > 0
for dims, k, P, B, T in (((3, 3), 3, 2, 1, 30), ((3, 5), 3, 3, 65, 40), ((5, 5), 4, 3, 7, 60), ((3, 3, 3), 3, 4, 33, 50), ((4, 8), 4, 2, 3, 70)):
    tt = O.TTTState(dims, k, P, B)
    O.ttt_rollout(tt, 5, 77, T)
    assert int(tt.n_episodes.sum()) > 0
bb = O.BlokusState(3)
O.blokus_rollout(bb, 9, 12345, 90)                                # a full game and the start of the next
assert int(bb.n_episodes.sum()) >= 3
print("ragged-ok")
"""


def test_sanitizer_sees_the_oracle(no_gpu_context):
    p = subprocess.run([sys.executable, "-c", CANARY], env=_asan_env(), cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "canary-survived" not in p.stdout
    assert "AddressSanitizer: heap-buffer-overflow" in p.stderr and "orc_philox4x32" in p.stderr


def test_oracle_is_clean_under_asan_ubsan(no_gpu_context):
    env = _asan_env()
    p = subprocess.run([sys.executable, "-c", RAGGED], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ragged-ok" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
    files = [os.path.join(ROOT, "tests", f) for f in ("test_oracle_tron.py", "test_oracle_ttt.py", "test_oracle_blokus.py", "test_oracle_rng.py")]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + files, env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert " passed" in p.stdout and "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr
