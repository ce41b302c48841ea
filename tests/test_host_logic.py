"""CPU-only checks of the host layer: config parsing, spawn layout, registry, C-ABI symbols, loud failure."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import colosseumrl_amd
from colosseumrl_amd import _native
from colosseumrl_amd.BaseEnvironment import SimpleConfigParser
from colosseumrl_amd.envs.tron import layout
from colosseumrl_amd.envs.tron.TronGridEnvironment import TronGridEnvironment, create_tron_config, parse_tron_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_names():
    names = colosseumrl_amd.available_environments()
    for n in ("blokus", "tron", "tictactoe", "tictactoe_3p", "tictactoe_4p"):
        assert n in names
    assert colosseumrl_amd.get_environment("tron") is TronGridEnvironment
    assert colosseumrl_amd.get_environment("tictactoe_3p")().observation_shape == {"board": (3, 5)}
    with pytest.raises(KeyError):
        colosseumrl_amd.get_environment("chess")


def test_tron_config_strings():
    assert tuple(parse_tron_config("")) == (19, 4, -1, False)
    assert parse_tron_config("20") == [20, 4, -1, False]
    assert parse_tron_config("20;6") == [20, 6, -1, False]
    assert parse_tron_config("40;4;5;True") == [40, 4, 5, True]
    assert parse_tron_config("40;4;-1;nope") == [40, 4, -1, False]
    assert create_tron_config(20, 4, -1, False) == "20;4;-1;False"
    env = TronGridEnvironment.create(21, 3)
    assert (env.N, env.num_players, env.fully_observable) == (21, 3, True)
    assert env.min_players == env.max_players == 3
    assert env.observation_names() == ["board", "heads", "directions", "deaths"]
    assert env.observation_shape["board"] == (21, 21)


def test_spawn_layout_matches_reference_golden(golden):
    g = golden("tron_reset")
    for (N, P, ro, so), heads, dirs in zip(g["cfg"], g["heads"], g["dirs"]):
        h, d = layout.start_positions(int(N), int(P), int(ro), [int(so)] * int(P))
        assert h == heads[:P].tolist() and d == dirs[:P].tolist(), (N, P, ro, so)


def test_spawn_layout_random_offsets_golden(golden):
    """Tuple spawn offsets draw from numpy's global RNG seeded with int(time()) (reference :222-224,255)."""
    g = golden("tron_reset_random")
    for (N, P, ro, lo, hi, tval), heads, dirs in zip(g["cfg"], g["heads"], g["dirs"]):
        env = TronGridEnvironment("%d;%d" % (N, P))
        np.random.seed(int(tval))
        h, d = env.generate_start_positions(int(ro), (int(lo), int(hi)))
        assert h.tolist() == heads[:P].tolist() and d.tolist() == dirs[:P].tolist()


def test_simple_config_parser():
    p = SimpleConfigParser(int, (float, 0.5), (str, "x"))
    assert p.parse("3") == [3, 0.5, "x"]
    assert p.parse("3;1.5;None") == [3, 1.5, None]
    with pytest.raises(ValueError):
        SimpleConfigParser(int, int).parse("1")
    assert p.store(7) == "7;0.5;x"


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads and exports exactly what include/colosseum_hip.h declares."""
    header = open(os.path.join(ROOT, "include", "colosseum_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(crl_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 15
    assert os.path.exists(_native.LIB_PATH), "run __graft_entry__.build() first"
    lib = C.CDLL(_native.LIB_PATH)
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_native.PROTOTYPES) == declared      # the ctypes table binds every one of them
    _native.lib()
    assert _native.lib().crl_version() == _native.CRL_ABI_VERSION


def test_binding_refuses_a_library_of_another_abi_revision(monkeypatch):
    """crl_*_stats structs travel by value: a stale / variant .so (CRL_LIB_PATH) must be refused at load, not called."""
    header = open(os.path.join(ROOT, "include", "colosseum_hip.h")).read()
    assert int(re.search(r"#define\s+CRL_ABI_VERSION\s+(\d+)", header).group(1)) == _native.CRL_ABI_VERSION
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "CRL_ABI_VERSION", _native.CRL_ABI_VERSION + 1)
    with pytest.raises(_native.NativeError, match="ABI revision"):
        _native.lib()


def test_native_constants_mirror_the_header():
    """Flag / limit constants the Python side passes through the C ABI are the header's."""
    header = open(os.path.join(ROOT, "include", "colosseum_hip.h")).read()
    defines = {m.group(1): int(m.group(2).rstrip("u"), 0) for m in re.finditer(r"#define\s+(CRL_[A-Z_]+)\s+(-?[0-9a-fx]+u?)\b", header)}
    for name in ("CRL_ABI_VERSION", "CRL_STEP_AUTO_RESET", "CRL_ROLLOUT_NO_LDS", "CRL_ROLLOUT_BYTES", "CRL_ROLLOUT_BITS"):
        assert getattr(_native, name) == defines[name], name


def test_no_cpu_fallback():
    """Without a GPU the product path must raise, not compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = TronGridEnvironment("20;4")
    with pytest.raises(_native.NativeError):
        env.new_state()
    from colosseumrl_amd.batched import TTTBatch
    with pytest.raises(_native.NativeError):
        TTTBatch((3, 3), 3, 2, 4)


def test_product_does_not_import_oracle():
    """Nothing under colosseumrl_amd/ may reference the oracle (test infrastructure)."""
    pkg = os.path.join(ROOT, "colosseumrl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f
                assert "/root/reference" not in text, f


def test_no_compiled_reference_inside_the_repo():
    """A Python reference must not travel to the GPU box in any form: oracle/build_ref.py compiles the reference's
    Cython kernel into a scratch directory outside the repository and __graft_entry__.build() does not build it."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (colosseumrl_amd/envs/tron/CyTronGrid.py is this repository's own stand-in for that module: Python over the C ABI)
    own = os.path.join(root, "colosseumrl_amd", "envs", "tron", "CyTronGrid.py")
    hits = [p for p in glob.glob(os.path.join(root, "**", "CyTronGrid*"), recursive=True) if p != own and "__pycache__" not in p]
    assert hits == [], hits
    assert "crl_tron_next_state_inplace64" in open(os.path.join(root, "colosseumrl_amd", "single.py")).read()
    from oracle import build_ref
    assert not os.path.abspath(build_ref.REF_DIR).startswith(root + os.sep)


@pytest.mark.parametrize("name,env_name", [("2p", "tictactoe"), ("3p", "tictactoe_3p"), ("4p", "tictactoe_4p")])
def test_tictactoe_current_rewards_golden(golden, name, env_name):
    """current_rewards (reference tictactoe_2p_env.py:219-238, 3p :220-239, 4p :250-269) against values the
    reference itself returned on states of its own games (tests/golden/ttt_rewards_*.npz, oracle/gen_golden.py).
    Pure host logic: no GPU involved."""
    g = golden("ttt_rewards_" + name)
    env = colosseumrl_amd.get_environment(env_name)()
    shape = tuple(int(x) for x in g["shape"])
    assert env.max_players == int(g["P"])
    seen = set()
    for board, winner, want in zip(g["board"], g["winner"], g["rewards"]):
        state = (board.reshape(shape), None if winner < 0 else int(winner))
        got = env.current_rewards(state)
        assert got == want.tolist() and all(isinstance(r, int) for r in got)
        seen.add(int(winner))
    assert seen == set(range(-1, env.max_players))          # undecided states and every possible winner are covered


def test_single_state_path_makes_no_copies():
    """The drop-in classes and their stepper (colosseumrl_amd/single.py) stage states in host-mapped memory: no torch
    tensors, no hipMemcpy -- one launch chain and ONE blocking call (crl_stream_synchronize) per GPU call.  Static check
    of the sources (the GPU suite checks the behaviour)."""
    import re
    pkg = os.path.join(ROOT, "colosseumrl_amd")
    files = [os.path.join(pkg, "single.py")]
    for sub in ("tron", "tictactoe", "blokus"):
        d = os.path.join(pkg, "envs", sub)
        files += [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".py")]
    for path in files:
        src = open(path).read()
        code = re.sub(r'""".*?"""', "", src, flags=re.S)
        code = re.sub(r"#.*", "", code)
        assert "torch" not in code, path + " touches torch: the single-state path must not go through device tensors"
        assert ".cpu()" not in code and "copy_(" not in code, path
    single = open(files[0]).read()
    # every public GPU call of a stepper ends in exactly one synchronise
    # (next_state64: its one-call form -- crl_tron_next_state_inplace64_host blocks by itself -- returns before the two-call form's sync)
    for name in ("next_state64", "relative_board64", "observe", "ranking", "reset", "step", "valid", "legal_ids", "is_valid"):
        bodies = re.findall(r"    def %s\(self.*?(?=\n    def |\nclass |\Z)" % name, single, flags=re.S)
        assert bodies, name
        for body in bodies:
            assert body.count("self.sync()") == 1, (name, body.count("self.sync()"))


def test_without_a_gpu_the_single_state_steppers_raise():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from colosseumrl_amd import get_environment
    for name, args in (("tictactoe", ()), ("blokus", ())):
        env = get_environment(name)(*args)
        state, players = env.new_state()                 # host objects only
        with pytest.raises(_native.NativeError):
            env.valid_actions(state, players[0])
