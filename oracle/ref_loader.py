"""Import the reference colosseumrl envs from /root/reference -- TEST INFRASTRUCTURE ONLY.

Used in the build container to (a) validate the C restatement in ``oracle/`` and
(b) generate the golden vectors committed under ``tests/golden/`` (see
``oracle/gen_golden.py``).  Nothing on the product path, in ``bench.py``'s timed
region, or in the ``-m gpu`` tests imports this module: ``/root/reference`` does
not exist on the GPU box.

How: ``import colosseumrl`` itself fails because the
package ``__init__`` files pull the network stack (spacetime), gym/ray and
pygame, none of which the hot path uses.  So we register *empty* package shells
whose ``__path__`` points into /root/reference and import only the leaf modules
of the hot path.  Three names the leaves import are provided here:

* ``colosseumrl.envs.tron.CyTronGrid``  -> the extension built by build_ref.py
  from the reference's own .pyx (no restatement involved);
* ``numba.jit``                         -> identity decorator.  The reference
  documents this fallback itself (commented block at the top of
  ``envs/blokus/computation.py``); the decorated functions are plain Python;
* ``colosseumrl.envs.blokus.gui``       -> empty module (pygame renderer, never
  called by new_state/next_state/valid_actions).

The reference tree is never modified and nothing is copied out of it.
"""
import importlib
import importlib.util
import os
import sys
import types

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "colosseumrl"))


def _shell(name, path):
    mod = types.ModuleType(name)
    mod.__path__ = [path]
    mod.__package__ = name
    sys.modules[name] = mod
    return mod


_loaded = {}


def load():
    """Return dict of reference classes/modules: tron, ttt2, ttt3, ttt4, blokus (+ modules)."""
    if _loaded:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    from . import build_ref

    so = build_ref.build()
    if so is None:
        raise RuntimeError("the reference's CyTronGrid.pyx could not be built")

    base = os.path.join(REF_ROOT, "colosseumrl")
    _shell("colosseumrl", base)
    _shell("colosseumrl.envs", os.path.join(base, "envs"))
    for sub in ("tron", "blokus", "tictactoe"):
        _shell("colosseumrl.envs." + sub, os.path.join(base, "envs", sub))

    # the real BaseEnvironment module (pure numpy/abc)
    importlib.import_module("colosseumrl.BaseEnvironment")

    spec = importlib.util.spec_from_file_location("colosseumrl.envs.tron.CyTronGrid", so)
    cy = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cy)
    sys.modules["colosseumrl.envs.tron.CyTronGrid"] = cy

    if "numba" not in sys.modules:
        try:
            import numba  # noqa: F401
        except ImportError:
            nb = types.ModuleType("numba")

            def jit(*a, **k):
                if len(a) == 1 and callable(a[0]) and not k:
                    return a[0]
                return lambda f: f

            nb.jit = jit
            sys.modules["numba"] = nb
    sys.modules["colosseumrl.envs.blokus.gui"] = types.ModuleType("colosseumrl.envs.blokus.gui")

    tron = importlib.import_module("colosseumrl.envs.tron.TronGridEnvironment")
    t2 = importlib.import_module("colosseumrl.envs.tictactoe.tictactoe_2p_env")
    t3 = importlib.import_module("colosseumrl.envs.tictactoe.tictactoe_3p_env")
    t4 = importlib.import_module("colosseumrl.envs.tictactoe.tictactoe_4p_env")
    blk = importlib.import_module("colosseumrl.envs.blokus.BlokusEnvironment")
    _loaded.update(
        tron=tron.TronGridEnvironment, tron_mod=tron, cytron=cy,
        ttt2=t2.TicTacToe2PlayerEnv, ttt3=t3.TicTacToe3PlayerEnv, ttt4=t4.TicTacToe4PlayerEnv,
        ttt2_mod=t2, ttt3_mod=t3, ttt4_mod=t4,
        blokus=blk.BlokusEnvironment, blokus_mod=blk,
        blokus_board=importlib.import_module("colosseumrl.envs.blokus.board"),
        blokus_comp=importlib.import_module("colosseumrl.envs.blokus.computation"),
        blokus_ai=importlib.import_module("colosseumrl.envs.blokus.ai"),
    )
    return _loaded
