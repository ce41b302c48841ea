"""Build recipe for the reference's Cython kernel -- TEST INFRASTRUCTURE, never shipped or measured as product.

Compiles the reference's only first-party native file,
``/root/reference/colosseumrl/envs/tron/CyTronGrid.pyx`` (79 lines of Cython),
from where it lies, with the locally installed Cython + gcc.  Outputs (generated
C and the extension ``.so``) go ONLY into a scratch directory OUTSIDE the repository
(``$TMPDIR/colosseumrl_ref_<uid>``): a Python reference must not travel to the GPU
box in any form, compiled or not, and ``gpurun`` ships the whole work tree.  No
reference source is copied anywhere: Cython reads the ``.pyx`` in place and the
generated C is deleted after the compile.

The shipped ``CyTronGrid.c`` (Cython 0.29.13) does not compile against CPython
3.10, so the ``.pyx`` is the build input.

If ``/root/reference`` is absent (the GPU box) there is nothing to build and nothing
to load: ``build()`` returns None.  ``oracle/ref_loader.load()`` builds on demand;
``__graft_entry__.build()`` does NOT call this.
"""
import glob
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(tempfile.gettempdir(), "colosseumrl_ref_%d" % os.getuid())   # never inside the repo
PYX = "/root/reference/colosseumrl/envs/tron/CyTronGrid.pyx"


def built_path():
    hits = glob.glob(os.path.join(REF_DIR, "CyTronGrid*.so"))
    return hits[0] if hits else None


def build(force=False):
    """Compile CyTronGrid.pyx -> <scratch>/CyTronGrid.<abi>.so.  Returns the path, or None without /root/reference."""
    if not os.path.exists(PYX):
        return None
    os.makedirs(REF_DIR, exist_ok=True)
    so = built_path()
    if so and not force and os.path.getmtime(so) >= os.path.getmtime(PYX):
        return so
    import numpy as np

    c_out = os.path.join(REF_DIR, "CyTronGrid.c")
    subprocess.check_call([sys.executable, "-m", "cython", "-3", PYX, "-o", c_out])
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    so = os.path.join(REF_DIR, "CyTronGrid" + ext)
    cc = shutil.which("gcc") or "cc"
    subprocess.check_call([
        cc, "-O2", "-fPIC", "-shared", "-fwrapv", "-fno-strict-aliasing",
        "-I" + sysconfig.get_paths()["include"], "-I" + np.get_include(),
        "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
        c_out, "-o", so,
    ])
    os.remove(c_out)  # generated C embeds the .pyx text as comments: keep only the binary
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
