"""ctypes binding of libcolosseum_hip.so (the C ABI in include/colosseum_hip.h).

There is no CPU fallback: if the library is missing or no MI355X is visible, every
compute entry point raises.  (The CPU restatement under oracle/ is test
infrastructure and is never imported from this package.)
"""
import ctypes as C
import os
import subprocess
import threading

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# CRL_LIB_PATH: load a diagnostic build (tools/blokus_stamps.sh, ...) kept OUTSIDE the tree instead of the shipped library
LIB_PATH = os.environ.get("CRL_LIB_PATH") or os.path.join(PKG_DIR, "libcolosseum_hip.so")
CSRC_DIR = os.path.join(PKG_DIR, "csrc")

# the revision of include/colosseum_hip.h this binding (struct layouts, argument lists, RNG contract) was written against
CRL_ABI_VERSION = 111
CRL_STEP_AUTO_RESET = 1
CRL_STEP_BYTES = 2
CRL_STEP_STAGED = 4
CRL_ROLLOUT_NO_LDS = 2
CRL_ROLLOUT_BYTES = 4
CRL_ROLLOUT_BITS = 8
CRL_ROLLOUT_QUAD = 16
CRL_ROLLOUT_QBITS = 32
CRL_ROLLOUT_GQUAD = 64
CRL_ROLLOUT_PAIR = 128

_lib = None
_lock = threading.Lock()


class NativeError(RuntimeError):
    """Raised when libcolosseum_hip.so is missing, cannot be loaded, or a call fails."""


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 with hipcc into colosseumrl_amd/libcolosseum_hip.so (in-tree)."""
    args = ["make", "-C", CSRC_DIR, "-j4"]
    if force:
        args.append("-B")
    proc = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout)
    if proc.returncode != 0:
        raise NativeError("building libcolosseum_hip.so failed (see output above)")
    return LIB_PATH


class TronStats(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("tcount", "tstep", "n_episodes", "win_count", "len_sum", "ret_sum", "last_winners", "last_len", "results",
                 "packed")]


class TTTStats(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("tcount", "tstep", "n_episodes", "win_count", "draw_count", "len_sum", "results")]


class BlokusStats(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("tcount", "tstep", "n_episodes", "win_count", "len_sum", "score_sum", "results")]


_VP, _I, _I64, _U32, _U64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64

# name -> (restype, argtypes); mirrors include/colosseum_hip.h one to one
PROTOTYPES = {
    "crl_last_error": (C.c_char_p, []),
    "crl_version": (_I, []),
    "crl_device_count": (_I, []),
    "crl_destroy": (None, [_VP]),
    "crl_host_alloc": (_I, [C.c_size_t, C.POINTER(_VP), C.POINTER(_VP)]),
    "crl_host_free": (_I, [_VP]),
    "crl_stream_create": (_I, [C.POINTER(_VP)]),
    "crl_stream_destroy": (_I, [_VP]),
    "crl_stream_synchronize": (_I, [_VP]),
    "crl_stream_wait_mapped": (_I, [_VP, _VP, _VP, _U32, C.c_double]),
    "crl_diag_issue_probe": (_I, [_VP, _VP, _I, _I, _VP]),
    "crl_diag_bounds": (_I, [_VP]),
    "crl_philox4x32": (_I, [_VP, _U32, _U32, _VP, _I64, _VP]),
    "crl_tron_create": (_I, [_I, _I, _VP, _VP, C.POINTER(_VP)]),
    "crl_tron_reset": (_I, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "crl_tron_step": (_I, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _U32, _VP]),
    "crl_tron_rollout": (_I, [_VP, _I64, _U64, _U64, _I, _VP, _VP, _VP, _VP, TronStats, _U32, _VP]),
    "crl_tron_rollout_timed": (_I, [_VP, _I64, _U64, _U64, _I, _VP, _VP, _VP, _VP, TronStats, _U32, _VP, _VP, _VP]),
    "crl_tron_observe": (_I, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "crl_tron_observe_all": (_I, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "crl_tron_ranking": (_I, [_VP, _I64, _VP, _VP, _VP, _VP]),
    "crl_tron_step_observe": (_I, [_VP, _I64, _U64, _U64] + [_VP] * 13 + [_U32, _VP]),
    "crl_tron_next_state_inplace64": (_I, [_VP, _I64] + [_VP] * 12 + [_VP]),
    "crl_tron_next_state_inplace64_host": (_I, [_VP] + [_VP] * 12 + [_VP, _VP, _U32, C.c_double]),
    "crl_tron_relative_player_inplace64": (_I, [_VP, _I64, _VP, _I64, _VP, _VP]),
    "crl_tron_sample": (_I, [_VP, _I64, _U64, _U64, _VP, _I, _VP, _VP]),
    "crl_tron_check_state": (_I, [_VP, _I64, _VP, _VP, _VP, _VP]),
    "crl_ttt_create": (_I, [_I, _I, _I, _I, _I, C.POINTER(_VP)]),
    "crl_ttt_lines": (_I, [_VP, _VP, _I]),
    "crl_ttt_reset": (_I, [_VP, _I64, _VP, _VP, _VP, _VP, _VP]),
    "crl_ttt_step": (_I, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _U32, _VP]),
    "crl_ttt_valid": (_I, [_VP, _I64, _VP, _VP, _VP]),
    "crl_ttt_board": (_I, [_VP, _I64, _VP, _VP, _I, _VP, _VP]),
    "crl_ttt_step_board": (_I, [_VP, _I64] + [_VP] * 9 + [_I, _U32, _VP]),
    "crl_ttt_step_board_host": (_I, [_VP] + [_VP] * 9 + [_I, _U32, _VP, _VP, _U32, C.c_double]),
    "crl_ttt_observe_board": (_I, [_VP, _I64, _VP, _VP, _I, _VP, _VP, _VP]),
    "crl_ttt_rollout": (_I, [_VP, _I64, _U64, _U64, _I, _VP, _VP, _VP, TTTStats, _VP]),
    "crl_ttt_sample": (_I, [_VP, _I64, _U64, _U64, _VP, _VP, _I, _VP, _VP]),
    "crl_ttt_step_observe": (_I, [_VP, _I64, _U64, _U64] + [_VP] * 10 + [_I, _U32, _VP]),
    "crl_blokus_create": (_I, [C.POINTER(_VP)]),
    "crl_blokus_placement": (_I, [_I, _I, _I, _VP]),
    "crl_blokus_stamps": (_I, [_VP, _I]),
    "crl_blokus_reset": (_I, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "crl_blokus_step": (_I, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _U32, _VP]),
    "crl_blokus_valid": (_I, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "crl_blokus_valid_list": (_I, [_VP, _I64] + [_VP] * 8 + [_I, _VP]),
    "crl_blokus_select": (_I, [_VP, _I64] + [_VP] * 9 + [_VP]),
    "crl_blokus_is_valid": (_I, [_VP, _I64] + [_VP] * 8 + [_VP]),
    "crl_blokus_fits": (_I, [_VP, _I64, _VP, _VP, _VP, _VP, _VP]),
    "crl_blokus_pack": (_I, [_VP, _I64, _VP, _VP, _VP]),
    "crl_blokus_board": (_I, [_VP, _I64, _VP, _VP, _VP]),
    "crl_blokus_observe": (_I, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "crl_blokus_rollout": (_I, [_VP, _I64, _U64, _U64, _I, _VP, _VP, _VP, _VP, _VP, BlokusStats, _VP]),
    "crl_blokus_sample": (_I, [_VP, _I64, _U64, _U64, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _VP]),
    "crl_blokus_step_observe": (_I, [_VP, _I64, _U64, _U64] + [_VP] * 15 + [_U32, _VP]),
}


def lib():
    """Load the shared library (once).  Raises NativeError when it is not there."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise NativeError(
                    "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
            # torch ships its own libamdhip64; import it first so this library binds to the SAME HIP
            # runtime instance (streams and device pointers are shared with torch tensors)
            import torch  # noqa: F401
            try:
                handle = C.CDLL(LIB_PATH)
            except OSError as e:  # pragma: no cover
                raise NativeError("cannot load %s: %s" % (LIB_PATH, e))
            for name, (res, args) in PROTOTYPES.items():
                try:
                    fn = getattr(handle, name)
                except AttributeError:
                    raise NativeError("%s does not export %s (stale build?)" % (LIB_PATH, name))
                fn.restype = res
                fn.argtypes = args
            got = handle.crl_version()
            if got != CRL_ABI_VERSION:                  # a stale or variant build (CRL_LIB_PATH): by-value structs would be mis-sized
                raise NativeError("%s is ABI revision %d, this binding needs %d (include/colosseum_hip.h CRL_ABI_VERSION): "
                                  "rebuild it" % (LIB_PATH, got, CRL_ABI_VERSION))
            _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().crl_last_error()
        raise NativeError("%s failed (code %d): %s" % (what or "native call", rc, msg.decode() if msg else "?"))


def bounds_report():
    """{'compiled': bool, 'tron' | 'ttt' | 'blokus': {'failures', 'code', 'value', 'limit'}} of the loaded library's bounds
    asserts (``crl_diag_bounds``): all zero for the shipped build, which compiles none (tools/lib_bounds.sh builds the variant
    that does; tools/gpu_bounds.sh runs the suite on it)."""
    out = (C.c_uint32 * 12)()
    rc = lib().crl_diag_bounds(out)
    if rc < 0:
        check(rc, "crl_diag_bounds")
    rep = {"compiled": rc == 1}
    for i, name in enumerate(("tron", "ttt", "blokus")):
        rep[name] = dict(zip(("failures", "code", "value", "limit"), [int(v) for v in out[4 * i:4 * i + 4]]))
    return rep


def require_gpu():
    """Fail loudly unless torch sees a ROCm device AND the HIP library loads."""
    import torch

    handle = lib()
    if not torch.cuda.is_available():
        raise NativeError("no MI355X/ROCm device visible to torch; colosseumrl_amd has no CPU path")
    return handle
