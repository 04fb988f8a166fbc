"""ctypes mirror of include/shems_hip.h -- the only way this package reaches the GPU.

The library is loaded from the package directory (built in-tree by `_build.py`).  If it is missing
or cannot be loaded the import FAILS: there is deliberately no NumPy/PyTorch fallback path.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SHEMS_HIP_LIB") or os.path.join(HERE, "libshems_hip.so")   # override: A/B builds of the library

OK, ERR_ARG, ERR_HIP, ERR_INDEX, ERR_NOMEM, ERR_NODEVICE, ERR_STATE = 0, -1, -2, -3, -4, -5, -6
NSTATE, NACTION, NCOL, NRESULT = 9, 2, 8, 23
TRACK_OFF, TRACK_DRL, TRACK_RULE = 0, 1, -1
ROLLOUT_RULE, ROLLOUT_RANDOM = 0, 1


class ShemsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[shems {code}] {msg}")
        self.code = code


class BoundsError(ShemsError, IndexError):
    """The reference raises Julia's BoundsError when next_state! reads row idx+1 > nrow (LU1:265-279)."""


class Config(C.Structure):
    _fields_ = [("cap_ev", C.c_float), ("soc_max", C.c_float), ("rate_max", C.c_double),
                ("disc_weight", C.c_double), ("disc_pot", C.c_double), ("penalty_weight", C.c_float),
                ("table_row0", C.c_int32), ("nrow", C.c_int32), ("reserved", C.c_int32)]


class View(C.Structure):
    _fields_ = [("n_envs", C.c_int64), ("maxsteps", C.c_int32), ("n_cfg", C.c_int32),
                ("obs", C.c_void_p), ("idx", C.c_void_p), ("step", C.c_void_p),
                ("cfg_of_env", C.c_void_p), ("cfgs", C.c_void_p), ("tables", C.c_void_p),
                ("total_rows", C.c_int64), ("err", C.c_void_p)]


class Replay(C.Structure):
    _fields_ = [("capacity", C.c_int64), ("s", C.c_void_p), ("a", C.c_void_p), ("r", C.c_void_p),
                ("s2", C.c_void_p), ("done", C.c_void_p)]


assert C.sizeof(Config) == 48

_lib = None


def _declare(L):
    vp, i32, i64, u64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint32
    PV = C.POINTER(View)
    sigs = {
        "shems_abi_version": ([], C.c_int),
        "shems_last_error": ([], C.c_char_p),
        "shems_device_count": ([C.POINTER(C.c_int)], C.c_int),
        "shems_create": ([i64, i32, C.c_int, C.POINTER(vp)], C.c_int),
        "shems_destroy": ([vp], C.c_int),
        "shems_n_envs": ([vp, C.POINTER(i64)], C.c_int),
        "shems_set_tables": ([vp, vp, i64], C.c_int),
        "shems_set_configs": ([vp, C.POINTER(Config), i32, vp], C.c_int),
        "shems_reset": ([vp, C.c_int, vp, vp], C.c_int),
        "shems_reset_seeded": ([vp, u64, u32], C.c_int),
        "shems_step": ([vp, vp, C.c_int, vp, vp, vp], C.c_int),
        "shems_action": ([vp, vp, vp], C.c_int),
        "shems_rule_action": ([vp, vp], C.c_int),
        "shems_finished": ([vp, vp], C.c_int),
        "shems_get_state": ([vp, vp, vp, vp], C.c_int),
        "shems_set_state": ([vp, vp, vp, vp], C.c_int),
        "shems_get_view": ([vp, PV], C.c_int),
        "shems_set_stream": ([vp, vp], C.c_int),
        "shems_check_error": ([vp], C.c_int),
        "shems_step_dev": ([PV, vp, C.c_int, vp, vp, vp, vp, vp], C.c_int),
        "shems_action_dev": ([PV, vp, C.c_int, vp, vp], C.c_int),
        "shems_reset_dev": ([PV, C.c_int, vp, vp, vp], C.c_int),
        "shems_reset_seeded_dev": ([PV, u64, u32, vp], C.c_int),
        "shems_scale_action_dev": ([vp, i64, vp, vp], C.c_int),
        "shems_rollout_dev": ([PV, C.c_int, i32, u64, vp, C.POINTER(Replay), i64, i64, vp], C.c_int),
        "shems_track_dev": ([PV, vp, i64, C.c_int, i32, vp, i64, vp, vp], C.c_int),
        "shems_track": ([vp, vp, vp, vp, C.c_int, i32, vp, vp], C.c_int),
    }
    for name, (args, res) in sigs.items():
        fn = getattr(L, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.argtypes = args
        fn.restype = res
    return sigs


def exported_symbols():
    """Names include/shems_hip.h declares (used by the CPU-side ABI test)."""
    import re
    hdr = os.path.join(os.path.dirname(HERE), "include", "shems_hip.h")
    txt = open(hdr).read()
    return sorted(set(re.findall(r"\b(shems_[a-z0-9_]+)\s*\(", txt)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64
        # (torch/lib, SONAME libamdhip64.so.7).  If /opt/rocm's copy is mapped first and torch's
        # second, the second HSA runtime finds no GPU.  Loading torch first makes our DT_NEEDED
        # libamdhip64.so.7 resolve to the copy torch already mapped, so tensors, streams and our
        # kernels share one runtime.  Without torch (e.g. a Julia host) /opt/rocm's runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        _declare(L)
        _lib = L
    return _lib


def check(rc):
    if rc == OK:
        return
    msg = lib().shems_last_error().decode("utf-8", "replace")
    if rc == ERR_INDEX:
        raise BoundsError(rc, msg)
    raise ShemsError(rc, msg)
