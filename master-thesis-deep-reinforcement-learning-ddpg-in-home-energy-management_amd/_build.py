"""In-tree build of libshems_hip.so with hipcc for gfx950 (no JIT cache: the .so travels with the tree).

Per translation unit flags matter: the environment kernels replicate the reference's Float32/Float64
arithmetic step by step and must be built with -ffp-contract=off (hipcc's default would fuse a*b+c);
the MFMA / DDPG kernels are tolerance-checked and keep the default contraction.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libshems_hip.so")
ARCH = "gfx950"

# (source, extra flags)
UNITS = [
    ("shems_capi.hip", []),
    ("shems_env.hip", ["-ffp-contract=off"]),
    ("shems_policy.hip", []),
    ("shems_ddpg.hip", []),
    ("shems_gupd.hip", []),
    ("shems_track.hip", ["-ffp-contract=off"]),
    ("shems_wide.hip", []),
    ("shems_train.hip", []),
    ("shems_dp.hip", []),
]
COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
          "-I" + os.path.join(ROOT, "include")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libshems_hip.so cannot be built (there is no CPU fallback)")


def _deps(src):
    d = [src]
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            d.append(os.path.join(CSRC, f))
    d.append(os.path.join(ROOT, "include", "shems_hip.h"))
    d.append(os.path.abspath(__file__))
    return d


def build(force=False, verbose=False, defines=(), tag=""):
    """defines / tag: a diagnostic variant (e.g. defines=("SHEMS_STAMP",), tag="_stamp" -> build/libshems_hip_stamp.so with in-kernel
    phase stamps, tools/stamp_update.py); the product library is the default call.  Variants are built on demand under <repo>/build/
    (objects and library), never in the package directory: only the product library ships with the package."""
    hipcc = _hipcc()
    objs = []
    rebuilt = False
    outdir = HERE if not tag else os.path.join(ROOT, "build")
    objdir = CSRC if not tag else outdir
    os.makedirs(outdir, exist_ok=True)
    lib = os.path.join(outdir, f"libshems_hip{tag}.so")
    for name, extra in UNITS:
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        extra = list(extra) + ["-D" + d for d in defines]
        obj = os.path.join(objdir, name.rsplit(".", 1)[0] + tag + ".o")
        stale = force or not os.path.exists(obj) or any(os.path.getmtime(p) > os.path.getmtime(obj) for p in _deps(src))
        if stale:
            cmd = [hipcc, *COMMON, *extra, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            rebuilt = True
        objs.append(obj)
    if rebuilt or not os.path.exists(lib):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib, *objs, "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    import sys
    if "--stamp-act" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, defines=("SHEMS_STAMP_ACT",), tag="_stampact"))
    elif "--stamp" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, defines=("SHEMS_STAMP",), tag="_stamp"))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
