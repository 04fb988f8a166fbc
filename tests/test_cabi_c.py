"""The drop-in boundary from a compiled caller: tests/cabi/drive_cabi.c is plain C against include/shems_hip.h (no Python, no
PyTorch in its process).  CPU: it compiles with -Wall -Werror and links against libshems_hip.so.  GPU: it runs, and everything it
wrote (start state, 12 x step! with the 23-column results rows, the rule-based action) is bit-identical to the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

import util as U
from util import oracle_c

SRC = os.path.join(U.ROOT, "tests", "cabi", "drive_cabi.c")
N, NSTEPS = 300, 12


def _build(built_lib):
    out_dir = os.path.join(U.ROOT, "tests", "cabi", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "drive_cabi")
    libdir = os.path.dirname(built_lib)
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(U.ROOT, "include"), SRC, "-o", exe,
           "-L", libdir, "-lshems_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_caller_compiles_and_links_against_the_header(built_lib):
    exe = _build(built_lib)
    assert os.access(exe, os.X_OK)


@pytest.mark.gpu
def test_c_caller_matches_the_oracle_bit_for_bit(built_lib, tmp_path):
    exe = _build(built_lib)
    tab = U.tables_mod().synthetic_table("train", 98)
    tpath, opath = str(tmp_path / "table.bin"), str(tmp_path / "out.bin")
    np.ascontiguousarray(tab, np.float32).tofile(tpath)
    run = subprocess.run([exe, tpath, str(tab.shape[0]), str(N), str(NSTEPS), opath], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr + run.stdout
    raw = np.fromfile(opath, np.uint8)
    off = 0

    def take(dtype, *shape):
        nonlocal off
        cnt = int(np.prod(shape)) * np.dtype(dtype).itemsize
        a = raw[off:off + cnt].view(dtype).reshape(shape)
        off += cnt
        return a

    idx0, obs0 = take(np.int32, N), take(np.float32, N, 9)
    ref = oracle_c.Batch(N, 72, tab, oracle_c.profile(98))
    ref.set_state(obs0, idx0)
    i = np.arange(N)
    for t in range(NSTEPS):
        a = np.stack([((i * 37 + t * 11) % 101).astype(np.float32) / np.float32(100), ((i * 53 + t * 29) % 97).astype(np.float32) / np.float32(96)], 1)
        rew, obs, res = take(np.float64, N), take(np.float32, N, 9), take(np.float64, N, 23)
        rc, r_ref, o_ref, res_ref = ref.step(a, 1, want_results=True)
        assert rc == 0
        assert (U.bits64(rew) == U.bits64(r_ref)).all() and (U.bits32(obs) == U.bits32(o_ref)).all(), t
        assert (U.bits64(res) == U.bits64(res_ref)).all(), t
    rule = take(np.float32, N, 2)
    assert (U.bits32(rule) == U.bits32(ref.action_rule())).all()
    assert off == raw.size
