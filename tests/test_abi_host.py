"""CPU tests: the C-ABI library loads and exports what include/shems_hip.h declares; host logic
(tables, configs) behaves; without a GPU the product path fails loudly instead of falling back."""
import ctypes as C
import importlib
import os
import re
import subprocess

import numpy as np
import pytest

import util as U


def test_library_exports_every_declared_symbol(built_lib):
    S = U.pkg()
    L = S._capi.lib()
    names = S._capi.exported_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/shems_hip.h but not exported by libshems_hip.so"
    assert L.shems_abi_version() == 1
    out = subprocess.check_output(["nm", "-D", "--defined-only", built_lib]).decode()
    exported = set(re.findall(r" T (shems_[a-z0-9_]+)", out))
    assert set(names) <= exported
    # nothing torch-typed crosses the boundary: the library does not link libtorch
    deps = subprocess.check_output(["ldd", built_lib]).decode()
    assert "torch" not in deps and "amdhip64" in deps


def test_struct_layouts_match_header():
    S = U.pkg()
    assert C.sizeof(S._capi.Config) == 48
    assert S._capi.Config.rate_max.offset == 8 and S._capi.Config.penalty_weight.offset == 32
    assert S._capi.Config.table_row0.offset == 36 and S._capi.Config.nrow.offset == 40
    assert C.sizeof(S._capi.View) == 80 and S._capi.View.obs.offset == 16
    assert C.sizeof(S._capi.Replay) == 48
    assert C.sizeof(U.HCConfig) == 48


def test_train_loop_record_matches_the_c_struct(tmp_path):
    """shems_train_loop is passed by pointer and advanced in place: the ctypes mirror must have the C struct's offsets (gcc lays the
    header's declaration out; compared field by field)."""
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    names = [f[0] for f in D.TrainLoop._fields_]
    src = tmp_path / "o.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "shems_hip.h"\nint main(void){printf("%zu", sizeof(shems_train_loop));' +
                   "".join(f'printf(" %zu", offsetof(shems_train_loop, {n}));' for n in names) + 'return 0;}\n')
    exe = tmp_path / "o"
    subprocess.check_call(["gcc", "-I", os.path.join(U.ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(x) for x in subprocess.check_output([str(exe)]).decode().split()]
    assert vals[0] == C.sizeof(D.TrainLoop)
    assert vals[1:] == [getattr(D.TrainLoop, n).offset for n in names]
    assert (D.LOOP_ORDERED, D.LOOP_PIPELINED, D.LOOP_PIPELINED_EXACT) == (0, 1, 2)


def test_gpu_code_object_is_gfx950(built_lib):
    blob = open(built_lib, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob[:0]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present: covered by the gpu tests")
def test_no_device_fails_loudly(built_lib):
    S = U.pkg()
    with pytest.raises(S.ShemsError) as ei:
        S.ShemsBatch(4, 72, S.tables.synthetic_table("train"))
    assert ei.value.code == S._capi.ERR_NODEVICE and "no CPU path" in str(ei.value)


def test_make_config_values():
    S = U.pkg()
    c = S.make_config(98, 0, 4320)
    assert c.cap_ev == np.float32(35.816) and c.soc_max == np.float32(6.75) and c.rate_max == 3.3
    assert c.disc_weight == float(np.float32(0.01)) and c.disc_pot == 2.0 and c.penalty_weight == np.float32(0.1)
    for cid in U.CHARGER_IDS + [97]:
        p = U.oracle_c.profile(cid)
        k = S.make_config(cid, 0, 10)
        assert (k.cap_ev, k.soc_max, k.rate_max) == (p.cap_ev, p.soc_max, p.rate_max)
        q = U.onp.Profile(cid)
        assert (np.float32(k.cap_ev), np.float32(k.soc_max)) == (q.ev_soc_max, q.b_soc_max)


def test_synthetic_table_properties_and_csv_roundtrip(tmp_path):
    T = U.tables_mod()
    for split in ("train", "eval", "test"):
        t = T.synthetic_table(split, 98)
        assert t.shape == (T.SPLIT_ROWS[split], 8) and t.dtype == np.float32
        h = t[:, 0]
        assert h.min() == -1 and h.max() <= 71 and (h == np.round(h)).all()
        zero = np.where(h[:-1] == 0)[0]
        assert (h[zero + 1] == -1).all() and (t[zero + 1, 1] == 1).all()        # row after a 0 is -1 / soc 1
        assert 0.25 < (h >= 0).mean() < 0.5
        assert (t[h == -1, 1] == 1).all() and (t[:, 4] == np.float32(0.4)).all()
        assert set(np.unique(t[:, 7])) <= {1.0, 2.0, 3.0, 4.0}
        dec = np.where((h[1:] >= 0) & (h[:-1] >= 0))[0]
        assert (h[dec + 1] == h[dec] - 1).all()                                  # countdown decrements
    a, b = T.synthetic_table("train", 98), T.synthetic_table("train", 98)
    assert (a == b).all() and not (a == T.synthetic_table("train", 4)).all()
    p = tmp_path / "t.csv"
    T.save_csv(p, a)
    assert (U.bits32(T.load_csv(p)) == U.bits32(a)).all()
    # 21-column reference schema with extra columns in a different order
    hdr = "electkwh,PV_generation,chargekwh,h_countdown,soc_ev,month,day,hour,nday,d_res,hour_cos,hour_sin,month_cos,month_sin,spring,summer,autumn,winter,season,p_buy,p_sell"
    with open(tmp_path / "r.csv", "w") as fh:
        fh.write(hdr + "\n")
        for r in a[:50]:
            v = dict(zip(T.COLUMNS, r))
            fh.write(",".join(repr(float(v.get(c, 0.0))) for c in hdr.split(",")) + "\n")
    assert (T.load_csv(tmp_path / "r.csv") == a[:50]).all()
    with open(tmp_path / "bad.csv", "w") as fh:
        fh.write("electkwh,PV_generation\n1,2\n")
    with pytest.raises(KeyError):
        T.load_csv(tmp_path / "bad.csv")


_C_SCALARS = {"int": "Cint", "int32_t": "Int32", "int64_t": "Int64", "uint64_t": "UInt64", "uint32_t": "UInt32", "uint16_t": "UInt16",
              "uint8_t": "UInt8", "float": "Float32", "double": "Float64", "shems_config": "ShemsConfig", "void": "Cvoid",
              "shems_env": "Cvoid", "shems_view": "ShemsView", "shems_act_params": "ShemsActParams", "shems_replay": "ShemsReplay",
              "shems_ring_window": "ShemsRingWindow", "shems_ddpg": "ShemsDdpg", "shems_train_loop": "ShemsTrainLoop", "shems_dp": "Cvoid",
              "shems_group": "ShemsGroup", "shems_group_w2t": "ShemsGroupW2T"}


def _julia_types_for(c_arg):
    """The Julia ccall argument types that are ABI-identical to one C parameter declaration (name stripped).  `T *` accepts Ptr{T}
    and Ref{T}; an opaque `shems_env *` is Ptr{Cvoid}, `shems_env **` Ptr{Ptr{Cvoid}}."""
    import re
    c = re.sub(r"\bstruct\b", " ", re.sub(r"/\*.*?\*/", "", c_arg, flags=re.S).replace("const", " ")).strip()
    m = re.match(r"^(\w+)\s*(\*{0,2})\s*(\w*)$", c)
    assert m, c_arg
    base, stars = _C_SCALARS[m.group(1)], len(m.group(2))
    if stars == 0:
        return {base}
    inner = base if stars == 1 else f"Ptr{{{base}}}"
    return {f"Ptr{{{inner}}}", f"Ref{{{inner}}}"}


def test_julia_type_mapping_rejects_wrong_widths():
    assert _julia_types_for("int32_t maxsteps") == {"Int32"} and _julia_types_for("int device") == {"Cint"}
    assert "Int64" not in _julia_types_for("int32_t n_cfg") and "Int32" not in _julia_types_for("int track_mode")
    assert _julia_types_for("const uint16_t *cfg_of_env") == {"Ptr{UInt16}", "Ref{UInt16}"}
    assert _julia_types_for("shems_env **out") == {"Ptr{Ptr{Cvoid}}", "Ref{Ptr{Cvoid}}"}
    assert _julia_types_for("shems_env *env") == {"Ptr{Cvoid}", "Ref{Cvoid}"}
    assert _julia_types_for("const shems_config *cfgs") == {"Ptr{ShemsConfig}", "Ref{ShemsConfig}"}


def test_julia_module_binds_only_declared_entry_points(built_lib):
    """julia/ShemsEnv_LU1.jl cannot be executed here (no Julia); what CAN be checked statically: every `ccall((:symbol, LIB), ...)` names
    an entry point that include/shems_hip.h declares and libshems_hip.so exports, with the argument count of the declaration; the
    struct mirror has the 48-byte layout; the methods the reference's callers use are all defined."""
    import re
    S = U.pkg()
    src = open(os.path.join(U.ROOT, "julia", "ShemsEnv_LU1.jl")).read()
    hdr = open(os.path.join(U.ROOT, "include", "shems_hip.h")).read()
    L = S._capi.lib()
    calls = re.findall(r"ccall\(\(:(\w+), LIB\), (\w+), \(([^)]*)\)", src)
    assert len(calls) >= 10
    for name, ret, args in calls:
        assert hasattr(L, name), name
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", hdr, re.S)
        assert m, f"{name} is not declared in include/shems_hip.h"
        decl = [] if m.group(1).strip() in ("", "void") else [a.strip() for a in m.group(1).split(",")]
        call = [a.strip() for a in args.split(",") if a.strip()]
        assert len(decl) == len(call), (name, decl, call)
        # argument WIDTHS, position by position: the likeliest defect of a never-executed binding is a wrong integer width or pointee
        # in the ccall tuple (Int32 vs Cint vs Int64, Ptr{UInt16} vs Ptr{Int32}), which would still pass a count check
        for pos, (c_arg, j_arg) in enumerate(zip(decl, call)):
            assert j_arg in _julia_types_for(c_arg), (name, pos, c_arg, j_arg)
        c_ret = re.search(r"(const char \*|int)\s*" + name + r"\s*\(", hdr).group(1).strip()
        assert ret == {"int": "Cint", "const char *": "Cstring"}[c_ret], (name, c_ret, ret)
    for needed in ("struct ShemsConfig", "mutable struct ShemsAction", "Base.minimum(::ShemsAction) = (0f0, 0f0)", "Base.maximum(::ShemsAction) = (1f0, 1f0)",
                   "function reset!(env::Shems; rng=0)", "function step!(env::Shems, s, a; track=0)", "function action(env::Shems, a::ShemsAction)",
                   "function action(env::Shems, track::Real=-1)", "finished(env::Shems, s′) = false", "using Distributions: Uniform", "using Random",
                   "module ShemsEnv_LU1"):
        assert needed in src, needed
    # ShemsConfig mirror: Float32 x2, Float64 x3, Float32, Int32 x3 = 48 bytes in declaration order (the C struct of the header)
    fields = re.search(r"struct ShemsConfig(.*?)\nend", src, re.S).group(1)
    types = re.findall(r"::(\w+)", fields)
    assert types == ["Float32", "Float32", "Float64", "Float64", "Float64", "Float32", "Int32", "Int32", "Int32"]
    assert C.sizeof(S._capi.Config) == 48


def _c_struct_fields(hdr, name):
    """Julia types of the fields of `typedef struct <name> { ... } <name>;` in declaration order (`float *a, *b;` -> two Ptr{Float32})."""
    import re
    body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.replace("const", " ").strip()
        if not decl:
            continue
        m = re.match(r"^(\w+)\s+(.*)$", decl, re.S)
        base = _C_SCALARS[m.group(1)]
        for d in m.group(2).split(","):
            d = d.strip()
            t = f"Ptr{{{base}}}" if d.startswith("*") else base
            arr = re.search(r"\[(\d+)\]$", d)                      # `double bp[2]` / `float *pub[2]`: an inline array = NTuple{2, T}
            out.append(f"NTuple{{{arr.group(1)}, {t}}}" if arr else t)
    return out


def _julia_struct_fields(src, name):
    import re
    body = re.search(r"(?:^|\n)struct " + name + r"\b(.*?)\nend", src, re.S).group(1)
    body = re.sub(r"#.*", "", body)
    return re.findall(r"\w+::((?:NTuple\{\d+, )?[\w{}]+\}?)", body)


def test_julia_learner_module_matches_the_header(built_lib):
    """julia/DDPG_hip.jl -- the reference's DDPG.jl / memory_plotting_saving.jl functions over the device-pointer API -- cannot be executed
    here either.  Statically: every `ccall((:symbol, LIB), ...)` names a declared, exported entry point with ABI-identical argument and
    return types (position by position), every struct mirror has the field types of its C struct in order, and the functions the
    reference's entry script calls (DDPG_reinforce_charger_v1.jl:27-105) are defined."""
    import re
    S = U.pkg()
    src = open(os.path.join(U.ROOT, "julia", "DDPG_hip.jl")).read()
    hdr = open(os.path.join(U.ROOT, "include", "shems_hip.h")).read()
    L = S._capi.lib()
    calls = re.findall(r"ccall\(\(:(\w+), LIB\), (\w+),\s*\(([^)]*)\)", src)
    assert len(calls) >= 16 and {c[0] for c in calls} >= {"shems_act_step_dev", "shems_ddpg_update", "shems_rollout_dev", "shems_minmax_dev",
                                                           "shems_track_dev", "shems_get_view", "shems_reset_seeded_dev", "shems_train_steps",
                                                           "shems_train_loop_release",
                                                           # round 6: learner groups on the tiled working layout
                                                           "shems_act_step_group_tiled_dev", "shems_ddpg_group_update_tiled", "shems_group_w2_to_tiled",
                                                           "shems_group_w2_to_flux", "shems_minmax_group_dev"}
    for name, ret, args in calls:
        assert hasattr(L, name), name
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", hdr, re.S)
        assert m, f"{name} is not declared in include/shems_hip.h"
        decl = [] if m.group(1).strip() in ("", "void") else [a.strip() for a in m.group(1).split(",")]
        call = [a.strip() for a in args.split(",") if a.strip()]
        assert len(decl) == len(call), (name, decl, call)
        for pos, (c_arg, j_arg) in enumerate(zip(decl, call)):
            assert j_arg in _julia_types_for(c_arg), (name, pos, c_arg, j_arg)
        c_ret = re.search(r"(const char \*|int)\s*" + name + r"\s*\(", hdr).group(1).strip()
        assert ret == {"int": "Cint", "const char *": "Cstring"}[c_ret], (name, c_ret, ret)
    for cname, jname in (("shems_config", "ShemsConfig"), ("shems_view", "ShemsView"), ("shems_replay", "ShemsReplay"),
                         ("shems_act_params", "ShemsActParams"), ("shems_ring_window", "ShemsRingWindow"), ("shems_ddpg", "ShemsDdpg"),
                         ("shems_train_loop", "ShemsTrainLoop"), ("shems_group", "ShemsGroup"), ("shems_group_w2t", "ShemsGroupW2T")):
        assert _julia_struct_fields(src, jname) == _c_struct_fields(hdr, cname), (cname, _julia_struct_fields(src, jname), _c_struct_fields(hdr, cname))
    # the ctypes mirrors the tests drive have the same sizes as those field lists imply (8-byte pointers, natural alignment)
    D = importlib.import_module(U.PKG_NAME + ".ddpg")
    size = {"Float32": 4, "Int32": 4, "UInt32": 4, "Float64": 8, "Int64": 8, "UInt64": 8}
    def c_size(types):
        off = 0
        for t in types:
            n = 8 if t.startswith("Ptr{") else size[t]
            off = (off + n - 1) // n * n + n
        return (off + 7) // 8 * 8
    assert c_size(_c_struct_fields(hdr, "shems_act_params")) == C.sizeof(D.ActParams)
    assert c_size(_c_struct_fields(hdr, "shems_ddpg")) == C.sizeof(D.DdpgArgs)
    assert c_size(_c_struct_fields(hdr, "shems_view")) == C.sizeof(S._capi.View) and c_size(_c_struct_fields(hdr, "shems_replay")) == C.sizeof(S._capi.Replay)
    for needed in ("module DDPG_hip", "function act(ag::Agent", "function act_step!(ag::Agent", "function replay(ag::Agent", "function populate_memory(ag::Agent",
                   "function min_max_buffer(ag::Agent", "function episode!(ag::Agent", "function run_episodes(ag::Agent", "function inference(env::EnvBatch",
                   "function train_steps!(ag::Agent",
                   "flat_params(params)", "mutable struct LearnerGroup", "function act_step!(g::LearnerGroup", "function replay(g::LearnerGroup",
                   "function episode!(g::LearnerGroup", "function flux!(g::LearnerGroup", "function min_max_buffer(g::LearnerGroup",
                   "function populate_memory(g::LearnerGroup"):
        assert needed in src, needed
