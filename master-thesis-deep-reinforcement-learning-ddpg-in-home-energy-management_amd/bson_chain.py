"""Actor / score snapshots in the reference's container: BSON files as BSON.jl writes them.

Reference: `BSON.@save ".../..._actor_<idx>.bson" actor` with `actor = cpu(Chain(Dense, Dense, Dense))`, and
`BSON.@save ".../..._scores_<idx>.bson" total_reward score_mean best_run noise_mean` (memory_plotting_saving.jl:263-281; BSON.jl
0.3.x, Flux 0.12.1 per Manifest.toml).

PARITY STATUS: "parity unpinned".  Every .bson file shipped with the reference is a git-LFS stub and there is no Julia here to write
one, so the document layout below is restated from BSON.jl's published lowering rules and has never met a file written by Julia:
    struct    -> {tag: "struct",   type: <datatype>, data: [fields...]}
    DataType  -> {tag: "datatype", name: [module path..., type name], params: [...]}
    Tuple     -> {tag: "tuple",    data: [...]}
    Array{T}  -> {tag: "array",    type: <datatype of T>, size: [dims...], data: <binary, column-major, little endian>}   (bits types)
    shared mutable values (BSON.jl also treats the repeated DataType descriptors so) -> {tag: "backref", ref: i} into the
    top-level "_backrefs" array.
The READER does not depend on the fine print: it resolves backrefs and takes the `array` leaves of the saved value in traversal
order -- for a Chain of three Dense layers that is W1, b1, W2, b2, W3, b3 = Flux.params order -- so a file that deviates from the
restated layout in its type descriptors still loads.  The WRITER emits the restated layout without backrefs (BSON.jl accepts
inline descriptors).  A Julia `out x in` column-major matrix is byte for byte this package's `[in][out]` C-order block, so
parameters move without transposition.

The BSON wire format itself (bsonspec.org) is encoded / decoded here directly: documents, arrays, strings, binary, doubles, int32 /
int64, booleans, null -- the subset BSON.jl produces.
"""
from __future__ import annotations

import struct

import numpy as np

L1, L2 = 250, 500


# ------------------------------------------------------------------------------------------------ wire format --
def _enc_cstring(s):
    b = s.encode("utf-8")
    if b"\x00" in b:
        raise ValueError("BSON key contains NUL")
    return b + b"\x00"


def _enc_value(v):
    if isinstance(v, bool):
        return b"\x08", b"\x01" if v else b"\x00"
    if isinstance(v, (int, np.integer)):
        v = int(v)
        return (b"\x10", struct.pack("<i", v)) if -2 ** 31 <= v < 2 ** 31 else (b"\x12", struct.pack("<q", v))
    if isinstance(v, (float, np.floating)):
        return b"\x01", struct.pack("<d", float(v))
    if isinstance(v, str):
        b = v.encode("utf-8") + b"\x00"
        return b"\x02", struct.pack("<i", len(b)) + b
    if isinstance(v, (bytes, bytearray, memoryview)):
        b = bytes(v)
        return b"\x05", struct.pack("<i", len(b)) + b"\x00" + b
    if v is None:
        return b"\x0a", b""
    if isinstance(v, dict):
        return b"\x03", encode_document(v)
    if isinstance(v, (list, tuple)):
        return b"\x04", encode_document({str(i): x for i, x in enumerate(v)})
    raise TypeError(f"cannot encode {type(v)} as BSON")


def encode_document(d):
    body = b""
    for k, v in d.items():
        t, payload = _enc_value(v)
        body += t + _enc_cstring(str(k)) + payload
    return struct.pack("<i", len(body) + 5) + body + b"\x00"


def decode_document(buf, pos=0, as_list=False):
    (size,) = struct.unpack_from("<i", buf, pos)
    end = pos + size
    if size < 5 or end > len(buf) or buf[end - 1] != 0:
        raise ValueError("malformed BSON document")
    pos += 4
    out = {}
    while pos < end - 1:
        t = buf[pos]
        pos += 1
        z = buf.index(b"\x00", pos)
        key = buf[pos:z].decode("utf-8")
        pos = z + 1
        if t == 0x01:
            val = struct.unpack_from("<d", buf, pos)[0]; pos += 8
        elif t == 0x02:
            (n,) = struct.unpack_from("<i", buf, pos); val = buf[pos + 4:pos + 4 + n - 1].decode("utf-8"); pos += 4 + n
        elif t == 0x03:
            val, pos = decode_document(buf, pos)
        elif t == 0x04:
            val, pos = decode_document(buf, pos, as_list=True)
        elif t == 0x05:
            (n,) = struct.unpack_from("<i", buf, pos); val = bytes(buf[pos + 5:pos + 5 + n]); pos += 5 + n
        elif t == 0x08:
            val = buf[pos] != 0; pos += 1
        elif t == 0x0A:
            val = None
        elif t == 0x10:
            val = struct.unpack_from("<i", buf, pos)[0]; pos += 4
        elif t == 0x12:
            val = struct.unpack_from("<q", buf, pos)[0]; pos += 8
        else:
            raise ValueError(f"BSON element type 0x{t:02x} is not produced by BSON.jl")
        out[key] = val
    if as_list:
        return [out[k] for k in sorted(out, key=int)], end
    return out, end


# ------------------------------------------------------------------------------------------ BSON.jl lowering --
def _datatype(*path, params=()):
    return {"tag": "datatype", "params": list(params), "name": list(path)}


_F32 = _datatype("Core", "Float32")
_F64 = _datatype("Core", "Float64")
_DTYPES = {("Core", "Float32"): np.float32, ("Core", "Float64"): np.float64, ("Core", "Int64"): np.int64, ("Core", "Int32"): np.int32,
           ("Core", "Bool"): np.bool_, ("Core", "UInt8"): np.uint8}


def lower_array(a, eltype=None):
    a = np.asarray(a)
    path = {np.dtype(np.float32): ("Core", "Float32"), np.dtype(np.float64): ("Core", "Float64"), np.dtype(np.int64): ("Core", "Int64"),
            np.dtype(np.int32): ("Core", "Int32")}[a.dtype] if eltype is None else eltype
    # Julia arrays are column-major: dims as Julia sees them, bytes in Fortran order
    return {"tag": "array", "type": _datatype(*path), "size": [int(n) for n in a.shape], "data": np.asfortranarray(a).tobytes(order="F")}


def _array_type(nd):
    return _datatype("Core", "Array", params=[_F32, nd])


def _fn(*path):
    """A singleton function value, e.g. relu: a struct of type typeof(relu) with no fields."""
    return {"tag": "struct", "type": _datatype(*path), "data": []}


def lower_dense(W_in_out, b, act):
    """Dense(W, b, sigma) of Flux 0.12.1 (fields weight, bias, sigma).  W_in_out: this package's [in][out] block = Julia's out x in matrix."""
    in_dim, out_dim = W_in_out.shape
    fpath = {"relu": ("NNlib", "#relu"), "tanh": ("Base", "#tanh"), "identity": ("Base", "#identity")}[act]
    W = {"tag": "array", "type": _F32, "size": [out_dim, in_dim], "data": np.ascontiguousarray(W_in_out, np.float32).tobytes()}
    return {"tag": "struct",
            "type": _datatype("Flux", "Dense", params=[_datatype(*fpath), _array_type(2), _array_type(1)]),
            "data": [W, lower_array(np.asarray(b, np.float32)), _fn(*fpath)]}


def lower_chain(flat, in_dim, out_dim, final_act, hidden=None):
    L1, L2 = hidden if hidden is not None else (globals()["L1"], globals()["L2"])     # Dense widths of THIS chain (default: the tuned 250, 500)
    flat = np.asarray(flat, np.float32).reshape(-1)
    sizes = [in_dim * L1, L1, L1 * L2, L2, L2 * out_dim, out_dim]
    if flat.size != sum(sizes):
        raise ValueError(f"expected {sum(sizes)} parameters, got {flat.size}")
    o = np.cumsum([0] + sizes)
    blocks = [flat[o[i]:o[i + 1]] for i in range(6)]
    layers = [lower_dense(blocks[0].reshape(in_dim, L1), blocks[1], "relu"), lower_dense(blocks[2].reshape(L1, L2), blocks[3], "relu"),
              lower_dense(blocks[4].reshape(L2, out_dim), blocks[5], final_act)]
    tup_t = _datatype("Core", "Tuple", params=[l["type"] for l in layers])
    return {"tag": "struct", "type": _datatype("Flux", "Chain", params=[tup_t]), "data": [{"tag": "tuple", "data": layers}]}


# ------------------------------------------------------------------------------------------------- raising --
def _resolve(x, refs):
    if isinstance(x, dict):
        if x.get("tag") == "backref":
            return _resolve(refs[int(x["ref"]) - 1], refs)          # Julia indices are 1-based
        return {k: _resolve(v, refs) for k, v in x.items()}
    if isinstance(x, list):
        return [_resolve(v, refs) for v in x]
    return x


def raise_array(node):
    name = tuple(node["type"]["name"])
    if name not in _DTYPES:
        raise ValueError(f"array element type {name} is not a bits type this reader knows")
    shape = [int(n) for n in node["size"]]
    a = np.frombuffer(node["data"], dtype=_DTYPES[name])
    if a.size != int(np.prod(shape)) if shape else a.size != 1:
        raise ValueError("array payload does not match its size")
    return a.reshape(shape, order="F") if shape else a.reshape(())


def arrays_in_order(node):
    """The `array` leaves of a lowered value, depth first, fields in declaration order."""
    out = []
    if isinstance(node, dict):
        if node.get("tag") == "array" and isinstance(node.get("data"), (bytes, bytearray)):
            return [raise_array(node)]
        for k in ("data",) if "tag" in node else node.keys():
            if k in node:
                out += arrays_in_order(node[k])
    elif isinstance(node, list):
        for v in node:
            out += arrays_in_order(v)
    return out


def read_file(path):
    buf = open(path, "rb").read()
    if buf.startswith(b"version https://git-lfs"):
        raise ValueError(f"{path} is a git-LFS pointer, not a BSON file")
    doc, _ = decode_document(buf)
    refs = doc.pop("_backrefs", [])
    return {k: _resolve(v, refs) for k, v in doc.items()}


def load_chain(path, key="actor", hidden=None):
    """-> flat float32 parameter vector in Flux.params order (W1 b1 W2 b2 W3 b3), each W as this package's [in][out] block.
    hidden: the (L1, L2) the file must hold (default: the tuned 250, 500)."""
    L1, L2 = hidden if hidden is not None else (globals()["L1"], globals()["L2"])
    doc = read_file(path)
    if key not in doc:
        raise KeyError(f"{path} holds {sorted(doc)}, not {key!r}")
    arrs = arrays_in_order(doc[key])
    if len(arrs) != 6 or [a.ndim for a in arrs] != [2, 1, 2, 1, 2, 1]:
        raise ValueError(f"{path}: expected the six arrays of Chain(Dense, Dense, Dense), found shapes {[a.shape for a in arrs]}")
    (W1, b1, W2, b2, W3, b3) = arrs
    if W1.shape[0] != L1 or W2.shape != (L2, L1) or W3.shape[1] != L2 or b1.shape != (L1,) or b2.shape != (L2,) or b3.shape != (W3.shape[0],):
        raise ValueError(f"{path}: not a (in -> {L1} -> {L2} -> out) chain: {[a.shape for a in arrs]}")
    # Julia out x in, column-major == C-order [in][out]: ravel in Fortran order
    return np.concatenate([a.astype(np.float32).ravel(order="F") for a in arrs])


def save_chain(path, flat, in_dim=9, out_dim=2, final_act="tanh", key="actor", hidden=None):
    with open(path, "wb") as fh:
        fh.write(encode_document({key: lower_chain(flat, in_dim, out_dim, final_act, hidden)}))
    return path


def save_scores(path, total_reward, score_mean, best_run, noise_mean):
    """BSON.@save path total_reward score_mean best_run noise_mean  (Vector{Float32}, Vector{Float64}, Int, Vector{Float32})."""
    doc = {"total_reward": lower_array(np.asarray(total_reward, np.float32)), "score_mean": lower_array(np.asarray(score_mean, np.float64)),
           "best_run": int(best_run), "noise_mean": lower_array(np.asarray(noise_mean, np.float32))}
    with open(path, "wb") as fh:
        fh.write(encode_document(doc))
    return path


def load_scores(path):
    doc = read_file(path)
    g = lambda k: raise_array(doc[k]) if isinstance(doc[k], dict) else doc[k]
    return g("total_reward"), g("score_mean"), int(g("best_run")), g("noise_mean")
