"""BSON interchange of the actor (memory_plotting_saving.jl:263-281).  "Parity unpinned": no BSON file written by
Julia exists in the reference (all are git-LFS stubs).  What is tested: the wire codec against hand-assembled bytes of the BSON
specification, the BSON.jl lowering layout as restated, the reader's independence from type descriptors and its backref handling."""
import importlib
import struct

import numpy as np
import pytest

import util as U
import ddpg_oracle as DO

B = importlib.import_module(U.PKG_NAME + ".bson_chain")


def test_wire_format_known_bytes():
    # {"hello": "world"} -- the example of bsonspec.org
    ref = b"\x16\x00\x00\x00\x02hello\x00\x06\x00\x00\x00world\x00\x00"
    assert B.encode_document({"hello": "world"}) == ref
    assert B.decode_document(ref)[0] == {"hello": "world"}
    # {"BSON": ["awesome", 5.05, 1986]} -- second example of the specification
    ref2 = (b"\x31\x00\x00\x00\x04BSON\x00\x26\x00\x00\x00\x020\x00\x08\x00\x00\x00awesome\x00\x011\x00\x33\x33\x33\x33\x33\x33\x14\x40"
            b"\x102\x00\xc2\x07\x00\x00\x00\x00")
    assert B.encode_document({"BSON": ["awesome", 5.05, 1986]}) == ref2
    assert B.decode_document(ref2)[0] == {"BSON": ["awesome", 5.05, 1986]}
    d = {"b": b"\x00\x01\xff", "i64": 2 ** 40, "t": True, "n": None, "nested": {"x": [1, [2.5, "s"]]}}
    assert B.decode_document(B.encode_document(d))[0] == d
    with pytest.raises(ValueError):
        B.decode_document(b"\x05\x00\x00\x00\x01")


def test_chain_layout_and_roundtrip(tmp_path):
    actor = DO.init_params(1231, 9, 2, 0)
    p = B.save_chain(str(tmp_path / "a.bson"), actor)
    doc = B.read_file(p)
    ch = doc["actor"]
    assert ch["tag"] == "struct" and ch["type"]["name"] == ["Flux", "Chain"] and ch["data"][0]["tag"] == "tuple"
    layers = ch["data"][0]["data"]
    assert [l["type"]["name"] for l in layers] == [["Flux", "Dense"]] * 3
    assert [l["type"]["params"][0]["name"] for l in layers] == [["NNlib", "#relu"], ["NNlib", "#relu"], ["Base", "#tanh"]]
    W1 = layers[0]["data"][0]
    assert W1["tag"] == "array" and W1["size"] == [250, 9] and W1["type"]["name"] == ["Core", "Float32"] and len(W1["data"]) == 250 * 9 * 4
    # Julia's out x in column-major matrix: element (o, i) at byte (i * 250 + o) * 4 == this package's [in][out] block
    Wj = B.raise_array(W1)
    assert Wj.shape == (250, 9) and Wj[7, 3] == actor[3 * 250 + 7]
    assert (B.load_chain(p) == actor).all()
    critic = DO.init_params(1231, 11, 1, 1)
    pc = B.save_chain(str(tmp_path / "c.bson"), critic, 11, 1, "identity", key="critic")
    assert (B.load_chain(pc, key="critic") == critic).all()
    with pytest.raises(KeyError):
        B.load_chain(pc, key="actor")
    with pytest.raises(ValueError):
        B.save_chain(str(tmp_path / "x.bson"), actor[:-1])
    (tmp_path / "stub.bson").write_bytes(b"version https://git-lfs.github.com/spec/v1\noid sha256:00\nsize 1\n")
    with pytest.raises(ValueError):
        B.read_file(str(tmp_path / "stub.bson"))


def test_reader_resolves_backrefs_and_ignores_type_descriptors(tmp_path):
    """BSON.jl replaces repeated mutable values -- the DataType descriptors among them -- by {tag: backref, ref: i} into "_backrefs"
    (1-based).  The reader must load such a file, whatever the descriptors say."""
    actor = DO.init_params(7, 9, 2, 0)
    low = B.lower_chain(actor, 9, 2, "tanh")
    f32 = {"tag": "datatype", "params": [], "name": ["Core", "Float32"]}
    refs = [f32, {"tag": "datatype", "params": ["whatever"], "name": ["Some", "Other", "Module", "Dense"]}]

    def rewrite(n):
        if isinstance(n, dict):
            if n.get("tag") == "array":
                return {**n, "type": {"tag": "backref", "ref": 1}}
            if n.get("tag") == "struct" and n["type"]["name"] == ["Flux", "Dense"]:
                return {"tag": "struct", "type": {"tag": "backref", "ref": 2}, "data": [rewrite(x) for x in n["data"]]}
            return {k: rewrite(v) for k, v in n.items()}
        if isinstance(n, list):
            return [rewrite(x) for x in n]
        return n

    p = tmp_path / "refs.bson"
    p.write_bytes(B.encode_document({"_backrefs": refs, "actor": rewrite(low)}))
    assert (B.load_chain(str(p)) == actor).all()


def test_scores_roundtrip(tmp_path):
    tr, sm, nm = np.arange(5, dtype=np.float32), np.array([1.5, -2.0]), np.zeros(5, np.float32)
    p = B.save_scores(str(tmp_path / "s.bson"), tr, sm, 401, nm)
    tr2, sm2, best, nm2 = B.load_scores(p)
    assert tr2.dtype == np.float32 and sm2.dtype == np.float64 and (tr2 == tr).all() and (sm2 == sm).all() and best == 401 and (nm2 == nm).all()
    doc = B.read_file(p)
    assert doc["score_mean"]["type"]["name"] == ["Core", "Float64"] and doc["best_run"] == 401
