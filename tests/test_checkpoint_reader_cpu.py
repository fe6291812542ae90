"""Native checkpoint reader (csrc/ckpt_reader.cpp through the C ABI) against torch.load / safetensors on the files the
loader reads (reference: modelcompose/model/builder.py:148, :157-168): torch zip checkpoints - flat state dicts, OrderedDict with
_metadata, nn.Parameter values, shared storages with offsets, non-contiguous views, every dtype, empty tensors, sharded files with an
index - and safetensors; bit-exact, zero-copy; hostile / malformed files are refused without executing anything."""
import collections
import ctypes as C
import json
import os
import pickle
import struct

import pytest
import torch

from modelcompose_amd import _lib
from modelcompose_amd.checkpoint_io import MappedCheckpoint, load_nested, load_tensors


def same(a: torch.Tensor, b: torch.Tensor):
    assert a.dtype == b.dtype and tuple(a.shape) == tuple(b.shape), (a.dtype, b.dtype, a.shape, b.shape)
    if a.numel():
        assert torch.equal(a.contiguous().reshape(-1).view(torch.uint8), b.contiguous().reshape(-1).view(torch.uint8))


def test_flat_state_dict_all_dtypes_bit_exact(tmp_path):
    g = torch.Generator().manual_seed(0)
    sd = {"f32": torch.randn(7, 33, generator=g), "f16": torch.randn(128, 64, generator=g).half(), "bf16": torch.randn(5, 4, 3, generator=g).bfloat16(),
          "f64": torch.randn(3, generator=g, dtype=torch.float64), "i64": torch.randint(-2 ** 40, 2 ** 40, (9,), generator=g),
          "i32": torch.randint(-1000, 1000, (4, 4), generator=g, dtype=torch.int32), "i16": torch.arange(-5, 5, dtype=torch.int16),
          "i8": torch.arange(-8, 8, dtype=torch.int8), "u8": torch.arange(0, 200, dtype=torch.uint8), "bool": torch.tensor([True, False, True]),
          "scalar": torch.tensor(3.5), "empty": torch.empty(0, 4), "base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight": torch.randn(8, 16, generator=g)}
    p = tmp_path / "adapter_model.bin"
    torch.save(sd, p)
    got = load_tensors(str(p))
    ref = torch.load(p, map_location="cpu", weights_only=True)
    assert list(got) == list(ref)                                   # insertion order preserved
    for k in ref:
        same(got[k], ref[k])


def test_ordered_dict_parameters_shared_storage_and_views(tmp_path):
    lin = torch.nn.Linear(6, 4)
    big = torch.arange(48, dtype=torch.float32).reshape(6, 8)
    sd = collections.OrderedDict()
    sd._metadata = {"": {"version": 1}}                            # what nn.Module.state_dict() attaches (pickled through BUILD)
    sd["weight"] = lin.weight                                       # nn.Parameter: _rebuild_parameter(_rebuild_tensor_v2(...))
    sd["bias"] = lin.bias
    sd["view_rows"] = big[2:5]                                      # storage offset 16, same storage as 'whole'
    sd["whole"] = big
    sd["transposed"] = big.t()                                      # non-contiguous strides (1, 8)
    sd["strided"] = big[:, ::2]
    p = tmp_path / "non_lora_trainables.bin"
    torch.save(sd, p)
    got = load_tensors(str(p))
    ref = torch.load(p, map_location="cpu", weights_only=True)
    assert list(got) == list(ref)
    for k in ref:
        same(got[k], ref[k].detach())
    assert got["transposed"].stride() == (1, 8) and got["view_rows"].storage_offset() >= 0
    # zero copy: the views alias one mapping of the file
    assert got["view_rows"].data_ptr() == got["whole"].data_ptr() + 16 * 4


def test_nested_containers_are_flattened_and_scalars_skipped(tmp_path):
    obj = {"model": {"a": torch.ones(2), "b": {"c": torch.zeros(3, dtype=torch.int64)}}, "step": 7, "lr": 1e-3, "name": "x", "flag": True, "none": None,
           "list": [1, 2, 3], "tuple": (torch.ones(1),)}
    p = tmp_path / "ckpt.bin"
    torch.save(obj, p)
    got = load_tensors(str(p))
    assert set(got) == {"model.a", "model.b.c", "tuple.0"}
    same(got["model.a"], torch.ones(2))
    same(got["model.b.c"], torch.zeros(3, dtype=torch.int64))
    # the tree form keeps the scalar leaves (BEATs files are {'cfg': {...}, 'model': state_dict})
    tree = load_nested(str(p))
    assert tree["step"] == 7 and tree["lr"] == 1e-3 and tree["name"] == "x" and tree["flag"] is True and tree["none"] is None
    assert tree["list"] == [1, 2, 3] and isinstance(tree["tuple"], list) and torch.equal(tree["tuple"][0], torch.ones(1))
    same(tree["model"]["b"]["c"], torch.zeros(3, dtype=torch.int64))


def test_empty_containers_keep_their_place_and_type(tmp_path):
    """a BEATs `cfg` entry holding [] or () must not fall back to the config default without notice."""
    obj = {"cfg": {"conv_bias": [], "layers": (), "extra": {}, "n": 3, "nested": [[], [1]]}, "model": {"w": torch.ones(2)}}
    p = tmp_path / "e.bin"
    torch.save(obj, p)
    tree = load_nested(str(p))
    assert tree["cfg"]["conv_bias"] == [] and isinstance(tree["cfg"]["conv_bias"], list)
    assert tree["cfg"]["layers"] == () and isinstance(tree["cfg"]["layers"], tuple)
    assert tree["cfg"]["extra"] == {} and tree["cfg"]["n"] == 3
    assert tree["cfg"]["nested"] == [[], [1]]
    assert set(load_tensors(str(p))) == {"model.w"}


def test_safetensors_without_the_suffix_goes_to_the_native_reader(tmp_path):
    """only files that carry torch's legacy magic number are legacy files."""
    from safetensors.torch import save_file
    from modelcompose_amd.checkpoint_io import _is_legacy_torch_file
    q = tmp_path / "weights.bin"
    save_file({"w": torch.arange(6, dtype=torch.float32).reshape(2, 3)}, str(q))
    assert not _is_legacy_torch_file(str(q))
    same(load_tensors(str(q))["w"], torch.arange(6, dtype=torch.float32).reshape(2, 3))
    old = tmp_path / "old.bin"
    torch.save({"w": torch.ones(3)}, old, _use_new_zipfile_serialization=False)
    assert _is_legacy_torch_file(str(old))
    same(load_tensors(str(old))["w"], torch.ones(3))


def test_keys_with_dots_and_numeric_keys_survive_the_tree_form(tmp_path):
    obj = {"cfg": {"a.b": 1, "neg": -5, "big": 2 ** 40, "f": -0.25, "s": "caf\u00e9", "empty": ""}, "model": {"enc.0.weight": torch.ones(2), "0": torch.zeros(1)},
           3: {"x": 2.5}}
    p = tmp_path / "t.bin"
    torch.save(obj, p)
    tree = load_nested(str(p))
    assert tree["cfg"] == {"a.b": 1, "neg": -5, "big": 2 ** 40, "f": -0.25, "s": "caf\u00e9", "empty": ""}
    assert set(tree["model"]) == {"enc.0.weight", "0"} and isinstance(tree["model"], dict)       # a dict keyed "0" is not a list
    assert tree["3"] == {"x": 2.5}


def test_beats_checkpoint_layout_loads_through_the_encoder_class(tmp_path):
    """the reference loads BEATs as checkpoint['cfg'] / checkpoint['model'] (beats/BEATs.py:120-148,
    multimodal_encoder/audio_encoder.py:20-31).  The class must find both in a real-layout file; the state dict is loaded on the GPU only,
    so here the config half is asserted and the weights half is checked to arrive under their un-prefixed names."""
    from modelcompose_amd.model.encoders_extra import BeatsConfig, HipBeatsAudioEncoder
    cfg = {"encoder_layers": 2, "encoder_embed_dim": 64, "encoder_ffn_embed_dim": 128, "encoder_attention_heads": 4, "activation_fn": "gelu",
           "deep_norm": True, "relative_position_embedding": True, "gru_rel_pos": True, "input_patch_size": 16, "embed_dim": 32,
           "num_buckets": 320, "max_distance": 800, "conv_pos": 128, "conv_pos_groups": 16, "dropout": 0.1, "layer_norm_first": False, "conv_bias": False}
    model = {"patch_embedding.weight": torch.randn(32, 1, 16, 16), "layer_norm.weight": torch.ones(32), "layer_norm.bias": torch.zeros(32),
             "encoder.layers.0.self_attn.k_proj.weight": torch.randn(64, 64)}
    p = tmp_path / "BEATs_iter3_plus_AS2M_finetuned_on_AS2M_cpt2.pt"
    torch.save({"cfg": cfg, "model": model}, p)
    ck = HipBeatsAudioEncoder._read_checkpoint(str(p))
    assert ck["cfg"] == cfg
    assert list(ck["model"]) == list(model)
    for k in model:
        same(ck["model"][k], model[k])
    c = BeatsConfig(ck["cfg"])
    assert c.deep_norm is True and c.gru_rel_pos is True and c.input_patch_size == 16 and c.encoder_layers == 2 and c.max_distance == 800
    enc = HipBeatsAudioEncoder(str(p), delay_load=True, device="cpu")           # constructor path: config from the file, weights deferred
    assert enc.cfg.encoder_embed_dim == 64 and enc.hidden_size == 64 and not enc.is_loaded
    q = tmp_path / "flat.pt"
    torch.save(model, q)
    with pytest.raises(KeyError, match="'cfg' and 'model'"):
        HipBeatsAudioEncoder._read_checkpoint(str(q))


def test_sharded_base_checkpoint_through_the_builder(tmp_path):
    from modelcompose_amd.model.builder import load_base_state_dict
    g = torch.Generator().manual_seed(1)
    sd = {f"model.layers.{i}.w": torch.randn(16, 8, generator=g).half() for i in range(6)}
    keys = sorted(sd)
    shards = {"pytorch_model-00001-of-00002.bin": keys[:3], "pytorch_model-00002-of-00002.bin": keys[3:]}
    for fn, ks in shards.items():
        torch.save({k: sd[k] for k in ks}, tmp_path / fn)
    json.dump({"weight_map": {k: fn for fn, ks in shards.items() for k in ks}}, open(tmp_path / "pytorch_model.bin.index.json", "w"))
    got = load_base_state_dict(str(tmp_path))
    assert sorted(got) == keys
    for k in keys:
        same(got[k], sd[k])


def test_safetensors_bit_exact(tmp_path):
    from safetensors.torch import save_file
    g = torch.Generator().manual_seed(2)
    sd = {"a.weight": torch.randn(32, 16, generator=g).bfloat16(), "b": torch.randn(5, generator=g), "c": torch.randint(0, 9, (3, 3), generator=g),
          "d": torch.randn(2, 2, 2, generator=g).half(), "e": torch.tensor([True, False])}
    p = tmp_path / "model.safetensors"
    save_file(sd, str(p), metadata={"format": "pt", "note": 'with "quotes" and \\ slashes'})
    got = load_tensors(str(p))
    assert set(got) == set(sd)
    for k in sd:
        same(got[k], sd[k])


def test_mapping_outlives_the_reader_object(tmp_path):
    p = tmp_path / "x.bin"
    torch.save({"w": torch.arange(1000, dtype=torch.float32)}, p)
    t = load_tensors(str(p))["w"]                                   # the MappedCheckpoint object itself is gone here
    import gc
    gc.collect()
    assert float(t.sum()) == 499500.0
    dev_copy = t.clone()
    t += 1                                                          # copy-on-write mapping: the file is not modified
    assert float(torch.load(p, weights_only=True)["w"].sum()) == 499500.0 and float(dev_copy.sum()) == 499500.0


class _Evil:
    def __reduce__(self):
        return (os.system, ("touch /tmp/mc_ckpt_pwned",))


def test_hostile_and_malformed_files_are_refused_without_executing(tmp_path):
    # a checkpoint whose pickle asks to call os.system: the native interpreter never calls anything, the object is inert
    marker = "/tmp/mc_ckpt_pwned"
    if os.path.exists(marker):
        os.remove(marker)
    p = tmp_path / "evil.bin"
    torch.save({"w": torch.ones(2), "boom": _Evil()}, p)
    got = load_tensors(str(p))
    assert set(got) == {"w"} and not os.path.exists(marker)
    with pytest.raises(Exception):
        torch.load(p, map_location="cpu", weights_only=True)        # torch's restricted unpickler rejects the same file
    # legacy (non-zip) torch format: the native reader refuses it, load_tensors falls back to torch's own restricted unpickler
    q = tmp_path / "legacy.bin"
    torch.save({"w": torch.arange(3.0), "n": {"v": torch.ones(2)}}, q, _use_new_zipfile_serialization=False)
    h = C.c_void_p(0)
    assert _lib.lib().mc_ckpt_open(str(q).encode(), C.byref(h)) == 1 and b"neither a zip" in _lib.lib().mc_last_error()
    leg = load_tensors(str(q))
    assert set(leg) == {"w", "n.v"} and torch.equal(leg["w"], torch.arange(3.0))
    q2 = tmp_path / "legacy_evil.bin"
    torch.save({"w": torch.ones(2), "boom": _Evil()}, q2, _use_new_zipfile_serialization=False)
    with pytest.raises(Exception):
        load_tensors(str(q2))
    assert not os.path.exists(marker)
    # truncated archive
    data = open(p, "rb").read()
    r = tmp_path / "trunc.bin"
    open(r, "wb").write(data[:len(data) // 2])
    with pytest.raises(ValueError):
        load_tensors(str(r))
    # tensor reaching outside its storage record: patch the numel of the storage is not enough, corrupt the shape in the pickle instead
    s = tmp_path / "liar.safetensors"
    hdr = json.dumps({"w": {"dtype": "F32", "shape": [1000], "data_offsets": [0, 16]}}).encode()
    open(s, "wb").write(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 16)
    with pytest.raises(ValueError, match="inconsistent"):
        load_tensors(str(s))
    with pytest.raises(ValueError, match="cannot open"):
        load_tensors(str(tmp_path / "missing.bin"))


def test_c_abi_enumeration_directly(tmp_path):
    p = tmp_path / "two.bin"
    torch.save({"a": torch.arange(6, dtype=torch.int32).reshape(2, 3), "b": torch.ones(4).half()}, p)
    L = _lib.lib()
    h = C.c_void_p(0)
    assert L.mc_ckpt_open(str(p).encode(), C.byref(h)) == 0
    n = C.c_int(0)
    assert L.mc_ckpt_count(h, C.byref(n)) == 0 and n.value == 2
    name, dt, nd = C.c_char_p(), C.c_int(0), C.c_int(0)
    shp, strd, data, sb = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_void_p(0), C.c_int64(0)
    assert L.mc_ckpt_entry(h, 0, C.byref(name), C.byref(dt), C.byref(nd), C.byref(shp), C.byref(strd), C.byref(data), C.byref(sb)) == 0
    assert name.value == b"a" and dt.value == 5 and nd.value == 2 and [shp[0], shp[1], strd[0], strd[1]] == [2, 3, 3, 1] and sb.value == 24
    assert list((C.c_int32 * 6).from_address(data.value)) == [0, 1, 2, 3, 4, 5]
    assert L.mc_ckpt_entry(h, 2, None, None, None, None, None, None, None) == 1
    assert L.mc_ckpt_close(h) == 0


# ---- crafted files: every size / offset / count of an untrusted file must be checked without wrapping -------------
def _zip_parts(path):
    """-> (bytes, eocd offset, [central directory entry offsets])."""
    b = bytearray(open(path, "rb").read())
    eocd = b.rfind(b"PK\x05\x06")
    total, cd_off = struct.unpack_from("<H", b, eocd + 10)[0], struct.unpack_from("<I", b, eocd + 16)[0]
    offs, p = [], cd_off
    for _ in range(total):
        assert b[p:p + 4] == b"PK\x01\x02"
        nlen, xlen, clen = struct.unpack_from("<HHH", b, p + 28)
        offs.append(p)
        p += 46 + nlen + xlen + clen
    return b, eocd, offs


def _entry_name(b, p):
    return bytes(b[p + 46:p + 46 + struct.unpack_from("<H", b, p + 28)[0]]).decode()


def _refused(path, match=None):
    h = C.c_void_p(0)
    rc = _lib.lib().mc_ckpt_open(str(path).encode(), C.byref(h))
    msg = _lib.lib().mc_last_error().decode()
    assert rc == 1 and not h.value, (rc, msg)
    if match:
        assert match in msg, msg


@pytest.mark.parametrize("usize", [2 ** 64 - 1, 2 ** 64 - 64, 2 ** 63, 2 ** 40])
def test_zip64_record_size_near_the_top_of_the_range_is_refused(tmp_path, usize):
    """A zip64 extended-information field with an uncompressed size of ~2^64: `data + usize` wraps to a small number.  Before the fix the
    record was accepted with a huge size and every tensor offset inside it passed the storage bound."""
    p = tmp_path / "ok.bin"
    torch.save({"w": torch.arange(64, dtype=torch.float32)}, p)
    b, eocd, offs = _zip_parts(p)
    tgt = next(o for o in offs if "data/" in _entry_name(b, o))
    nlen, xlen, clen = struct.unpack_from("<HHH", b, tgt + 28)
    # rebuild that central-directory entry with 0xFFFFFFFF sizes and a zip64 extra field carrying (usize, csize)
    head = bytearray(b[tgt:tgt + 46])
    struct.pack_into("<II", head, 20, 0xFFFFFFFF, 0xFFFFFFFF)
    extra = struct.pack("<HHQQ", 1, 16, usize, usize)
    struct.pack_into("<H", head, 30, len(extra))
    new_entry = bytes(head) + bytes(b[tgt + 46:tgt + 46 + nlen]) + extra + bytes(b[tgt + 46 + nlen + xlen:tgt + 46 + nlen + xlen + clen])
    grow = len(new_entry) - (46 + nlen + xlen + clen)
    nb = b[:tgt] + new_entry + b[tgt + 46 + nlen + xlen + clen:]
    e2 = eocd + grow
    if struct.unpack_from("<I", nb, e2 + 12)[0] != 0xFFFFFFFF:
        struct.pack_into("<I", nb, e2 + 12, struct.unpack_from("<I", nb, e2 + 12)[0] + grow)      # central directory size
    if nb[e2 - 20:e2 - 16] == b"PK\x06\x07":                                                      # torch's writer emits zip64 records
        e64 = struct.unpack_from("<Q", nb, e2 - 20 + 8)[0] + grow
        struct.pack_into("<Q", nb, e2 - 20 + 8, e64)
        assert nb[e64:e64 + 4] == b"PK\x06\x06"
        struct.pack_into("<Q", nb, e64 + 40, struct.unpack_from("<Q", nb, e64 + 40)[0] + grow)
    q = tmp_path / "huge.bin"
    open(q, "wb").write(nb)
    _refused(q, "outside the file")
    with pytest.raises(ValueError):
        load_tensors(str(q))


def test_corrupt_zip64_locator_and_directory_fields_are_refused(tmp_path):
    p = tmp_path / "ok.bin"
    torch.save({"w": torch.ones(8)}, p)
    b, eocd, offs = _zip_parts(p)
    # (1) a zip64 locator in front of the EOCD pointing near 2^64 (e64 + 56 wraps)
    loc = struct.pack("<IIQI", 0x07064b50, 0, 2 ** 64 - 8, 1)
    q = tmp_path / "loc.bin"
    open(q, "wb").write(bytes(b[:eocd]) + loc + bytes(b[eocd:]))
    _refused(q, "zip64")
    has64 = b[eocd - 20:eocd - 16] == b"PK\x06\x07"
    e64 = struct.unpack_from("<Q", b, eocd - 20 + 8)[0] if has64 else None
    # (2) central directory offset beyond the file / offset + size wrapping 64 bits
    c = bytearray(b)
    struct.pack_into("<I", c, eocd + 16, 0xFFFFFF00)
    if has64:
        struct.pack_into("<Q", c, e64 + 48, 2 ** 64 - 16)
    q = tmp_path / "cd.bin"
    open(q, "wb").write(c)
    _refused(q, "central directory")
    # (3) local-header offset beyond the file
    c = bytearray(b)
    struct.pack_into("<I", c, offs[0] + 42, 0xFFFFFFF0)
    q = tmp_path / "lho.bin"
    open(q, "wb").write(c)
    _refused(q, "local file header")
    # (4) entry count far larger than the directory can hold
    c = bytearray(b)
    struct.pack_into("<H", c, eocd + 10, 0xFFFF)
    if has64:
        struct.pack_into("<Q", c, e64 + 32, 2 ** 63)
    q = tmp_path / "count.bin"
    open(q, "wb").write(c)
    _refused(q)


def _save_with_patched_pickle(tmp_path, name, patch):
    """torch.save a one-tensor dict, then rewrite data.pkl in place (same length) through `patch(bytes) -> bytes`."""
    p = tmp_path / (name + "_src.bin")
    torch.save({"w": torch.arange(6, dtype=torch.float32).reshape(2, 3)}, p)
    b, eocd, offs = _zip_parts(p)
    tgt = next(o for o in offs if _entry_name(b, o).endswith("data.pkl"))
    lho = struct.unpack_from("<I", b, tgt + 42)[0]
    size = struct.unpack_from("<I", b, tgt + 24)[0]
    data = lho + 30 + struct.unpack_from("<H", b, lho + 26)[0] + struct.unpack_from("<H", b, lho + 28)[0]
    new = patch(bytes(b[data:data + size]))
    assert len(new) == size
    b[data:data + size] = new
    q = tmp_path / (name + ".bin")
    open(q, "wb").write(b)
    return q


def test_shapes_and_strides_that_overflow_int64_are_refused(tmp_path):
    # shape (2, 3) is pickled as K\x02 K\x03 \x86 (BININT1 x2 + TUPLE2); strides (3, 1) follow the same way.  Replace the stride tuple's
    # two BININT1 by ... not enough room for 8-byte ints, so grow the numbers through the storage offset instead: the offset is the
    # BININT1 right after the persistent-id tuple ('K\x00').  A same-length patch cannot carry a 2^62 value, so this case builds its
    # pickle by hand below.
    import io
    import zipfile

    def make(shape, stride, offset, numel=6):
        pk = io.BytesIO()
        P = pickle.Pickler(pk, protocol=2)          # only used for its opcode constants through dumps of plain objects

        def long1(v):
            raw = v.to_bytes(8, "little", signed=True)
            return b"\x8a\x08" + raw
        body = b"\x80\x02}q\x00X\x01\x00\x00\x00wq\x01ctorch._utils\n_rebuild_tensor_v2\nq\x02("
        body += b"(X\x07\x00\x00\x00storageq\x03ctorch\nFloatStorage\nq\x04X\x01\x00\x00\x000q\x05X\x03\x00\x00\x00cpuq\x06" + long1(numel) + b"tq\x07Q"
        body += long1(offset)
        body += b"(" + b"".join(long1(d) for d in shape) + b"t"
        body += b"(" + b"".join(long1(d) for d in stride) + b"t"
        body += b"\x89ccollections\nOrderedDict\nq\x08)Rq\ttq\nRq\x0bs."
        zb = io.BytesIO()
        with zipfile.ZipFile(zb, "w", zipfile.ZIP_STORED) as z:
            z.writestr("archive/data.pkl", body)
            z.writestr("archive/data/0", struct.pack("<6f", *range(6)))
            z.writestr("archive/version", b"3\n")
        return zb.getvalue()
    ok = tmp_path / "hand_ok.bin"
    open(ok, "wb").write(make((2, 3), (3, 1), 0))
    same(load_tensors(str(ok))["w"], torch.arange(6, dtype=torch.float32).reshape(2, 3))          # the hand-built pickle is well-formed
    for tag, (shape, stride, offset) in {
            "span_mul": ((2 ** 62, 2 ** 62), (2 ** 62, 1), 0),           # (shape-1)*stride overflows
            "span_add": ((2 ** 62, 2), (3, 2 ** 62), 0),                 # sum overflows
            "bytes_mul": ((2 ** 61,), (1,), 2 ** 61),                    # (offset + span) * 4 overflows
            "offset_big": ((1,), (1,), 2 ** 63 - 1),                     # offset + span overflows
            "past_end": ((2, 3), (3, 1), 1),                             # plain out-of-record view
            "neg": ((2, 3), (-3, 1), 0)}.items():
        q = tmp_path / f"hand_{tag}.bin"
        open(q, "wb").write(make(shape, stride, offset))
        _refused(q, "tensor 'w'")


def test_self_referential_and_deep_object_trees_are_refused_not_overflowed(tmp_path):
    import io
    import zipfile

    def archive(body):
        zb = io.BytesIO()
        with zipfile.ZipFile(zb, "w", zipfile.ZIP_STORED) as z:
            z.writestr("archive/data.pkl", body)
            z.writestr("archive/version", b"3\n")
        return zb.getvalue()
    # d = {}; d['k'] = d  (EMPTY_DICT, BINPUT 0, 'k', BINGET 0, SETITEM, STOP): legal pickle, infinite tree
    q = tmp_path / "cycle.bin"
    open(q, "wb").write(archive(b"\x80\x02}q\x00X\x01\x00\x00\x00kh\x00s."))
    _refused(q, "deeper than 64")
    # 10 000 nested lists
    q = tmp_path / "deep.bin"
    open(q, "wb").write(archive(b"\x80\x02" + b"]" * 10000 + b"a" * 9999 + b"."))
    _refused(q, "deeper than 64")
    # a DAG bomb: l0 = [0]; l(i) = [l(i-1), l(i-1)] through the memo - 40 levels = 2^40 leaves from 300 bytes, inside the depth limit
    body = b"\x80\x02]q\x00K\x00a"                                # (earlier levels stay on the stack below the result: legal)
    for i in range(1, 41):
        body += b"]q" + bytes([i]) + b"h" + bytes([i - 1]) + b"ah" + bytes([i - 1]) + b"a"
    body += b"."
    q = tmp_path / "dag.bin"
    open(q, "wb").write(archive(body))
    import time as _t
    t0 = _t.perf_counter()
    _refused(q, "more than 4000000 nodes")
    assert _t.perf_counter() - t0 < 30
    # safetensors metadata with 100 000 nested arrays
    s = tmp_path / "deep.safetensors"
    hdr = (b'{"__metadata__":' + b"[" * 100000 + b"]" * 100000 + b',"w":{"dtype":"F32","shape":[1],"data_offsets":[0,4]}}')
    open(s, "wb").write(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 4)
    _refused(s, "nested deeper")


def test_safetensors_shape_products_that_overflow_are_refused(tmp_path):
    for shape in ([2 ** 32, 2 ** 32], [2 ** 62, 4], [99999999999999999999, 1]):
        s = tmp_path / "ovf.safetensors"
        hdr = json.dumps({"w": {"dtype": "F32", "shape": shape, "data_offsets": [0, 0]}}).encode()
        open(s, "wb").write(struct.pack("<Q", len(hdr)) + hdr)
        _refused(s)
