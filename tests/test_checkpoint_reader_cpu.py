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
from modelcompose_amd.checkpoint_io import MappedCheckpoint, load_tensors


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
    assert set(got) == {"model.a", "model.b.c"}
    same(got["model.a"], torch.ones(2))
    same(got["model.b.c"], torch.zeros(3, dtype=torch.int64))


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
    # legacy (non-zip) torch format
    q = tmp_path / "legacy.bin"
    torch.save({"w": torch.ones(2)}, q, _use_new_zipfile_serialization=False)
    with pytest.raises(ValueError, match="neither a zip"):
        load_tensors(str(q))
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
