"""Checkpoint file access for the loaders: the files the reference reads with torch.load / safetensors
(builder.py:138-185: sharded `pytorch_model-0000x-of-0000y.bin`, `adapter_model.bin`, `non_lora_trainables.bin`, encoder checkpoints).

The files are opened by the native reader of libmc_hip.so (csrc/ckpt_reader.cpp, C ABI `mc_ckpt_*`): it maps the file, walks the zip
directory and interprets the pickle (or the safetensors header) itself - Python never unpickles anything, nothing in a checkpoint can
execute, and no tensor is copied on the host: each returned torch tensor is a zero-copy view of the mapped file (copy-on-write pages),
touched once, by its host-to-device copy.  The 13.5 GB base model therefore never exists twice in host memory."""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict

import torch

from . import _lib

_DTYPES = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16, 3: torch.float64, 4: torch.int64, 5: torch.int32, 6: torch.int16,
           7: torch.int8, 8: torch.uint8, 9: torch.bool}


class MappedCheckpoint:
    """One open checkpoint file; keeps the mapping alive for as long as any tensor view of it is referenced."""

    def __init__(self, path: str):
        self.path = path
        self._h = C.c_void_p(0)
        _lib.check(_lib.lib().mc_ckpt_open(str(path).encode(), C.byref(self._h)), "mc_ckpt_open")

    def __del__(self):
        try:
            if self._h:
                _lib.lib().mc_ckpt_close(self._h)
                self._h = C.c_void_p(0)
        except Exception:
            pass

    def tensors(self, by_path: bool = False) -> Dict[str, torch.Tensor]:
        """'.'-joined name -> tensor (by_path: the structured path of mc_ckpt_entry_path instead of the name)."""
        L = _lib.lib()
        n = C.c_int(0)
        _lib.check(L.mc_ckpt_count(self._h, C.byref(n)), "mc_ckpt_count")
        out: Dict[str, torch.Tensor] = {}
        name, dt, nd = C.c_char_p(), C.c_int(0), C.c_int(0)
        path = C.c_char_p()
        shp, strd = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)()
        data, sbytes = C.c_void_p(0), C.c_int64(0)
        for i in range(n.value):
            _lib.check(L.mc_ckpt_entry(self._h, i, C.byref(name), C.byref(dt), C.byref(nd), C.byref(shp), C.byref(strd), C.byref(data),
                                       C.byref(sbytes)), "mc_ckpt_entry")
            dtype = _DTYPES[dt.value]
            shape = [shp[k] for k in range(nd.value)]
            stride = [strd[k] for k in range(nd.value)]
            es = torch.empty((), dtype=dtype).element_size()
            numel_storage = sbytes.value // es
            if numel_storage == 0 or 0 in shape:
                t = torch.empty(shape, dtype=dtype)
            else:
                buf = (C.c_char * (numel_storage * es)).from_address(data.value)
                buf._mc_owner = self                                 # tensor -> buffer object -> this mapping: the views keep the file mapped
                flat = torch.frombuffer(buf, dtype=dtype, count=numel_storage)
                t = flat.as_strided(shape, stride)
            if by_path:
                _lib.check(L.mc_ckpt_entry_path(self._h, i, C.byref(path)), "mc_ckpt_entry_path")
                out[path.value.decode()] = t
            else:
                out[name.value.decode()] = t
        return out

    def scalars(self, by_path: bool = False) -> Dict[str, Any]:
        """Non-tensor leaves (None / bool / int / float / str) of the file's object tree, keyed like tensors()."""
        L = _lib.lib()
        n = C.c_int(0)
        _lib.check(L.mc_ckpt_scalar_count(self._h, C.byref(n)), "mc_ckpt_scalar_count")
        out: Dict[str, Any] = {}
        name, path, kind, iv, fv, sv, sl = C.c_char_p(), C.c_char_p(), C.c_int(0), C.c_int64(0), C.c_double(0), C.c_void_p(0), C.c_int64(0)
        for i in range(n.value):
            _lib.check(L.mc_ckpt_scalar(self._h, i, C.byref(name), C.byref(path), C.byref(kind), C.byref(iv), C.byref(fv), C.byref(sv),
                                        C.byref(sl)), "mc_ckpt_scalar")
            k = kind.value
            if k == 0:
                v = None
            elif k == 1:
                v = bool(iv.value)
            elif k == 2:
                v = int(iv.value)
            elif k == 3:
                v = float(fv.value)
            elif k in (5, 6, 7):                                   # MC_CKPT_EMPTY_LIST / _TUPLE / _DICT: the container keeps its place
                v = [] if k == 5 else () if k == 6 else {}
            else:
                v = C.string_at(sv.value, sl.value).decode("utf-8", "surrogateescape") if sl.value else ""
            out[(path if by_path else name).value.decode("utf-8", "surrogateescape")] = v
        return out

    def nested(self):
        """The file's object tree rebuilt from the reader's paths: dicts (string / integer keys come back as strings), lists (tuples
        come back as lists; EMPTY lists / tuples / dicts keep their own type), tensors and scalar leaves.  Opaque objects are not represented."""
        SEP, IDX = "\x1f", "\x1e"
        leaves = list(self.tensors(by_path=True).items()) + list(self.scalars(by_path=True).items())
        if len(leaves) == 1 and leaves[0][0] == "":
            return leaves[0][1]                                   # the file holds a bare tensor / scalar
        root: dict = {}
        for path, leaf in leaves:
            parts = path.split(SEP)
            node = root
            for k, part in enumerate(parts):
                last = k == len(parts) - 1
                if last:
                    node[part] = leaf
                else:
                    node = node.setdefault(part, {})

        def fix(node):
            if not isinstance(node, dict):
                return node
            if node and all(k.startswith(IDX) for k in node):
                items = sorted(((int(k[1:]), fix(v)) for k, v in node.items()))
                return [v for _, v in items]
            return {k: fix(v) for k, v in node.items()}
        return fix(root)


def _is_legacy_torch_file(path: str) -> bool:
    """torch.save before 1.6 (or _use_new_zipfile_serialization=False): a pickled magic number, not a zip archive."""
    try:
        with open(path, "rb") as f:
            head = f.read(14)
    except OSError:
        return False
    # detected POSITIVELY: protocol-2 pickle of torch's legacy magic number 0x1950a86a20f9469cfc6c (serialization.py MAGIC_NUMBER);
    # everything else - zip archives, safetensors with or without the suffix - goes to the native reader, which sniffs by content
    return head == b"\x80\x02\x8a\x0a\x6c\xfc\x9c\x46\xf9\x20\x6a\xa8\x50\x19"


def _legacy_load(path: str):
    # The native reader handles zip archives and safetensors only.  Pre-zip files go through torch's OWN restricted unpickler
    # (weights_only=True refuses every global outside its tensor allow-list), never through plain pickle.
    return torch.load(path, map_location="cpu", weights_only=True)


def _flatten(obj, prefix="", out=None):
    out = {} if out is None else out
    if isinstance(obj, dict):
        for k, v in obj.items():
            _flatten(v, f"{prefix}.{k}" if prefix else str(k), out)
    elif isinstance(obj, (list, tuple)):
        for k, v in enumerate(obj):
            _flatten(v, f"{prefix}.{k}" if prefix else str(k), out)
    elif torch.is_tensor(obj):
        out[prefix] = obj
    return out


def load_tensors(path: str) -> Dict[str, torch.Tensor]:
    """name -> CPU tensor (zero-copy view of the mapped file) for a torch zip checkpoint or a .safetensors file.  Nested containers
    are flattened with '.'-joined names; non-tensor values are dropped (load_nested keeps them)."""
    if _is_legacy_torch_file(path):
        return _flatten(_legacy_load(path))
    return MappedCheckpoint(path).tensors()


def load_nested(path: str):
    """The checkpoint's object tree with its non-tensor leaves (config dicts stored next to the weights, e.g. BEATs' {'cfg', 'model'})."""
    if _is_legacy_torch_file(path):
        return _legacy_load(path)
    return MappedCheckpoint(path).nested()
