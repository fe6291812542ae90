"""Checkpoint file access for the loaders: the files the reference reads with torch.load / safetensors
(builder.py:138-185: sharded `pytorch_model-0000x-of-0000y.bin`, `adapter_model.bin`, `non_lora_trainables.bin`, encoder checkpoints).

The files are opened by the native reader of libmc_hip.so (csrc/ckpt_reader.cpp, C ABI `mc_ckpt_*`): it maps the file, walks the zip
directory and interprets the pickle (or the safetensors header) itself - Python never unpickles anything, nothing in a checkpoint can
execute, and no tensor is copied on the host: each returned torch tensor is a zero-copy view of the mapped file (copy-on-write pages),
touched once, by its host-to-device copy.  The 13.5 GB base model therefore never exists twice in host memory."""
from __future__ import annotations

import ctypes as C
from typing import Dict

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

    def tensors(self) -> Dict[str, torch.Tensor]:
        L = _lib.lib()
        n = C.c_int(0)
        _lib.check(L.mc_ckpt_count(self._h, C.byref(n)), "mc_ckpt_count")
        out: Dict[str, torch.Tensor] = {}
        name, dt, nd = C.c_char_p(), C.c_int(0), C.c_int(0)
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
            out[name.value.decode()] = t
        return out


def load_tensors(path: str) -> Dict[str, torch.Tensor]:
    """name -> CPU tensor (zero-copy view of the mapped file) for a torch zip checkpoint or a .safetensors file."""
    return MappedCheckpoint(path).tensors()
