"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/mc_hip.h declares (no compute calls without a GPU), and argument validation reports through mc_last_error."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from modelcompose_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib


def test_header_symbols_are_exported_and_bound(lib):
    L = lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), f"libmc_hip.so lacks {s}"
    bound = set(lib.exported_symbols())
    missing = [s for s in syms if s not in bound]
    assert not missing, f"ctypes signatures missing for {missing}"
    assert L.mc_abi_version() == lib.ABI_VERSION


def test_the_fp16_instantiation_exports_the_same_abi(lib):
    """libmc_hip_f16.so = the same sources on IEEE-half storage (csrc/common.h; the reference's own dtype, model/builder.py:41): same entry
    points, same ABI version, and it says which storage element it was built on."""
    import subprocess
    import sys
    assert os.path.exists(lib.LIB_PATHS["fp16"]), "make -C modelcompose_amd/csrc builds both libraries"
    L = lib.lib()
    assert lib.storage_name() == "bf16" and L.mc_storage_dtype() == 1
    # one storage dtype per process: the other library is checked in a child interpreter
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from modelcompose_amd import _lib\n"
            "import torch\n"
            "assert _lib.storage_name() == 'fp16' and _lib.storage_dtype() == torch.float16\n"
            "L = _lib.lib()\n"
            "assert L.mc_storage_dtype() == 2 and L.mc_abi_version() == _lib.ABI_VERSION\n"
            "from modelcompose_amd import ops\n"
            "from modelcompose_amd.model import multimodal_llama as mm\n"
            "assert ops.BF16 == torch.float16 and mm.BF16 == torch.float16\n"
            "import modelcompose_amd\n"
            "modelcompose_amd.set_storage_dtype('fp16')\n"                    # the loaded one: a no-op
            "try:\n"
            "    modelcompose_amd.set_storage_dtype('bf16')\n"                # no switch under tensors / packed weights of the other element
            "    raise SystemExit('a mid-process switch was accepted')\n"
            "except _lib.MCError as e:\n"
            "    assert 'already loaded' in str(e)\n"
            "assert ops.BF16 == torch.float16 and _lib.lib().mc_storage_dtype() == 2\n"
            "print('ok')\n") % ROOT
    env = dict(os.environ, MC_STORAGE_DTYPE="fp16")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
    # before the first use the choice can still be made from code
    code2 = ("import sys; sys.path.insert(0, %r)\n"
             "import torch, modelcompose_amd\n"
             "from modelcompose_amd import _lib\n"
             "modelcompose_amd.set_storage_dtype('fp16')\n"
             "from modelcompose_amd import ops\n"
             "assert ops.BF16 == torch.float16 and _lib.lib().mc_storage_dtype() == 2\n"
             "print('ok')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k != "MC_STORAGE_DTYPE"})
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_argument_errors_do_not_touch_the_gpu(lib):
    L = lib.lib()
    # null pointers / bad shapes are rejected before any HIP call
    rc = L.mc_gemm_bf16(None, 0, None, None, None, 0, None, 0, 1, 1, 1, 0, 0, 1.0, 1.0, None)
    assert rc == 1 and b"null pointer" in L.mc_last_error()
    with pytest.raises(ValueError):
        lib.check(rc, "mc_gemm_bf16")
    n = C.c_int64(0)
    assert L.mc_packed_weight_elems(4096, 588, C.byref(n)) == 0 and n.value == 4096 * 640
    cfg = lib.LlmConfigC(100, 64, 1, 3, 3, 33, 97, 1, 128, 1e-5)
    h = C.c_void_p(0)
    assert L.mc_llm_create(C.byref(cfg), C.byref(h)) == 1 and b"unsupported geometry" in L.mc_last_error()
    cfg = lib.LlmConfigC(128, 192, 2, 2, 2, 64, 128, 2, 256, 1e-5)
    assert L.mc_llm_create(C.byref(cfg), C.byref(h)) == 0 and h.value
    assert L.mc_llm_prefill(h, None, 1, 1, None, None, None, None, None, None, None, None, 1, 1, None, None, 1, None, None,
                            None, None, None) == 1
    assert b"mc_llm_set_weights has not been called" in L.mc_last_error()
    v = C.c_int(-1)
    assert L.mc_llm_get_option(h, b"graph_active", C.byref(v)) == 0 and v.value == 0
    assert L.mc_llm_get_option(h, b"no_such_option", C.byref(v)) == 1
    assert L.mc_llm_destroy(h) == 0


def test_runtime_bounds_are_enforced_in_the_c_abi(lib):
    """the group table of mc_llm_prefill is a fixed 64-entry array and mc_llm_decode appends one key per step: both bounds are
    checked by the library itself (not only by the Python wrapper), before any launch."""
    L = lib.lib()
    cfg = lib.LlmConfigC(128, 192, 1, 2, 2, 64, 128, 1, 256, 1e-5)
    h = C.c_void_p(0)
    assert L.mc_llm_create(C.byref(cfg), C.byref(h)) == 0
    one = C.c_void_p(16)                                          # non-null dummy device pointers: validation precedes any dereference
    ptrs = (C.c_void_p * 4)(16, 16, 16, 16)
    assert L.mc_llm_set_weights(h, ptrs, one, one, one, one, one) == 0
    gs = (C.c_int32 * 66)(*range(66))
    ga = (C.c_int32 * 65)(*([0] * 65))
    rc = L.mc_llm_prefill(h, one, 65, 65, gs, ga, one, one, one, one, one, one, 1, 65, one, one, 128, one, None, None, one, None)
    assert rc == 1 and b"at most 64 groups" in L.mc_last_error()
    rc = L.mc_llm_decode(h, 1, 8, one, one, 8, one, one, one, 64, 60, one, None, None)
    assert rc == 1 and b"KV cache overflow" in L.mc_last_error()
    assert L.mc_llm_destroy(h) == 0


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from modelcompose_amd import _lib, ops
    from modelcompose_amd.model import MultimodalConfig, MultimodalLlamaForCausalLM
    with pytest.raises(_lib.MCError):
        MultimodalLlamaForCausalLM(MultimodalConfig())
    with pytest.raises(ValueError):
        ops.rmsnorm(torch.zeros(2, 64, dtype=torch.bfloat16), torch.ones(64, dtype=torch.bfloat16), 1e-5)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under modelcompose_amd/ may import it except smoke.py's checker."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "modelcompose_amd")):
        for f in fs:
            if f.endswith(".py") and f != "smoke.py":
                if re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(dp, f)).read(), flags=re.M):
                    bad.append(f)
    assert not bad, bad


def test_hot_kernels_do_not_spill():
    """The dominant kernels sit AT the register limit of two waves per SIMD (gemm_tile256_kernel: 256 VGPRs): a source change that pushes
    hipcc over it does not fail the build, it spills into scratch inside the main loop.  Read the code objects' metadata out of the built
    library (tools/kernel_resources.py) and refuse spills in the shipped instantiations."""
    import importlib.util
    import os
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (os.path.exists("/opt/rocm/lib/llvm/bin/clang-offload-bundler") and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf")
            and shutil.which("true")):
        pytest.skip("ROCm LLVM tools not available")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(root, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ks = {k["name"]: k for k in mod.kernel_resources()}
    assert len(ks) > 50, "no kernels found in libmc_hip.so"
    hot = ["_Z19gemm_tile256_kernelILi0ELi4ELi233EEvPKDF16bl8G2Groupsii8Epilogueiii", "_Z19gemm_tile256_kernelILi0ELi3ELi233EEvPKDF16bl8G2Groupsii8Epilogueiii",
           "_Z19attn_prefill_kernelILi128ELb0ELi4ELi2ELb0ELi1ELb0ELb0EEv10AttnParams", "_Z18attn_decode_kernelILi128ELi4EEv12DecodeParams",
           "_Z21attn_prefill32_kernelILb1EEv10AttnParams"]
    for name in hot:
        assert name in ks, f"{name} is not in the library (renamed instantiation? update this list)"
        k = ks[name]
        assert k["vgpr_spill_count"] == 0 and k["sgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
        assert k["vgpr_count"] + k["agpr_count"] <= 256, k          # two waves per SIMD
    k = ks["_Z20compose_multi_kernelILi4ELb1ELb1EEv18ComposeMultiParams"]          # (a few scalar spills into vector lanes are harmless; scratch is not)
    assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0 and k["vgpr_count"] + k["agpr_count"] <= 256, k
    n_strip = 0
    for name, k in ks.items():                                     # the M <= 64 GEMM and the decode attention: every instantiation
        if "gemm_strip_kernel" in name or "attn_decode_kernel" in name:
            n_strip += "gemm_strip_kernel" in name
            assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
    assert n_strip >= 30, n_strip
