"""modelcompose_amd — MI355X (gfx950) native forward / generation path for ModelCompose's
composed multimodal LLM.  The compute lives in ``libmc_hip.so`` (hand-written HIP, C ABI in
``include/mc_hip.h``); this package is the host-side mirror of the reference's Python interface
(``modelcompose.model.builder.load_pretrained_model`` / ``MultimodalLlamaForCausalLM``).
There is no CPU fallback: without the HIP library the ops raise."""
__version__ = "0.1.0"


def set_storage_dtype(name):
    """"bf16" (default: BASELINE.json's dtype, the headline) or "fp16" (the reference's own inference dtype, libmc_hip_f16.so: the parity
    instrument).  One storage dtype per process; call before building models (or set MC_STORAGE_DTYPE=fp16 in the environment)."""
    from . import _lib
    return _lib.set_storage_dtype(name)
