"""modelcompose_amd — MI355X (gfx950) native forward / generation path for ModelCompose's
composed multimodal LLM.  The compute lives in ``libmc_hip.so`` (hand-written HIP, C ABI in
``include/mc_hip.h``); this package is the host-side mirror of the reference's Python interface
(``modelcompose.model.builder.load_pretrained_model`` / ``MultimodalLlamaForCausalLM``).
There is no CPU fallback: without the HIP library the ops raise."""
__version__ = "0.1.0"
