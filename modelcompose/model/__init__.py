"""modelcompose/model/__init__.py:1-4 of the reference."""
from modelcompose_amd.model import LlavaLlamaForCausalLM, MultimodalConfig, MultimodalLlamaForCausalLM  # noqa: F401
