"""modelcompose/model/language_model/multimodal_llama.py of the reference."""
from modelcompose_amd.model.config import MultimodalConfig  # noqa: F401
from modelcompose_amd.model.multimodal_llama import (CausalLMOutputWithPast, MultimodalLlamaForCausalLM,  # noqa: F401
                                                     MultimodalLlamaModel)
