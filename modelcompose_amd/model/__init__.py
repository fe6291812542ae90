from .config import MultimodalConfig, infer_modals  # noqa: F401
from .multimodal_llama import LlavaLlamaForCausalLM, MultimodalLlamaForCausalLM  # noqa: F401
