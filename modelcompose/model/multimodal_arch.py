"""modelcompose/model/multimodal_arch.py of the reference: encode_modal_inputs / prepare_inputs_labels_for_multimodal live on the
model class of the HIP path (they are methods of MultimodalMetaForCausalLM in the reference, :169-459)."""
from modelcompose_amd.model.multimodal_llama import MultimodalLlamaForCausalLM as MultimodalMetaForCausalLM  # noqa: F401
from modelcompose_amd.model.multimodal_llama import MultimodalLlamaModel as MultimodalMetaModel  # noqa: F401
