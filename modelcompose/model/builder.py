"""modelcompose/model/builder.py of the reference: load_pretrained_model (multimodal branch, :138-185)."""
from modelcompose_amd.model.builder import build_from_state_dict, load_pretrained_model  # noqa: F401
