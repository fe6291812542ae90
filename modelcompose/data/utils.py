"""modelcompose/data/utils.py of the reference: conversation preprocessing (label masking per template)."""
from modelcompose_amd.data import preprocess, preprocess_llama_2, preprocess_mpt, preprocess_plain, preprocess_v1  # noqa: F401
