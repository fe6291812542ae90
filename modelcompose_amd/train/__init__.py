"""Stage-2 finetune step on the HIP path (BASELINE config 5)."""
from .step import MultimodalTrainStep  # noqa: F401
