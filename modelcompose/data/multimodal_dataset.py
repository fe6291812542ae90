"""modelcompose/data/multimodal_dataset.py of the reference."""
from modelcompose_amd.data import DataCollatorForSupervisedDataset, MultimodalDataset  # noqa: F401
