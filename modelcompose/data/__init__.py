"""modelcompose/data/__init__.py of the reference."""
from modelcompose_amd.data import DataCollatorForSupervisedDataset, MultimodalDataset, make_multimodal_data_module  # noqa: F401
