"""modelcompose/train/llava_trainer.py of the reference: the length-grouped samplers (:60-134); the HF Trainer subclass itself is
out of scope - one optimisation step is modelcompose_amd.train.MultimodalTrainStep."""
from modelcompose_amd.train.sampler import (LengthGroupedSampler, get_length_grouped_indices, get_modality_length_grouped_indices,  # noqa: F401
                                            split_to_even_chunks)
