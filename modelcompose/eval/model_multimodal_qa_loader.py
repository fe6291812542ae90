"""modelcompose/eval/model_multimodal_qa_loader.py of the reference: `python -m modelcompose.eval.model_multimodal_qa_loader ...`
(scripts/model_composition/test/MCUB-4.sh:42-58) runs the HIP eval loop with the same flags."""
from modelcompose_amd.eval.model_multimodal_qa_loader import (ChunkedMultimodalDataset, create_data_loader, eval_model, get_chunk,  # noqa: F401
                                                              parse_args, split_list)

if __name__ == "__main__":
    eval_model(parse_args())
