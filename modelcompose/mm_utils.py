"""modelcompose/mm_utils.py of the reference: sentinel tokenisation, image batch preparation, stopping criteria."""
from modelcompose_amd.mm_utils import (KeywordsStoppingCriteria, expand2square, get_model_name_from_path, load_image_from_base64,  # noqa: F401
                                       process_images, split_string_by_list, tokenizer_image_token, tokenizer_modal_token)
