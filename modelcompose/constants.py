"""modelcompose/constants.py of the reference: sentinel ids / token strings."""
from modelcompose_amd.constants import *  # noqa: F401,F403
from modelcompose_amd.constants import (DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_IMAGE_TOKEN,  # noqa: F401
                                        IGNORE_INDEX, IMAGE_TOKEN_INDEX, MODAL_TOKEN_INDEXES, MODAL_TOKEN_MAPPING, MODAL_TOKENS)
