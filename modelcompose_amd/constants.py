"""Sentinel ids and token strings of the reference (modelcompose/constants.py:7-31) — the input contract of
the splice."""
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"
MODAL_TOKENS = {"vision": "<image>", "relrep": "<relrep>", "text": "<text>", "audio": "<audio>", "video": "<video>",
                "point": "<point>"}
MODAL_TOKEN_INDEXES = {"vision": -200, "relrep": -201, "text": -202, "audio": -203, "video": -204, "point": -205}
MODAL_TOKEN_MAPPING = {MODAL_TOKENS[k]: MODAL_TOKEN_INDEXES[k] for k in MODAL_TOKENS}
