"""modelcompose/conversation.py of the reference: prompt templates.  Callers assign `conversation_lib.default_conversation = ...`
(eval/model_multimodal_qa_loader.py, train_multimodal.py) and the preprocessing reads it back, so this path must be the SAME module
object as the implementation, not a copy of its names."""
import sys

from modelcompose_amd import conversation as _impl

sys.modules[__name__] = _impl
