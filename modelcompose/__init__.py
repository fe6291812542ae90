"""Drop-in import root: the reference's callers (`modelcompose/eval/model_multimodal_qa_loader.py:11-18`,
`modelcompose/train/train_multimodal.py:33-40`, `demo_app.py:13-20`) import `modelcompose.*`; every module under this
package re-exports the MI355X implementation in `modelcompose_amd` under the reference's module path, so those call
sites resolve with zero edits.  `modelcompose/__init__.py:1` of the reference exports LlavaLlamaForCausalLM."""
from .model import LlavaLlamaForCausalLM  # noqa: F401
