"""the reference's callers import `modelcompose.*`; every one of those import lines must resolve against this repo with
zero edits and give objects with the reference's signatures.  The lines below are the import statements of
modelcompose/eval/model_multimodal_qa_loader.py:11-18, modelcompose/train/train_multimodal.py:33-40, modelcompose/model/builder.py:23
and demo_app.py:13-20 of the reference (statements, not code: they are the interface under test)."""
import inspect

import pytest

REFERENCE_IMPORT_LINES = [
    "from modelcompose.constants import IMAGE_TOKEN_INDEX, DEFAULT_IMAGE_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN",
    "from modelcompose.constants import IGNORE_INDEX, DEFAULT_IMAGE_PATCH_TOKEN",
    "from modelcompose import conversation as conversation_lib",
    "from modelcompose.conversation import conv_templates, SeparatorStyle",
    "from modelcompose.model.builder import load_pretrained_model",
    "from modelcompose.utils import disable_torch_init",
    "from modelcompose.mm_utils import tokenizer_image_token, process_images, get_model_name_from_path",
    "from modelcompose.mm_utils import tokenizer_modal_token",
    "from modelcompose.data.multimodal_dataset import MultimodalDataset, DataCollatorForSupervisedDataset",
    "from modelcompose.data import make_multimodal_data_module",
    "from modelcompose.model import MultimodalLlamaForCausalLM",
    "from modelcompose.model import *",
    "from modelcompose import LlavaLlamaForCausalLM",
    "from modelcompose.model.language_model.multimodal_llama import MultimodalLlamaForCausalLM, MultimodalConfig",
    "from modelcompose.train.llava_trainer import LengthGroupedSampler",
    "import modelcompose.eval.model_multimodal_qa_loader",
    # serve/model_worker.py:20-22 (the model-facing imports of the worker) and the worker's own module path
    "from modelcompose.mm_utils import process_images, load_image_from_base64, tokenizer_image_token, KeywordsStoppingCriteria",
    "from modelcompose.serve.model_worker import ModelWorker",
]


@pytest.mark.parametrize("line", REFERENCE_IMPORT_LINES)
def test_reference_import_line_resolves(line):
    exec(line, {})


def _params(fn):
    return list(inspect.signature(fn).parameters)


def test_signatures_match_the_reference_surface():
    from modelcompose.model import LlavaLlamaForCausalLM, MultimodalConfig, MultimodalLlamaForCausalLM
    from modelcompose.model.builder import load_pretrained_model
    # builder.py:27
    sig = inspect.signature(load_pretrained_model)
    assert [k for k, v in sig.parameters.items() if v.kind == v.POSITIONAL_OR_KEYWORD] == ["model_path", "model_base", "model_name", "load_8bit", "load_4bit",
                                                                                           "device_map", "device"]
    assert [k for k, v in sig.parameters.items() if v.kind == v.KEYWORD_ONLY] == ["torch_dtype"]        # extension: the reference hard-codes float16 (:41)
    assert sig.parameters["device_map"].default == "auto" and sig.parameters["device"].default == "cuda"
    assert LlavaLlamaForCausalLM is MultimodalLlamaForCausalLM and MultimodalLlamaForCausalLM.config_class is MultimodalConfig
    M = MultimodalLlamaForCausalLM
    # multimodal_llama.py:676-688
    fwd = inspect.signature(M.forward).parameters
    positional = [k for k, v in fwd.items() if v.kind == v.POSITIONAL_OR_KEYWORD]
    assert positional[1:] == ["input_ids", "attention_mask", "past_key_values", "inputs_embeds", "labels", "use_cache",
                              "output_attentions", "output_hidden_states", "modal_inputs", "return_dict"]
    # extensions are keyword-only: a positional call written against the reference means the same thing here
    assert all(v.kind == v.KEYWORD_ONLY for k, v in fwd.items() if k not in positional)
    # multimodal_arch.py:197, :287
    assert _params(M.encode_modal_inputs)[1:] == ["inputs", "prefix_tokens", "suffix_tokens"]
    assert _params(M.prepare_inputs_labels_for_multimodal)[1:] == ["input_ids", "attention_mask", "past_key_values", "labels", "modal_inputs",
                                                                   "prefix_tokens", "suffix_tokens"]
    # model_multimodal_qa_loader.py:94-102: the keywords the eval loop passes to generate()
    gen = _params(M.generate)
    for kw in ("input_ids", "modal_inputs", "do_sample", "temperature", "top_p", "num_beams", "max_new_tokens", "use_cache"):
        assert kw in gen, kw
    for name in ("get_model", "get_modal_encoders", "get_modal_projectors", "get_modal_processors", "prepare_inputs_for_generation"):
        assert callable(getattr(M, name)), name
    # train_multimodal.py:307-325, :396-399, :436-465: the train() caller's entry points
    assert isinstance(inspect.getattr_static(M, "from_pretrained"), classmethod)
    fp = inspect.signature(M.from_pretrained)
    assert list(fp.parameters)[0] == "pretrained_model_name_or_path" and "cache_dir" in fp.parameters
    assert any(v.kind == v.VAR_KEYWORD for v in fp.parameters.values())              # lora_* / local_* / mm_*_encoder keywords update the config
    from modelcompose_amd.model.multimodal_llama import MultimodalLlamaModel
    assert _params(MultimodalLlamaModel.initialize_multimodal_modules)[1:] == ["model_args", "fsdp"]
    assert inspect.signature(MultimodalLlamaModel.initialize_multimodal_modules).parameters["fsdp"].default is None
    for cls in (M, MultimodalLlamaModel):
        for name in ("named_parameters", "parameters", "requires_grad_"):
            assert callable(getattr(cls, name)), (cls, name)


def test_requires_grad_selection_follows_the_reference_loop():
    """The selection loop of train_multimodal.py:436-465 over named_parameters() / get_modal_projectors().parameters(), on the host-side
    parameter views (no GPU needed: the views only carry names and flags)."""
    import weakref
    from modelcompose_amd.model.multimodal_llama import MultimodalLlamaForCausalLM as M, MultimodalLlamaModel

    class Owner:                                        # the state the views read: reference-grammar tensors and the selection
        named_parameters, parameters, requires_grad_, trainable_names = M.named_parameters, M.parameters, M.requires_grad_, M.trainable_names
        _param_tensor = M._param_tensor

        def __init__(self):
            self._requires_grad = {}
            self._raw = {k: None for k in ("model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.q_proj.lora_A.default.weight",
                                           "model.layers.0.self_attn.q_proj.lora_B.vision.weight", "model.modal_projectors.vision.0.weight", "lm_head.weight")}
            self.prefix_tokens, self.suffix_tokens = {"default": None, "vision": None}, None
    o = Owner()
    inner = MultimodalLlamaModel(config=None)
    inner._owner = weakref.ref(o)
    inner.modal_projectors = {"vision": object()}
    o.requires_grad_(False)
    for n, p in o.named_parameters():
        if "prefix_tokens" in n:
            p.requires_grad = True
    for p in inner.get_modal_projectors().parameters():
        p.requires_grad = True
    for n, p in inner.named_parameters():
        if "lora" in n:
            p.requires_grad = True
    assert sorted(o.trainable_names()) == ["model.layers.0.self_attn.q_proj.lora_A.default.weight", "model.layers.0.self_attn.q_proj.lora_B.vision.weight",
                                           "model.modal_projectors.vision.0.weight", "prefix_tokens.default", "prefix_tokens.vision"]
    inner.requires_grad_(False)                         # freeze_backbone (:327-328): everything under model.
    assert sorted(o.trainable_names()) == ["prefix_tokens.default", "prefix_tokens.vision"]


def test_conversation_module_is_shared_state():
    """The eval loop assigns conversation_lib.default_conversation and the preprocessing reads it back: one module object."""
    from modelcompose import conversation as ref_path
    from modelcompose_amd import conversation as impl
    assert ref_path is impl
    old = impl.default_conversation
    try:
        ref_path.default_conversation = ref_path.conv_templates["llama_2"]
        assert impl.default_conversation.version == "llama_v2"
    finally:
        impl.default_conversation = old


def test_sentinel_constants_and_cli_entry_points():
    from modelcompose.constants import MODAL_TOKEN_INDEXES, MODAL_TOKEN_MAPPING
    assert MODAL_TOKEN_INDEXES == {"vision": -200, "relrep": -201, "text": -202, "audio": -203, "video": -204, "point": -205}
    assert MODAL_TOKEN_MAPPING["<image>"] == -200 and MODAL_TOKEN_MAPPING["<point>"] == -205
    from modelcompose.eval.model_multimodal_qa_loader import parse_args
    a = parse_args(["--model-path", "x/multimodal-y", "--question-file", "q.json", "--answers-file", "a.jsonl", "--conv-mode", "v1",
                    "--num-chunks", "8", "--chunk-idx", "3", "--temperature", "0", "--model-base", "None"])       # MCUB-4.sh:42-58
    assert a.num_chunks == 8 and a.chunk_idx == 3 and a.model_base is None and a.temperature == 0
