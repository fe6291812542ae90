"""Caller side of the path: conversation -> (input_ids with sentinels, labels) and the batch collator that produces the
`modal_inputs` layouts the model consumes.  Host logic (string / integer / index work), bit-exact with the reference for the same
tokenizer.  Mirrors

  modelcompose/data/utils.py:17-26    _mask_targets            :28-46  _add_speaker_and_signal     :48-72  _tokenize_fn
                        :74-153, :156-235  preprocess_llama_2 / preprocess_v1 (same arithmetic, different separator)
                        :304-323  preprocess_plain        :326-370  preprocess (dispatch + the v0 '### ' format)
  modelcompose/data/multimodal_dataset.py:51-131  MultimodalDataset        :139-214  DataCollatorForSupervisedDataset

The media decoders the reference's dataset calls (PIL for images, decord / torchaudio inside the processors for video / audio files)
are CPU codec work outside the path; images are opened with PIL as in the reference, the other modalities are handed to the modal
processors as the dataset stores them (file names or already-decoded arrays)."""
from __future__ import annotations

import copy
import json
import random
from collections import defaultdict
from dataclasses import dataclass
from typing import Dict, Sequence

import torch

from . import conversation as conversation_lib
from .constants import IGNORE_INDEX
from .mm_utils import tokenizer_image_token, tokenizer_modal_token


def _mask_targets(target, tokenized_lens, speakers):
    cur = tokenized_lens[0]
    target[:cur] = IGNORE_INDEX
    for n, speaker in zip(tokenized_lens[1:], speakers):
        if speaker == "human":
            target[cur + 2:cur + n] = IGNORE_INDEX
        cur += n


def _add_speaker_and_signal(header, source, get_conversation=True):
    """'### <role>: <text>\\n' per sentence (rewrites sentence['value'] in place, as the reference does), closing '### '."""
    roles = conversation_lib.default_conversation.roles
    text = header
    for sentence in source:
        who = sentence["from"].lower()
        name = roles[0] if who == "human" else roles[1] if who == "gpt" else "unknown"
        sentence["value"] = "### " + name + ": " + sentence["value"] + "\n"
        if get_conversation:
            text += sentence["value"]
    return text + "### "


def _tokenize_fn(strings, tokenizer):
    toks = [tokenizer(t, return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length, truncation=True) for t in strings]
    ids = [t.input_ids[0] for t in toks]
    lens = [t.input_ids.ne(tokenizer.pad_token_id).sum().item() for t in toks]
    return dict(input_ids=ids, labels=ids, input_ids_lens=lens, labels_lens=lens)


def _render(sources, conv):
    roles = {"human": conv.roles[0], "gpt": conv.roles[1]}
    prompts = []
    for i, source in enumerate(sources):
        if roles[source[0]["from"]] != conv.roles[0]:
            source = source[1:]                                   # a conversation must open with the human turn
        conv.messages = []
        for j, sentence in enumerate(source):
            role = roles[sentence["from"]]
            assert role == conv.roles[j % 2], f"{i}"
            conv.append_message(role, sentence["value"])
        prompts.append(conv.get_prompt())
    return prompts


def _preprocess_rounds(sources, tokenizer, has_image, conv, sep):
    """preprocess_v1 / preprocess_llama_2: render with the template, tokenize, then hide everything but the assistant replies:
    per round (split at sep2) the instruction part (up to and including `sep`) minus 2 tokens is ignored; a length mismatch between
    the walked rounds and the tokenized prompt voids the whole sample."""
    prompts = _render(sources, conv)
    if has_image:
        input_ids = torch.stack([tokenizer_modal_token(p, tokenizer, return_tensors="pt") for p in prompts], dim=0)
    else:
        input_ids = tokenizer(prompts, return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length, truncation=True).input_ids
    targets = input_ids.clone()
    count = (lambda s: len(tokenizer_modal_token(s, tokenizer))) if has_image else (lambda s: len(tokenizer(s).input_ids))
    for prompt, target in zip(prompts, targets):
        total = int(target.ne(tokenizer.pad_token_id).sum())
        cur = 1
        target[:cur] = IGNORE_INDEX
        for rou in prompt.split(conv.sep2):
            if rou == "":
                break
            parts = rou.split(sep)
            if len(parts) != 2:
                break
            target[cur:cur + count(parts[0] + sep) - 2] = IGNORE_INDEX
            cur += count(rou)
        target[cur:] = IGNORE_INDEX
        if cur < tokenizer.model_max_length and cur != total:
            target[:] = IGNORE_INDEX
    return dict(input_ids=input_ids, labels=targets)


def preprocess_v1(sources, tokenizer, has_image: bool = False) -> Dict:
    conv = conversation_lib.default_conversation.copy()
    assert conv.sep_style == conversation_lib.SeparatorStyle.TWO
    return _preprocess_rounds(sources, tokenizer, has_image, conv, conv.sep + conv.roles[1] + ": ")


def preprocess_llama_2(sources, tokenizer, has_image: bool = False) -> Dict:
    conv = conversation_lib.default_conversation.copy()
    assert conv.sep_style == conversation_lib.SeparatorStyle.LLAMA_2
    return _preprocess_rounds(sources, tokenizer, has_image, conv, "[/INST] ")


def preprocess_mpt(sources, tokenizer) -> Dict:
    conv = conversation_lib.default_conversation.copy()
    assert conv.sep_style == conversation_lib.SeparatorStyle.MPT
    prompts = _render(sources, conv)
    input_ids = torch.stack([tokenizer_modal_token(p, tokenizer, return_tensors="pt") for p in prompts], dim=0)
    targets = input_ids.clone()
    sep = conv.sep + conv.roles[1]
    for prompt, target in zip(prompts, targets):
        total = int(target.ne(tokenizer.pad_token_id).sum())
        rounds = prompt.split(conv.sep)
        merged = [conv.sep.join(rounds[:3])] + [conv.sep.join(rounds[i:i + 2]) for i in range(3, len(rounds), 2)]
        cur = 0
        for rou in merged:
            if rou == "":
                break
            parts = rou.split(sep)
            if len(parts) != 2:
                break
            target[cur:cur + len(tokenizer_modal_token(parts[0] + sep, tokenizer))] = IGNORE_INDEX
            cur += len(tokenizer_modal_token(rou, tokenizer)) + len(tokenizer_image_token(conv.sep, tokenizer))
        target[cur:] = IGNORE_INDEX
        if cur < tokenizer.model_max_length and cur != total:
            target[:] = IGNORE_INDEX
            print(f"WARNING: tokenization mismatch: {cur} vs. {total}. (ignored)")
    return dict(input_ids=input_ids, labels=targets)


def preprocess_plain(sources, tokenizer) -> Dict:
    """Stage-1 (projector pre-training) format: '<placeholder>' + caption + sep; only the caption is a target."""
    prompts = []
    for source in sources:
        assert len(source) == 2
        prompts.append(source[0]["value"] + source[1]["value"] + conversation_lib.default_conversation.sep)
    input_ids = [tokenizer_modal_token(p, tokenizer, return_tensors="pt") for p in prompts]
    targets = copy.deepcopy(input_ids)
    for target, source in zip(targets, sources):
        target[:len(tokenizer_modal_token(source[0]["value"], tokenizer))] = IGNORE_INDEX
    return dict(input_ids=input_ids, labels=targets)


def preprocess(sources, tokenizer, has_image: bool = False) -> Dict:
    """Dispatch on the active template (conversation.default_conversation), data/utils.py:326-370."""
    conv = conversation_lib.default_conversation
    if conv.sep_style == conversation_lib.SeparatorStyle.PLAIN:
        return preprocess_plain(sources, tokenizer)
    if conv.sep_style == conversation_lib.SeparatorStyle.LLAMA_2:
        return preprocess_llama_2(sources, tokenizer, has_image=has_image)
    if conv.version.startswith("v1"):
        return preprocess_v1(sources, tokenizer, has_image=has_image)
    if conv.version == "mpt":
        return preprocess_mpt(sources, tokenizer)
    header = f"{conv.system}\n\n"
    prompts = [_add_speaker_and_signal(header, source) for source in sources]
    if has_image:
        input_ids = [tokenizer_modal_token(p, tokenizer, return_tensors="pt") for p in prompts]
    else:
        input_ids = _tokenize_fn(prompts, tokenizer)["input_ids"]
    targets = copy.deepcopy(input_ids)
    for target, source in zip(targets, sources):
        texts = [header] + [s["value"] for s in source]
        lens = [len(tokenizer_modal_token(t, tokenizer)) for t in texts] if has_image else _tokenize_fn(texts, tokenizer)["input_ids_lens"]
        _mask_targets(target, lens, [s["from"] for s in source])
    return dict(input_ids=input_ids, labels=targets)


class MultimodalDataset(torch.utils.data.Dataset):
    """json list of {'id', 'conversations': [{'from', 'value'}...], 'modal_inputs': {modal: [file, ...]}} (multimodal_dataset.py:51-131).
    `video_loader` (optional) maps the stored video entries to (3, T, 224, 224) tensors like the reference's in-dataset video
    processor call (:101-103); without it the entries are passed through to the collator's video processor."""

    def __init__(self, data_path, tokenizer, data_args=None, video_loader=None):
        super().__init__()
        self.data_args, self.tokenizer, self.video_loader = data_args, tokenizer, video_loader
        self.data = json.load(open(data_path)) if isinstance(data_path, str) else list(data_path)

    def __len__(self):
        return len(self.data)

    @property
    def modality_lengths(self):
        """:68-84: word count, negative for language-only samples, plus nominal token counts of vision / video blocks (the sampler's key)."""
        out = []
        for sample in self.data:
            n = sum(len(conv["value"].split()) for conv in sample["conversations"])
            mi = sample.get("modal_inputs", {})
            if len(mi) == 0:
                n = -n
            if "vision" in mi:
                n += 256
            if "video" in mi:
                n += 257 if mi["video"][0].endswith(".jpg") else 257 * 8
            out.append(n)
        return out

    def get_modal_inputs(self, modal_inputs):
        from PIL import Image
        for modal in modal_inputs:
            if modal == "vision":
                modal_inputs[modal] = [im if isinstance(im, Image.Image) else Image.open(im).convert("RGB") for im in modal_inputs[modal]]
            elif modal == "video" and self.video_loader is not None:
                modal_inputs[modal] = self.video_loader(modal_inputs[modal])
        return modal_inputs

    def __getitem__(self, index):
        example = copy.deepcopy(self.data[index])
        try:
            modal_inputs = self.get_modal_inputs(example.get("modal_inputs", {}))
        except Exception:                                          # corrupted media file: draw another sample (:113-118)
            new_index = random.randint(0, len(self.data) - 1)
            print(f"Corrupted: {index}, try {new_index}")
            return self.__getitem__(new_index)
        d = preprocess([example["conversations"]], self.tokenizer, has_image=len(modal_inputs) != 0)
        if isinstance(index, int):
            d = dict(input_ids=d["input_ids"][0], labels=d["labels"][0])
        d["modal_inputs"] = modal_inputs
        return d


class ChunkedMultimodalDataset(MultimodalDataset):
    """eval/model_multimodal_qa_loader.py:36-48: rank k of n evaluates the k-th contiguous chunk (dist.get_chunk)."""

    def __init__(self, data_path, tokenizer, data_args=None, num_chunks=1, chunk_idx=0, video_loader=None):
        super().__init__(data_path, tokenizer, data_args, video_loader)
        from .dist import get_chunk
        self.data = get_chunk(self.data, num_chunks, chunk_idx)


class _Cfg:
    def __init__(self, d):
        for k, v in d.items():
            setattr(self, k, v)


@dataclass
class DataCollatorForSupervisedDataset:
    """multimodal_dataset.py:139-214: right-pad ids (pad id) and labels (-100), truncate to model_max_length, attention mask = non-pad;
    concatenate every sample's per-modality items in batch order and run each modality's processor once."""
    tokenizer: object
    modal_processors: dict
    modal_configs: dict = None

    def __call__(self, instances: Sequence[Dict]) -> Dict[str, torch.Tensor]:
        pad = self.tokenizer.pad_token_id
        input_ids = torch.nn.utils.rnn.pad_sequence([i["input_ids"] for i in instances], batch_first=True, padding_value=pad)
        labels = torch.nn.utils.rnn.pad_sequence([i["labels"] for i in instances], batch_first=True, padding_value=IGNORE_INDEX)
        input_ids = input_ids[:, :self.tokenizer.model_max_length]
        labels = labels[:, :self.tokenizer.model_max_length]
        batch = dict(input_ids=input_ids, labels=labels, attention_mask=input_ids.ne(pad))
        if "modal_inputs" in instances[0]:
            merged = defaultdict(list)
            for inst in instances:
                for modal, items in inst["modal_inputs"].items():
                    merged[modal].extend(items)
            batch["modal_inputs"] = self.process_modal_inputs(merged)
        return batch

    def process_modal_inputs(self, modal_inputs):
        from .mm_utils import process_images
        out = {}
        for key, items in modal_inputs.items():
            proc = self.modal_processors[key]
            if key == "text":
                out[key] = proc(items, return_tensors="pt", padding=True)
            elif key == "vision":
                cfg = _Cfg(self.modal_configs["vision"]) if isinstance(self.modal_configs, dict) and "vision" in self.modal_configs else _Cfg({})
                out[key] = process_images(items, proc, cfg)
            elif key == "audio":
                feats, mask = proc(items)
                out[key] = {"audio_inputs": feats, "audio_padding_mask": mask}
            elif key == "video":
                if not all(torch.is_tensor(v) for v in items):
                    items = list(proc(items)["pixel_values"])
                frames = max(v.shape[1] for v in items)             # single-frame (.jpg) items are repeated to the clip length (:194-198)
                items = [v if v.shape[1] == frames else v.expand(-1, frames, -1, -1) for v in items]
                out[key] = torch.stack(items, dim=0)
            elif key == "point":
                out[key] = proc(items)
        return out


def make_multimodal_data_module(tokenizer, data_args, modal_data_configs) -> Dict:
    """data/__init__.py:5-17: dataset + collator for supervised finetuning (train_multimodal.py:484-486)."""
    train_dataset = MultimodalDataset(data_path=data_args.data_path, tokenizer=tokenizer, data_args=data_args)
    collator = DataCollatorForSupervisedDataset(tokenizer=tokenizer, modal_processors=data_args.modal_processors, modal_configs=modal_data_configs)
    return dict(train_dataset=train_dataset, eval_dataset=None, data_collator=collator)
