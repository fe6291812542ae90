"""The evaluation loop that produces the headline metric's samples: question file -> prompts -> generate -> answers.jsonl.
Mirror of modelcompose/eval/model_multimodal_qa_loader.py:24-154 with the same CLI (`--model-path --model-base --question-file
--answers-file --conv-mode --num-chunks --chunk-idx --temperature --top_p --num_beams`), so scripts/model_composition/test/*.sh run
unchanged: one process per GPU, rank k evaluating chunk k (the reference shards by process and concatenates the answer files).

Differences, all opt-in: `--batch-size B` runs B questions per generate() call (the reference is fixed at 1; prompts in a batch are
right-padded by the collator and spliced per row); `--pipeline` overlaps the decode of one batch with the prefill of the next;
`--max-new-tokens` (reference: constant 128); answer ids come from a counter-free
uuid4 hex prefix instead of shortuuid (absent from this image)."""
from __future__ import annotations

import argparse
import json
import os
import uuid

import torch

from .. import _lib
from .. import conversation as conversation_lib
from ..conversation import SeparatorStyle, conv_templates
from ..data import ChunkedMultimodalDataset, DataCollatorForSupervisedDataset
from ..dist import get_chunk, split_list  # noqa: F401
from ..mm_utils import get_model_name_from_path
from ..model.builder import load_pretrained_model


def create_data_loader(data_path, tokenizer, modal_processors, num_chunks=1, chunk_idx=0, batch_size=1, num_workers=0):
    """:51-54 (image_aspect_ratio 'pad' for the vision processor).  num_workers defaults to 0: the modal processors run on the GPU and
    must not be forked into DataLoader workers."""
    dataset = ChunkedMultimodalDataset(data_path, tokenizer, None, num_chunks, chunk_idx)
    collate = DataCollatorForSupervisedDataset(tokenizer, modal_processors, {"vision": {"image_aspect_ratio": "pad"}})
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, num_workers=num_workers, shuffle=False, collate_fn=collate)


def _to_device(modal_inputs, device, dtype):
    out = {}
    for modal, v in modal_inputs.items():
        if isinstance(v, list):
            out[modal] = [x.to(device=device, dtype=dtype) for x in v]
        elif isinstance(v, dict):
            out[modal] = {k: (x.to(device=device, dtype=dtype) if x.is_floating_point() else x.to(device)) for k, x in v.items()}
        else:
            out[modal] = v.to(device=device, dtype=dtype)
    return out


def eval_model(args, loaded=None):
    """`loaded` = (tokenizer, model, modal_processors, context_len) skips load_pretrained_model (tests)."""
    model_path = os.path.expanduser(args.model_path)
    model_name = get_model_name_from_path(model_path)
    tokenizer, model, modal_processors, _ = loaded or load_pretrained_model(model_path, args.model_base, model_name)
    conversation_lib.default_conversation = conv_templates[args.conv_mode]
    tokenizer.pad_token_id = tokenizer.eos_token_id                                        # :67
    answers_file = os.path.expanduser(args.answers_file)
    os.makedirs(os.path.dirname(answers_file) or ".", exist_ok=True)
    batch_size = getattr(args, "batch_size", 1)
    loader = create_data_loader(args.question_file, tokenizer, modal_processors, args.num_chunks, args.chunk_idx, batch_size=batch_size)
    questions = get_chunk(json.load(open(args.question_file)), args.num_chunks, args.chunk_idx)
    conv = conv_templates[args.conv_mode]
    stop_str = conv.sep if conv.sep_style != SeparatorStyle.TWO else conv.sep2
    n_done = 0
    gen_kw = dict(do_sample=args.temperature > 0, temperature=args.temperature, top_p=args.top_p, num_beams=args.num_beams,
                  max_new_tokens=getattr(args, "max_new_tokens", 128), use_cache=True)
    meta = []                                                       # prompt length of every batch handed to the model, in order

    def batches():
        for batch in loader:
            input_ids = batch["input_ids"].to(model.device)
            modal_inputs = _to_device(batch["modal_inputs"], model.device, _lib.storage_dtype()) if "modal_inputs" in batch else {}
            meta.append(input_ids)
            # batch 1 (the reference): no mask, as the reference's generate() call.  Batched: the collator right-pads the prompts; the mask
            # gives every row its own length, so each row generates exactly what it would alone
            am = batch["attention_mask"].to(model.device) if input_ids.shape[0] > 1 and "attention_mask" in batch else None
            yield input_ids, modal_inputs, am

    def results():
        if getattr(args, "pipeline", False):                        # decode of batch i beside the prefill of batch i+1
            yield from model.generate_pipelined(batches(), **gen_kw)
        else:
            for input_ids, modal_inputs, am in batches():
                with torch.inference_mode():
                    yield model.generate(input_ids, modal_inputs=modal_inputs, attention_mask=am, **gen_kw)

    with open(answers_file, "w") as ans_file:
        for k, output_ids in enumerate(results()):
            input_ids = meta[k]
            n_in = input_ids.shape[1]
            if int((input_ids != output_ids[:, :n_in]).sum().item()) > 0:
                print("[Warning] output_ids are not the same as the input_ids")
            texts = tokenizer.batch_decode(output_ids[:, n_in:], skip_special_tokens=True)
            for text in texts:
                text = text.strip()
                if text.endswith(stop_str):
                    text = text[:-len(stop_str)]
                q = questions[n_done]
                ans_file.write(json.dumps({"question_id": q["id"], "prompt": q["conversations"][0]["value"], "text": text.strip(),
                                           "answer_id": uuid.uuid4().hex[:22], "model_id": model_name, "metadata": {}}) + "\n")
                n_done += 1
            ans_file.flush()
    return n_done


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--model-path", type=str, default="facebook/opt-350m")
    p.add_argument("--model-base", type=str, default=None)
    p.add_argument("--image-folder", type=str, default="")
    p.add_argument("--question-file", type=str, default="tables/question.jsonl")
    p.add_argument("--answers-file", type=str, default="answer.jsonl")
    p.add_argument("--conv-mode", type=str, default="llava_v1")
    p.add_argument("--num-chunks", type=int, default=1)
    p.add_argument("--chunk-idx", type=int, default=0)
    p.add_argument("--temperature", type=float, default=0.2)
    p.add_argument("--top_p", type=float, default=None)
    p.add_argument("--num_beams", type=int, default=1)
    p.add_argument("--no_add_image_token", action="store_true")
    p.add_argument("--batch-size", type=int, default=1)
    p.add_argument("--max-new-tokens", type=int, default=128)
    p.add_argument("--pipeline", action="store_true", help="overlap the decode of one batch with the prefill of the next (generate_pipelined)")
    args = p.parse_args(argv)
    if args.model_base == "" or args.model_base == "None":                                  # :141-142
        args.model_base = None
    return args


if __name__ == "__main__":
    eval_model(parse_args())
