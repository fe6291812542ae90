"""Prompt templates of the reference (modelcompose/conversation.py:6-381): the text either side of the modality placeholders that
tokenizer_modal_token turns into the spliced input ids.  Host-side string formatting only; the template strings must be byte-identical
to the reference's for the token ids (and so the generated text) to match, so they are data, restated here for the templates the
composed-Vicuna path uses (v1 / vicuna_v1, llava_v1, plain, llama_2, llava_llama_2, llava_v0, mpt).

Differences in scope: the demo-only helpers (get_images, to_gradio_chatbot, the 'mmtag' variants and the long few-shot v0 template)
belong to the reference's web UI and are not part of the path."""
from __future__ import annotations

import dataclasses
from enum import Enum, auto
from typing import List, Optional


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()
    MPT = auto()
    PLAIN = auto()
    LLAMA_2 = auto()


def _text(message):
    return message[0] if type(message) is tuple else message


@dataclasses.dataclass
class Conversation:
    system: str
    roles: List[str]
    messages: List[List[str]]
    offset: int
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "###"
    sep2: Optional[str] = None
    version: str = "Unknown"
    skip_next: bool = False

    def get_prompt(self) -> str:
        """conversation.py:29-107."""
        msgs = self.messages
        if len(msgs) > 0 and type(msgs[0][1]) is tuple:          # first turn carries an image: normalise to '<image>\\n' + text
            msgs = list(msgs)
            role, first = msgs[0]
            msgs[0] = (role, "<image>\n" + first[0].replace("<image>", "").strip())
        style = self.sep_style
        if style == SeparatorStyle.SINGLE:
            out = self.system + self.sep
            for role, m in msgs:
                out += (role + ": " + _text(m) + self.sep) if m else (role + ":")
            return out
        if style == SeparatorStyle.TWO:
            seps = (self.sep, self.sep2)
            out = self.system + seps[0]
            for i, (role, m) in enumerate(msgs):
                out += (role + ": " + _text(m) + seps[i % 2]) if m else (role + ":")
            return out
        if style == SeparatorStyle.MPT:
            out = self.system + self.sep
            for role, m in msgs:
                out += (role + _text(m) + self.sep) if m else role
            return out
        if style == SeparatorStyle.LLAMA_2:
            out = ""
            for i, (role, m) in enumerate(msgs):
                if i == 0:
                    assert m, "first message should not be none"
                    assert role == self.roles[0], "first message should come from user"
                if not m:
                    continue
                m = _text(m)
                if i == 0:
                    m = f"<<SYS>>\n{self.system}\n<</SYS>>\n\n" + m
                out += (self.sep + f"[INST] {m} [/INST]") if i % 2 == 0 else (" " + m + " " + self.sep2)
            return out.lstrip(self.sep)
        if style == SeparatorStyle.PLAIN:
            seps = (self.sep, self.sep2)
            out = self.system
            for i, (_, m) in enumerate(msgs):
                if m:
                    out += _text(m) + seps[i % 2]
            return out
        raise ValueError(f"Invalid style: {style}")

    def append_message(self, role, message):
        self.messages.append([role, message])

    def copy(self) -> "Conversation":
        return Conversation(system=self.system, roles=self.roles, messages=[[r, m] for r, m in self.messages], offset=self.offset,
                            sep_style=self.sep_style, sep=self.sep, sep2=self.sep2, version=self.version)

    def dict(self):
        return {"system": self.system, "roles": self.roles, "messages": [[r, _text(m)] for r, m in self.messages], "offset": self.offset,
                "sep": self.sep, "sep2": self.sep2}


_CHAT_USER = ("A chat between a curious user and an artificial intelligence assistant. "
              "The assistant gives helpful, detailed, and polite answers to the user's questions.")
_CHAT_HUMAN = ("A chat between a curious human and an artificial intelligence assistant. "
               "The assistant gives helpful, detailed, and polite answers to the human's questions.")

conv_vicuna_v1 = Conversation(system=_CHAT_USER, roles=("USER", "ASSISTANT"), version="v1", messages=(), offset=0,
                              sep_style=SeparatorStyle.TWO, sep=" ", sep2="</s>")
conv_llava_v1 = Conversation(system=_CHAT_HUMAN, roles=("USER", "ASSISTANT"), version="v1", messages=(), offset=0,
                             sep_style=SeparatorStyle.TWO, sep=" ", sep2="</s>")
conv_llava_v0 = Conversation(system=_CHAT_HUMAN, roles=("Human", "Assistant"), messages=(), offset=0, sep_style=SeparatorStyle.SINGLE, sep="###")
conv_llava_plain = Conversation(system="", roles=("", ""), messages=(), offset=0, sep_style=SeparatorStyle.PLAIN, sep="\n")
conv_llama_2 = Conversation(
    system="You are a helpful, respectful and honest assistant. Always answer as helpfully as possible, while being safe.  "
           "Your answers should not include any harmful, unethical, racist, sexist, toxic, dangerous, or illegal content. "
           "Please ensure that your responses are socially unbiased and positive in nature.\n\n"
           "If a question does not make any sense, or is not factually coherent, explain why instead of answering something not correct. "
           "If you don't know the answer to a question, please don't share false information.",
    roles=("USER", "ASSISTANT"), version="llama_v2", messages=(), offset=0, sep_style=SeparatorStyle.LLAMA_2, sep="<s>", sep2="</s>")
conv_llava_llama_2 = Conversation(
    system="You are a helpful language and vision assistant. You are able to understand the visual content that the user provides, "
           "and assist the user with a variety of tasks using natural language.",
    roles=("USER", "ASSISTANT"), version="llama_v2", messages=(), offset=0, sep_style=SeparatorStyle.LLAMA_2, sep="<s>", sep2="</s>")
conv_mpt = Conversation(
    system="<|im_start|>system\nA conversation between a user and an LLM-based AI assistant. The assistant gives helpful and honest answers.",
    roles=("<|im_start|>user\n", "<|im_start|>assistant\n"), version="mpt", messages=(), offset=0, sep_style=SeparatorStyle.MPT, sep="<|im_end|>")

default_conversation = conv_vicuna_v1
conv_templates = {"v1": conv_vicuna_v1, "vicuna_v1": conv_vicuna_v1, "llama_2": conv_llama_2, "plain": conv_llava_plain,
                  "v0_plain": conv_llava_plain, "llava_v0": conv_llava_v0, "llava_v1": conv_llava_v1, "llava_llama_2": conv_llava_llama_2,
                  "mpt": conv_mpt}
