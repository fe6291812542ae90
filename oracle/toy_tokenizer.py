"""TEST INFRASTRUCTURE: a deterministic stand-in for the LLaMA tokenizer (no sentencepiece model travels with the repo) with the three
members the reference's prompt helpers touch: __call__(text).input_ids (BOS first), bos_token_id, batch_decode."""
from types import SimpleNamespace


class ToyTokenizer:
    bos_token_id = 1
    pad_token_id = 0
    eos_token_id = 2

    def __init__(self, add_bos=True, model_max_length=2048):
        self.add_bos = add_bos
        self.model_max_length = model_max_length
        self.vocab = {}
        self.inv = {}

    def _id(self, piece):
        if piece not in self.vocab:
            i = 3 + len(self.vocab)
            self.vocab[piece] = i
            self.inv[i] = piece
        return self.vocab[piece]

    def __call__(self, text, return_tensors=None, padding=None, max_length=None, truncation=False):
        if isinstance(text, (list, tuple)) or return_tensors == "pt":
            import torch
            rows = [self(t).input_ids for t in ([text] if isinstance(text, str) else text)]
            if truncation and max_length:
                rows = [r[:max_length] for r in rows]
            width = max(len(r) for r in rows)
            return SimpleNamespace(input_ids=torch.tensor([r + [self.pad_token_id] * (width - len(r)) for r in rows], dtype=torch.long))
        # pieces: runs of non-space characters and the single spaces between them, so that decode(encode(x)) == x
        pieces, cur = [], ""
        for ch in text:
            if ch == " ":
                if cur:
                    pieces.append(cur)
                    cur = ""
                pieces.append(" ")
            else:
                cur += ch
        if cur:
            pieces.append(cur)
        ids = ([self.bos_token_id] if self.add_bos else []) + [self._id(p) for p in pieces]
        return SimpleNamespace(input_ids=ids)

    def batch_decode(self, ids, skip_special_tokens=True):
        out = []
        for row in ids:
            out.append("".join(self.inv.get(int(i), "") for i in row if not (skip_special_tokens and int(i) < 3)))
        return out


class FakeProc:
    """stand-in modal processors for the collator fixtures (deterministic functions of their inputs)"""
    image_mean = (0.5, 0.25, 0.125)

    def __init__(self, kind):
        self.kind = kind

    def preprocess(self, image, return_tensors="pt", **kw):
        import numpy as _np
        import torch
        a = torch.from_numpy(_np.asarray(image).astype(_np.float32)).permute(2, 0, 1)
        return {"pixel_values": [a[:, :2, :2] / 255.0]}

    def __call__(self, items, **kw):
        import torch
        if self.kind == "vision":
            return {"pixel_values": torch.stack([self.preprocess(i)["pixel_values"][0] for i in items])}
        if self.kind == "audio":
            n = len(items)
            return torch.arange(n * 6, dtype=torch.float32).view(n, 3, 2), torch.zeros(n, 3, dtype=torch.bool)
        if self.kind == "point":
            return torch.stack([torch.as_tensor(i, dtype=torch.float32) for i in items])
        raise KeyError(self.kind)
