"""TEST INFRASTRUCTURE: a deterministic stand-in for the LLaMA tokenizer (no sentencepiece model travels with the repo) with the three
members the reference's prompt helpers touch: __call__(text).input_ids (BOS first), bos_token_id, batch_decode."""
from types import SimpleNamespace


class ToyTokenizer:
    bos_token_id = 1

    def __init__(self, add_bos=True):
        self.add_bos = add_bos
        self.vocab = {}
        self.inv = {}

    def _id(self, piece):
        if piece not in self.vocab:
            i = 3 + len(self.vocab)
            self.vocab[piece] = i
            self.inv[i] = piece
        return self.vocab[piece]

    def __call__(self, text):
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
