"""Host-side mirror of the helpers of modelcompose/mm_utils.py that sit on the path (image batch preparation, model name)."""
from __future__ import annotations

import torch


def process_images(images, image_processor, model_cfg):
    """mm_utils.py:29-44: image_aspect_ratio == 'pad' -> expand2square with the processor's mean colour, then preprocess; else the
    processor on the whole list.  With the HIP processor the padding is part of the same kernels (a virtual canvas)."""
    aspect = getattr(model_cfg, "image_aspect_ratio", None)
    if aspect == "pad":
        new_images = [image_processor.preprocess(im, return_tensors="pt", pad_to_square=True)["pixel_values"][0] for im in images]
        if all(x.shape == new_images[0].shape for x in new_images):
            new_images = torch.stack(new_images, dim=0)
        return new_images
    return image_processor(images, return_tensors="pt")["pixel_values"]


def get_model_name_from_path(model_path: str) -> str:
    """mm_utils.py:103-109."""
    model_path = model_path.strip("/")
    parts = model_path.split("/")
    if parts[-1].startswith("checkpoint-"):
        return parts[-2] + "_" + parts[-1]
    return parts[-1]


def load_image_from_base64(image):
    """mm_utils.py:10-11."""
    import base64
    from io import BytesIO
    from PIL import Image
    return Image.open(BytesIO(base64.b64decode(image)))


def expand2square(pil_img, background_color):
    """mm_utils.py:14-25: centre the image on a square canvas of the longer side (the HIP image processor does the same padding
    inside its resize kernel when called with pad_to_square=True; this PIL form is kept for callers that want the image)."""
    from PIL import Image
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    canvas = Image.new(pil_img.mode, (side, side), background_color)
    canvas.paste(pil_img, ((side - w) // 2, (side - h) // 2))
    return canvas


def _to_ids(input_ids, return_tensors):
    if return_tensors is None:
        return input_ids
    if return_tensors == "pt":
        return torch.tensor(input_ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


def tokenizer_image_token(prompt, tokenizer, image_token_index=None, return_tensors=None):
    """mm_utils.py:43-62: tokenize the text between '<image>' placeholders separately and join the pieces with the sentinel id;
    a BOS produced for every piece is kept once, at the front."""
    from .constants import IMAGE_TOKEN_INDEX
    sentinel = IMAGE_TOKEN_INDEX if image_token_index is None else image_token_index
    pieces = [tokenizer(chunk).input_ids for chunk in prompt.split("<image>")]
    has_bos = len(pieces) > 0 and len(pieces[0]) > 0 and pieces[0][0] == tokenizer.bos_token_id
    skip = 1 if has_bos else 0
    ids = [pieces[0][0]] if has_bos else []
    for n, piece in enumerate(pieces):
        if n:
            ids.append(sentinel)
        ids.extend(piece[skip:])
    return _to_ids(ids, return_tensors)


def split_string_by_list(input_string, split_list):
    """mm_utils.py:64-79: cut the prompt at every placeholder of split_list, scanning left to right; returns (text, placeholder)
    pairs, the trailing text paired with None.  When several placeholders end at the same character the first in split_list wins."""
    out, cur = [], ""
    for ch in input_string:
        cur += ch
        hit = next((sep for sep in split_list if sep in cur), None)
        if hit is not None:
            out.append((cur.split(hit, 1)[0], hit))
            cur = ""
    if cur:
        out.append((cur, None))
    return out


def tokenizer_modal_token(prompt, tokenizer, return_tensors=None):
    """mm_utils.py:81-101: like tokenizer_image_token for every modality placeholder ('<image>', '<audio>', '<video>', '<point>', ...):
    each becomes its sentinel id (-200 ... -205), the input contract of the splice."""
    from .constants import MODAL_TOKEN_MAPPING
    chunks = split_string_by_list(prompt, list(MODAL_TOKEN_MAPPING.keys()))
    pieces = [tokenizer(text).input_ids for text, _ in chunks]
    has_bos = len(pieces) > 0 and len(pieces[0]) > 0 and pieces[0][0] == tokenizer.bos_token_id
    skip = 1 if has_bos else 0
    ids = [pieces[0][0]] if has_bos else []
    for piece, (_, sep) in zip(pieces, chunks):
        ids.extend(piece[skip:])
        if sep is not None:
            ids.append(MODAL_TOKEN_MAPPING[sep])
    return _to_ids(ids, return_tensors)


class KeywordsStoppingCriteria:
    """mm_utils.py:114-144: stop when the generated tail equals a keyword's token ids or the decoded tail contains a keyword
    (batch size 1).  Callable as criteria(output_ids, scores) like a transformers StoppingCriteria; generate() evaluates it after
    every token when passed in `stopping_criteria`."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        self.keyword_ids = []
        self.max_keyword_len = 0
        for keyword in keywords:
            ids = tokenizer(keyword).input_ids
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.max_keyword_len = max(self.max_keyword_len, len(ids))
            self.keyword_ids.append(torch.tensor(ids))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def __call__(self, output_ids, scores=None, **kwargs) -> bool:
        assert output_ids.shape[0] == 1, "Only support batch size 1 (yet)"
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        self.keyword_ids = [k.to(output_ids.device) for k in self.keyword_ids]
        for k in self.keyword_ids:
            if (output_ids[0, -k.shape[0]:] == k).all():
                return True
        text = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(keyword in text for keyword in self.keywords)
