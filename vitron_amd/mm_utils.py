"""Prompt -> ids plumbing on the boundary of the hot path: host-side mirror of the reference's vitron/mm_utils.py
(same names, arguments and return values), so app.py / inference_image.py keep working unchanged."""
from __future__ import annotations

import torch

from .constants import IMAGE_TOKEN_INDEX, OBJS_TOKEN_INDEX


def load_image_from_base64(image):
    """reference mm_utils.py:48-49 -- a base64 string (what the Gradio front end posts) -> PIL image."""
    import base64
    import io

    from PIL import Image
    return Image.open(io.BytesIO(base64.b64decode(image)))


def expand2square(pil_img, background_color):
    """reference mm_utils.py:51-63 -- pad the shorter side with `background_color` so that the picture sits centred on a square canvas
    (the offset of the odd pixel goes to the far side: (long - short) // 2 in front); a square picture is returned as it is."""
    from PIL import Image
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    canvas = Image.new(pil_img.mode, (side, side), background_color)
    canvas.paste(pil_img, ((side - w) // 2, (side - h) // 2))
    return canvas


def process_images(images, image_processor, model_cfg):
    """reference mm_utils.py:66-77 -- `image_aspect_ratio == 'pad'`: every picture squared on the processor's mean colour and run through
    `image_processor.preprocess` on its own, stacked when the shapes agree (a list otherwise); any other setting: the processor on the
    whole list."""
    if getattr(model_cfg, "image_aspect_ratio", None) != "pad":
        return image_processor(images, return_tensors="pt")["pixel_values"]
    fill = tuple(int(c * 255) for c in image_processor.image_mean)
    out = [image_processor.preprocess(expand2square(im, fill), return_tensors="pt")["pixel_values"][0] for im in images]
    if all(t.shape == out[0].shape for t in out):
        return torch.stack(out, dim=0)
    return out


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None, is_first=True):
    """reference mm_utils.py:80-99 -- split on '<image>', tokenise the chunks, keep one BOS (only when is_first),
    put one sentinel between chunks."""
    chunks = [tokenizer(c).input_ids for c in prompt.split("<image>")]
    input_ids, offset = [], 0
    if chunks and chunks[0] and chunks[0][0] == tokenizer.bos_token_id and is_first:
        offset = 1
        input_ids.append(chunks[0][0])
    sep = [image_token_index] * (offset + 1)
    for i, ch in enumerate(chunks):
        if i:
            input_ids.extend(sep[offset:])
        input_ids.extend(ch[offset:])
    return _ret(input_ids, return_tensors)


def tokenizer_image_region_token(prompt, tokenizer, region_token_index=OBJS_TOKEN_INDEX, return_tensors=None):
    """reference mm_utils.py:102-117 -- split on '<objs>' first; later chunks keep their BOS (is_first=False)."""
    input_ids = []
    chunks = prompt.split("<objs>")
    for idx, ck in enumerate(chunks):
        input_ids.extend(tokenizer_image_token(ck, tokenizer, is_first=(idx == 0)))
        if idx < len(chunks) - 1:
            input_ids.append(region_token_index)
    return _ret(input_ids, return_tensors)


def _ret(ids, return_tensors):
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


def preprocess_region(region, image_size, target_size, return_tensors=None):
    """reference mm_utils.py:121-135 -- rescale a box from the original image to the tower's input size."""
    x1, y1, x2, y2 = region
    sx, sy = target_size[0] / image_size[0], target_size[1] / image_size[1]
    out = [x1 * sx, y1 * sy, x2 * sx, y2 * sy]
    if return_tensors == "pt":
        return torch.tensor(out, dtype=torch.long)
    if return_tensors is not None:
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return out


def get_model_name_from_path(model_path):
    """reference mm_utils.py:138-144."""
    parts = model_path.strip("/").split("/")
    return parts[-2] + "_" + parts[-1] if parts[-1].startswith("checkpoint-") else parts[-1]


class KeywordsStoppingCriteria:
    """Stop test with the reference's semantics (mm_utils.py:146-177): a row is finished when its generated tail
    equals the ids of one stop keyword, or when the decoded tail (as many new tokens as the longest keyword has)
    contains a keyword as text; a batch is finished when every row is. Called as criteria(output_ids, scores)."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = list(keywords)
        self.tokenizer = tokenizer
        self.start_len = int(input_ids.shape[1])
        bos = tokenizer.bos_token_id
        self.keyword_ids = [self._strip_bos(tokenizer(kw).input_ids, bos) for kw in self.keywords]
        self.max_keyword_len = max((int(k.numel()) for k in self.keyword_ids), default=0)

    @staticmethod
    def _strip_bos(ids, bos):
        drop = 1 if len(ids) > 1 and ids[0] == bos else 0
        return torch.tensor(ids[drop:])

    def _row_done(self, row):  # row: [1, L]
        length = row.shape[1]
        for n, kid in enumerate(self.keyword_ids):
            if kid.device != row.device:
                kid = self.keyword_ids[n] = kid.to(row.device)
            k = int(kid.numel())
            if 0 < k <= length and torch.equal(row[0, length - k:], kid):
                return True
        window = min(length - self.start_len, self.max_keyword_len)
        if window <= 0:
            return False
        tail = self.tokenizer.batch_decode(row[:, length - window:], skip_special_tokens=True)[0]
        return any(kw in tail for kw in self.keywords)

    def call_for_batch(self, output_ids, scores, **kwargs):
        return self._row_done(output_ids[:1])

    def __call__(self, output_ids, scores, **kwargs):
        return all(self._row_done(output_ids[r:r + 1]) for r in range(output_ids.shape[0]))
