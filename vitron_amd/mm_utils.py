"""Prompt -> ids plumbing on the boundary of the hot path: host-side mirror of the reference's vitron/mm_utils.py
(same names, arguments and return values), so app.py / inference_image.py keep working unchanged."""
from __future__ import annotations

import torch

from .constants import IMAGE_TOKEN_INDEX, OBJS_TOKEN_INDEX


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
    """reference mm_utils.py:146-177 -- stop when the tail of the output matches a keyword's ids, or the decoded
    tail contains the keyword. Called as criteria(output_ids, scores) -> bool."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        self.keyword_ids = []
        self.max_keyword_len = 0
        for kw in keywords:
            ids = tokenizer(kw).input_ids
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.max_keyword_len = max(self.max_keyword_len, len(ids))
            self.keyword_ids.append(torch.tensor(ids))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def call_for_batch(self, output_ids, scores, **kwargs):
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        self.keyword_ids = [k.to(output_ids.device) for k in self.keyword_ids]
        for k in self.keyword_ids:
            if output_ids.shape[1] >= k.shape[0] and bool((output_ids[0, -k.shape[0]:] == k).all()):
                return True
        if offset <= 0:
            return False
        text = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(kw in text for kw in self.keywords)

    def __call__(self, output_ids, scores, **kwargs):
        return all(self.call_for_batch(output_ids[i].unsqueeze(0), scores) for i in range(output_ids.shape[0]))
