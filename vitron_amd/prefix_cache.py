"""Multi-turn reuse for the chat loop.

The reference's `app.py:predict` re-renders the WHOLE conversation every turn (`app.py:493-514`) and calls
`model.generate(input_ids, images=[every image/video so far], ...)` (`app.py:562-571`): all towers and the whole
prefill run again although only the last user message is new. Two host-side caches remove that work without touching
the call signature:

  * VisualFeatureCache -- projected visual tokens (and the region feature) of an image / clip, keyed by a checksum of
    its pixel tensor (+ the box): a history image is encoded once.
  * PrefixKV -- the paged KV of the previous `generate` call stays allocated together with a per-row signature
    (token id, or (visual block key, row) for spliced rows); the next call prefills only the rows behind the longest
    common prefix, rounded DOWN to a 64-token page (a page is the unit the kernels append to, and a page that starts
    with a new token is rewritten completely, so no stale key can survive).

Both are exact-match caches: a different pixel, box or token changes the signature and falls back to full work.
"""
from __future__ import annotations

import collections
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

PAGE = 64
_MASK40 = (1 << 40) - 1


def tensor_key(t: torch.Tensor) -> Tuple:
    """Content key of a pixel tensor: shape, dtype and a 128-bit BLAKE2b digest of its bytes (one device -> host copy of
    the tensor: 0.3 MB for a 224 px image, 5.4 MB for an 8-frame 336 px clip). A cryptographic digest, not a checksum: two
    different images sharing a key would silently reuse each other's visual tokens and KV prefix."""
    import hashlib
    x = t.detach().contiguous()
    raw = x.view(torch.uint8).reshape(-1).cpu().numpy()
    return (tuple(x.shape), str(x.dtype), hashlib.blake2b(raw.tobytes(), digest_size=16).hexdigest())


def key64(key) -> int:
    """Stable 40-bit hash of a python key (tuples of ints/strings/floats)."""
    h = 1469598103934665603
    for ch in repr(key).encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h & _MASK40


class VisualFeatureCache:
    """LRU map: (pixel key, box or None) -> (visual tokens [rows, H] bf16, region feature [1, H] or None)."""

    def __init__(self, max_entries: int = 16):
        self.max_entries = int(max_entries)
        self._d: "collections.OrderedDict" = collections.OrderedDict()
        self.hits = 0
        self.misses = 0

    def get(self, key):
        v = self._d.get(key)
        if v is None:
            self.misses += 1
            return None
        self._d.move_to_end(key)
        self.hits += 1
        return v

    def put(self, key, feats: torch.Tensor, region: Optional[torch.Tensor]):
        if self.max_entries <= 0:
            return
        self._d[key] = (feats, region)
        self._d.move_to_end(key)
        while len(self._d) > self.max_entries:
            self._d.popitem(last=False)

    def clear(self):
        self._d.clear()


def row_signature(plan: np.ndarray, block_keys: Sequence[Tuple[int, int, int]], region_keys: Sequence[int]) -> np.ndarray:
    """int64 signature per spliced row. plan [S, 2] = (kind, index) as built by build_splice_plan_np
    (0 token id, 1 visual row, 2 region row, 3 padding); block_keys = (first visual row, rows, 40-bit key) per feature
    block; region_keys = 40-bit key per region row. Equal signatures <=> the rows hold the same embedding."""
    kind = plan[:, 0].astype(np.int64)
    idx = plan[:, 1].astype(np.int64)
    sig = np.where(kind == 0, idx, np.int64(-1))
    if len(block_keys):
        starts = np.array([b[0] for b in block_keys], dtype=np.int64)
        keys = np.array([b[2] for b in block_keys], dtype=np.int64)
        vis = kind == 1
        if vis.any():
            blk = np.searchsorted(starts, idx[vis], side="right") - 1
            local = idx[vis] - starts[blk]
            sig[vis] = (np.int64(1) << 62) | (keys[blk] << 20) | local
    reg = kind == 2
    if reg.any():
        rk = np.array(list(region_keys), dtype=np.int64)
        sig[reg] = (np.int64(3) << 61) | rk[idx[reg]]
    return sig


def common_prefix(a: np.ndarray, b: np.ndarray) -> int:
    n = min(len(a), len(b))
    if n == 0:
        return 0
    ne = np.nonzero(a[:n] != b[:n])[0]
    return int(ne[0]) if len(ne) else n


def reusable_tokens(cached_sig: np.ndarray, new_sig: np.ndarray) -> int:
    """Tokens of the cached sequence the new prompt can keep: the common prefix, at most len(new) - 1 (the last prompt row
    must run to produce logits), rounded down to whole pages."""
    p = min(common_prefix(cached_sig, new_sig), len(new_sig) - 1)
    return max(p, 0) // PAGE * PAGE


class PrefixKV:
    """Pages + row signature of the last finished generate() call (batch 1)."""

    def __init__(self, sig: np.ndarray, pages: List[int]):
        self.sig = sig
        self.pages = pages
