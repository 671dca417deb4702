"""Multimodal glue: host-side mirror of the reference's vitron/model/llava_arch.py.

Same public surface (LlavaMetaModel accessors, LlavaMetaForCausalLM.encode_images / encode_videos /
prepare_inputs_labels_for_multimodal) with the tensor work moved to HIP kernels:
  towers -> vt_vit_forward, region_extractor -> vt_region_forward, mm_projector -> vt_projector_forward,
  the list surgery of llava_arch.py:306-398 / :479-558 -> an integer plan built here + vt_embed_splice.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import ops
from ..engine import pair_lo, with_lo
from ..constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX, OBJS_TOKEN_INDEX

KIND_TOKEN, KIND_VISUAL, KIND_REGION, KIND_ZERO = 0, 1, 2, 3


def build_splice_plan_np(input_ids, attention_mask, feature_blocks: Sequence[Tuple[int, int]],
                         region_rows: Optional[Sequence[int]], max_length: Optional[int] = None,
                         padding_side: str = "right"):
    """Integer plan of prepare_inputs_labels_for_multimodal (reference llava_arch.py:300-398 / :474-558), numpy form.

    input_ids       [B][L] ints with -200 / -300 sentinels; attention_mask [B][L] (None = all ones)
    feature_blocks  one (first_row, n_rows) per flat visual feature (one per image, T per video) in list order
    region_rows     per flat feature: row of its region feature (or -1), None when the caller passed no regions
    Returns (plan int32 [B,S,2] = (kind, index), mask int32 [B,S], position_ids int64 [B,S], lengths list[B]).
    Quirks kept from the reference: a sample without <image> still consumes one feature slot (:317-324);
    <objs> takes the region of the most recently consumed feature (:350-351); truncate before padding (:363-366).
    """
    ids_all = np.asarray(input_ids, dtype=np.int64)
    am_all = None if attention_mask is None else np.asarray(attention_mask).astype(bool)
    use_regions = region_rows is not None
    samples = []
    cur = 0
    for b in range(ids_all.shape[0]):
        ids = ids_all[b] if am_all is None else ids_all[b][am_all[b]]
        specials = np.nonzero(ids < 0)[0]
        if not np.any(ids == IMAGE_TOKEN_INDEX):
            if specials.size:
                raise ValueError(f"sample {b}: sentinel {int(ids[specials[0]])} in a sample without <image> "
                                 "(the reference would index embed_tokens with it)")
            samples.append(np.stack([np.zeros(ids.shape[0], np.int64), ids], 1))
            cur += 1
            continue
        parts = []
        prev = 0
        for sp in specials.tolist():
            if sp > prev:
                seg = ids[prev:sp]
                parts.append(np.stack([np.zeros(seg.shape[0], np.int64), seg], 1))
            t = int(ids[sp])
            if t == IMAGE_TOKEN_INDEX:
                if cur >= len(feature_blocks):
                    raise ValueError(f"sample {b}: more <image> sentinels than visual features ({len(feature_blocks)})")
                first, n = feature_blocks[cur]
                parts.append(np.stack([np.full(n, KIND_VISUAL, np.int64), np.arange(first, first + n, dtype=np.int64)], 1))
                cur += 1
            elif t == OBJS_TOKEN_INDEX and use_regions:
                rr = region_rows[cur - 1] if cur >= 1 else -1
                if rr < 0:
                    raise ValueError(f"sample {b}: <objs> follows a video frame or no image (no region feature to bind)")
                parts.append(np.array([[KIND_REGION, rr]], dtype=np.int64))
            else:
                raise ValueError(f"sample {b}: sentinel {t} but no regions were passed")
            prev = sp + 1
        if prev < ids.shape[0]:
            seg = ids[prev:]
            parts.append(np.stack([np.zeros(seg.shape[0], np.int64), seg], 1))
        samples.append(np.concatenate(parts, 0) if parts else np.zeros((0, 2), np.int64))
    if max_length is not None:
        samples = [p[:max_length] for p in samples]
    lengths = [int(p.shape[0]) for p in samples]
    S = max(lengths) if lengths else 0
    B = len(samples)
    plan = np.zeros((B, S, 2), dtype=np.int32)
    plan[:, :, 0] = KIND_ZERO
    mask = np.zeros((B, S), dtype=np.int32)
    pos = np.zeros((B, S), dtype=np.int64)
    for b, p in enumerate(samples):
        n = p.shape[0]
        if n == 0:
            continue
        sl = slice(S - n, S) if padding_side == "left" else slice(0, n)
        plan[b, sl] = p
        mask[b, sl] = 1
        pos[b, sl] = np.arange(n)
    return plan, mask, pos, lengths


def build_splice_plan(input_ids, attention_mask, feature_blocks, region_rows, max_length=None, padding_side="right"):
    """List-of-tuples view of build_splice_plan_np: (plan [B][S] of (kind, index), mask [B][S], position_ids [B][S], lengths)."""
    plan, mask, pos, lengths = build_splice_plan_np(input_ids, attention_mask, feature_blocks, region_rows, max_length, padding_side)
    return [[(int(k), int(i)) for k, i in row] for row in plan.tolist()], mask.tolist(), pos.tolist(), lengths


class LlavaMetaModel:
    """Owner of the multimodal sub-modules under the reference's attribute names (llava_arch.py:28-58): they are part
    of the checkpoint format (`model.mm_projector.*`, `model.region_extractor.*`)."""

    image_tower = None
    video_tower = None
    mm_projector = None
    region_extractor = None

    def get_image_tower(self):
        t = getattr(self, "image_tower", None)
        return t[0] if type(t) is list else t

    def get_video_tower(self):
        t = getattr(self, "video_tower", None)
        return t[0] if type(t) is list else t

    def get_region_extractor(self):
        t = getattr(self, "region_extractor", None)
        return t[0] if type(t) is list else t


class LlavaMetaForCausalLM:
    """Mixin with the reference's multimodal methods (llava_arch.py:153-573)."""

    def get_model(self):
        raise NotImplementedError

    def get_image_tower(self):
        return self.get_model().get_image_tower()

    def get_video_tower(self):
        return self.get_model().get_video_tower()

    def get_region_extractor(self):
        return self.get_model().get_region_extractor()

    # ---- reference llava_arch.py:168-181 -----------------------------------------------------------------------
    def encode_images(self, images, regions=None):
        image_features = self.get_model().get_image_tower()(images)              # [B, P, mm_hidden]
        region_features = None
        if regions is not None:
            region_features = self.get_model().get_region_extractor()(image_features, regions)   # [B, 1, hidden]
        image_features = self.get_model().mm_projector(image_features)            # [B, P, hidden]
        if region_features is not None:
            return image_features, region_features
        return image_features, torch.zeros_like(image_features)

    # ---- reference llava_arch.py:183-187 -----------------------------------------------------------------------
    def encode_videos(self, videos):
        video_features = self.get_model().get_video_tower()(videos)               # [B, T, P, mm_hidden]
        return self.get_model().mm_projector(video_features)                      # [B, T, P, hidden]

    def _encode_with_cache(self, images, image_idx, video_idx, regions, cache, plan, blocks):
        """encode_images / encode_videos with a per-item content cache (prefix_cache.VisualFeatureCache): only the items whose
        pixels (+ box) were not seen before run through the towers, as one batch. Returns the same (visual chunks, region rows)
        the uncached branch builds and records the row signature of the spliced sequence."""
        from ..prefix_cache import key64, row_signature, tensor_key
        dev = self.device
        block_keys, region_keys = [], []
        vis_chunks: List[torch.Tensor] = []
        region_buf = None
        self.last_tower_items = 0
        if image_idx:
            keys = [(tensor_key(images[i]), None if regions is None else tuple(float(v) for v in regions[i])) for i in image_idx]
            got = [cache.get(k) for k in keys]
            miss = [j for j, g in enumerate(got) if g is None]
            if miss:
                batch = torch.stack([images[image_idx[j]] for j in miss]).to(dev)
                rb = [regions[image_idx[j]] for j in miss] if regions is not None else None
                feats, regs = self.encode_images(batch, rb)
                for n, j in enumerate(miss):
                    got[j] = (feats[n].contiguous(), regs[n].contiguous() if regions is not None else None)
                    cache.put(keys[j], *got[j])
                self.last_tower_items += len(miss)
            vis_chunks.append(torch.cat([g[0] for g in got], 0))
            if regions is not None:
                region_buf = torch.cat([g[1].reshape(1, -1) for g in got], 0)
            for j, k in enumerate(keys):
                start, cnt = blocks[image_idx[j]][0]
                block_keys.append((start, cnt, key64((k[0], 0))))
                region_keys.append(key64(k))
        if video_idx:
            keys = [(tensor_key(images[i]), None) for i in video_idx]
            got = [cache.get(k) for k in keys]
            miss = [j for j, g in enumerate(got) if g is None]
            if miss:
                batch = torch.stack([images[video_idx[j]] for j in miss]).to(dev)
                feats = self.encode_videos(batch)                                     # [b, T, P, H]
                for n, j in enumerate(miss):
                    got[j] = (feats[n].reshape(-1, feats.shape[-1]).contiguous(), None)
                    cache.put(keys[j], *got[j])
                self.last_tower_items += len(miss)
            vis_chunks.append(torch.cat([g[0] for g in got], 0))
            for j, k in enumerate(keys):
                for t, (start, cnt) in enumerate(blocks[video_idx[j]]):       # one block per frame
                    block_keys.append((start, cnt, key64((k[0], t))))
        block_keys.sort()
        self._last_row_sig = [row_signature(plan[b], block_keys, region_keys) for b in range(plan.shape[0])]
        return vis_chunks, region_buf

    # ---- reference llava_arch.py:189-573 -----------------------------------------------------------------------
    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images, regions=None, feature_cache=None, input_ids_host=None):
        """`feature_cache` (vitron_amd.prefix_cache.VisualFeatureCache, only passed by generate()) lets an image / clip that
        was already encoded in an earlier turn skip tower + region extractor + projector; it also makes this call record
        the per-row signature generate() needs for KV prefix reuse (self._last_row_sig).
        `input_ids_host` (optional, a CPU copy of `input_ids` -- what the tokenizer returned before `.cuda()`): the splice plan is
        integer work on the host, and reading the ids back from the device costs a stream synchronisation right where a
        serving loop wants the next request's launches queued behind the previous one's; with the host copy at hand the call
        never waits for the device."""
        image_tower, video_tower = self.get_image_tower(), self.get_video_tower()
        if (image_tower is None and video_tower is None) or images is None or input_ids.shape[1] == 1:
            # decode step (:196-205): extend the mask to past_len + 1, positions = sum(mask) - 1
            if past_key_values is not None and (image_tower is not None or video_tower is not None) \
                    and images is not None and input_ids.shape[1] == 1:
                target = past_key_values.seq_length() + 1
                attention_mask = torch.cat((attention_mask, torch.ones(
                    (attention_mask.shape[0], target - attention_mask.shape[1]), dtype=attention_mask.dtype,
                    device=attention_mask.device)), dim=1)
                position_ids = torch.sum(attention_mask, dim=1).unsqueeze(-1) - 1
            return input_ids, position_ids, attention_mask, past_key_values, None, labels

        use_regions = regions is not None and len(regions) > 0                    # :233
        images = list(images)                                                     # a batched tensor iterates along dim 0
        image_idx = [i for i, im in enumerate(images) if im.ndim == 3]
        video_idx = [i for i, im in enumerate(images) if im.ndim == 4]
        dev = self.device
        if image_idx and image_tower is None:
            raise ValueError("images were passed but the model has no image tower")
        if video_idx and video_tower is None:
            raise ValueError("videos were passed but the model has no video tower")

        # ---- host side first: the integer plan only needs token ids and feature-block SIZES (known from the tower
        # configs), so it is built and uploaded before any GPU work is queued and never stalls the device ----------------
        blocks: List[Optional[List[Tuple[int, int]]]] = [None] * len(images)
        reg_rows: List[Optional[List[int]]] = [None] * len(images)
        nvis = 0
        if image_idx:
            P = image_tower.num_patches
            for j, i in enumerate(image_idx):
                blocks[i] = [(nvis + j * P, P)]
                reg_rows[i] = [j]
            nvis += len(image_idx) * P
        if video_idx:
            P = video_tower.num_patches
            for j, i in enumerate(video_idx):
                T = images[i].shape[1]
                blocks[i] = [(nvis + (j * T + t) * P, P) for t in range(T)]       # each frame = one "image" (:255-258)
                reg_rows[i] = [-1] * T
                nvis_end = nvis + (j + 1) * T * P
            nvis = nvis_end
        flat_blocks = [blk for bl in blocks for blk in bl]
        flat_regs = [r for rl in reg_rows for r in rl] if use_regions else None
        if input_ids_host is not None:
            if tuple(input_ids_host.shape) != tuple(input_ids.shape):
                raise ValueError("input_ids_host must be a host copy of input_ids (same shape)")
            ids_host = input_ids_host.cpu().numpy()
        else:
            ids_host = input_ids.cpu().numpy()
        am_host = None if attention_mask is None else attention_mask.cpu().numpy()
        plan, mask, pos, lengths = build_splice_plan_np(
            ids_host, am_host, flat_blocks, flat_regs, getattr(self.config, "tokenizer_model_max_length", None),
            getattr(self.config, "tokenizer_padding_side", "right"))
        B, S = plan.shape[0], plan.shape[1]
        tok_rows = plan[..., 1][plan[..., 0] == KIND_TOKEN]
        n_embed = self.get_model().llama.embed_rows
        if tok_rows.size and (int(tok_rows.min()) < 0 or int(tok_rows.max()) >= n_embed):   # nn.Embedding would raise
            bad = int(tok_rows.min()) if int(tok_rows.min()) < 0 else int(tok_rows.max())
            raise IndexError(f"token id {bad} outside the embedding table ({n_embed} rows)")
        plan_t = torch.from_numpy(plan.reshape(B * S, 2)).to(dev, non_blocking=True)

        # ---- device side: towers -> region extractor -> projector -> gather/splice ------------------------------------------
        vis_chunks: List[torch.Tensor] = []
        lo_chunks: List[Optional[torch.Tensor]] = []       # precise level 2: the low halves of the projected visual tokens (operand pairs)
        region_buf = None
        self._last_row_sig = None
        if feature_cache is not None and getattr(self, "precise_level", 0) >= 2:
            feature_cache = None                           # the cache holds 16-bit features: a verification pass encodes afresh
        if feature_cache is not None:
            vis_chunks, region_buf = self._encode_with_cache(images, image_idx, video_idx, regions if use_regions else None,
                                                             feature_cache, plan, blocks)
        else:
            if image_idx:
                batch = torch.stack([images[i] for i in image_idx]).to(dev)
                rb = [regions[i] for i in image_idx] if use_regions else None         # :241
                feats, region_buf = self.encode_images(batch, rb)
                vis_chunks.append(feats.reshape(-1, feats.shape[-1]))
                lo_chunks.append(None if pair_lo(feats) is None else pair_lo(feats).reshape(-1, feats.shape[-1]))
                region_buf = region_buf.reshape(-1, region_buf.shape[-1]) if use_regions else None
            if video_idx:
                batch = torch.stack([images[i] for i in video_idx]).to(dev)
                feats = self.encode_videos(batch)                                     # [b, T, P, H]
                vis_chunks.append(feats.reshape(-1, feats.shape[-1]))
                lo_chunks.append(None if pair_lo(feats) is None else pair_lo(feats).reshape(-1, feats.shape[-1]))
        vis = torch.cat(vis_chunks, 0) if len(vis_chunks) > 1 else vis_chunks[0]
        if vis.shape[0] != nvis:
            raise RuntimeError(f"visual token count mismatch: planned {nvis}, encoded {vis.shape[0]}")
        embeds = ops.embed_splice(self.get_model().embed_tokens_weight, vis.contiguous(), region_buf, plan_t).view(B, S, -1)
        if any(l is not None for l in lo_chunks):
            # the same gather over the LOW halves: token and region rows are exact 16-bit values (low half zero -- a one-row zero table: the
            # splice kernel returns a zero row for every index outside its table), visual rows bring the projector's low half
            vis_lo = torch.cat([l if l is not None else torch.zeros_like(v) for l, v in zip(lo_chunks, vis_chunks)], 0)
            zero_row = torch.zeros((1, embeds.shape[-1]), dtype=embeds.dtype, device=embeds.device)
            zero_reg = None if region_buf is None else torch.zeros_like(region_buf)
            with_lo(embeds, ops.embed_splice(zero_row, vis_lo.contiguous(), zero_reg, plan_t).view(B, S, -1))
        ids_host = ids_host.tolist()
        am_host = None if am_host is None else am_host.tolist()
        mask, pos = mask.tolist(), pos.tolist()
        plan = [[(int(k), int(i)) for k, i in row] for row in plan.tolist()] if labels is not None else None

        new_labels = None
        if labels is not None:  # labels follow the same layout: text keeps its label, visual rows are IGNORE_INDEX
            new_labels = torch.full((B, S), IGNORE_INDEX, dtype=labels.dtype, device=labels.device)
            for b in range(B):
                src = [l for l, m in zip(labels[b].tolist(), am_host[b] if am_host else [1] * labels.shape[1]) if m]
                toks = [t for t, m in zip(ids_host[b], am_host[b] if am_host else [1] * len(ids_host[b])) if m]
                it = iter(l for l, t in zip(src, toks) if t >= 0)
                row = [next(it) if k == KIND_TOKEN else IGNORE_INDEX for (k, _), mm in zip(plan[b], mask[b]) if mm]
                off = S - len(row) if getattr(self.config, "tokenizer_padding_side", "right") == "left" else 0
                new_labels[b, off:off + len(row)] = torch.tensor(row, dtype=labels.dtype)
        out_mask = None if attention_mask is None else torch.tensor(mask, dtype=attention_mask.dtype, device=attention_mask.device)
        out_pos = None if position_ids is None else torch.tensor(pos, dtype=torch.long, device=dev)
        self._last_splice = (mask, pos, lengths)  # host copies for forward() (avoids a device round trip)
        return None, out_pos, out_mask, past_key_values, embeds, new_labels
