"""Multimodal glue: host-side mirror of the reference's vitron/model/llava_arch.py.

Same public surface (LlavaMetaModel accessors, LlavaMetaForCausalLM.encode_images / encode_videos /
prepare_inputs_labels_for_multimodal) with the tensor work moved to HIP kernels:
  towers -> vt_vit_forward, region_extractor -> vt_region_forward, mm_projector -> vt_projector_forward,
  the list surgery of llava_arch.py:306-398 / :479-558 -> an integer plan built here + vt_embed_splice.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from .. import ops
from ..constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX, OBJS_TOKEN_INDEX

KIND_TOKEN, KIND_VISUAL, KIND_REGION, KIND_ZERO = 0, 1, 2, 3


def build_splice_plan(input_ids: Sequence[Sequence[int]], attention_mask: Optional[Sequence[Sequence[int]]],
                      feature_blocks: Sequence[Tuple[int, int]], region_rows: Optional[Sequence[int]],
                      max_length: Optional[int] = None, padding_side: str = "right"):
    """Integer plan of prepare_inputs_labels_for_multimodal (reference llava_arch.py:300-398 / :474-558).

    input_ids       [B][L] ints with -200 / -300 sentinels; attention_mask [B][L] (None = all ones)
    feature_blocks  one (first_row, n_rows) per flat visual feature (one per image, T per video) in list order
    region_rows     per flat feature: row of its region feature (or -1), None when the caller passed no regions
    Returns (plan [B][S][2] int, mask [B][S] int, position_ids [B][S] int, lengths [B]).
    Quirks kept from the reference: a sample without <image> still consumes one feature slot (:317-324);
    <objs> takes the region of the most recently consumed feature (:350-351); truncate before padding (:363-366).
    """
    use_regions = region_rows is not None
    samples: List[List[Tuple[int, int]]] = []
    cur = 0
    for b, row in enumerate(input_ids):
        ids = [int(t) for j, t in enumerate(row) if attention_mask is None or int(attention_mask[b][j])]
        plan: List[Tuple[int, int]] = []
        if not any(t == IMAGE_TOKEN_INDEX for t in ids):
            for t in ids:
                if t < 0:
                    raise ValueError(f"sample {b}: sentinel {t} in a sample without <image> (the reference would index embed_tokens with it)")
                plan.append((KIND_TOKEN, t))
            samples.append(plan)
            cur += 1
            continue
        for t in ids:
            if t == IMAGE_TOKEN_INDEX:
                if cur >= len(feature_blocks):
                    raise ValueError(f"sample {b}: more <image> sentinels than visual features ({len(feature_blocks)})")
                first, n = feature_blocks[cur]
                plan.extend((KIND_VISUAL, first + r) for r in range(n))
                cur += 1
            elif t == OBJS_TOKEN_INDEX and use_regions:
                rr = region_rows[cur - 1] if cur >= 1 else -1
                if rr < 0:
                    raise ValueError(f"sample {b}: <objs> follows a video frame or no image (no region feature to bind)")
                plan.append((KIND_REGION, rr))
            elif t < 0:
                raise ValueError(f"sample {b}: sentinel {t} but no regions were passed")
            else:
                plan.append((KIND_TOKEN, t))
        samples.append(plan)
    if max_length is not None:
        samples = [p[:max_length] for p in samples]
    lengths = [len(p) for p in samples]
    S = max(lengths) if lengths else 0
    plan_out, mask, pos = [], [], []
    for p in samples:
        n = len(p)
        padn = S - n
        if padding_side == "left":
            plan_out.append([(KIND_ZERO, 0)] * padn + p)
            mask.append([0] * padn + [1] * n)
            pos.append([0] * padn + list(range(n)))
        else:
            plan_out.append(p + [(KIND_ZERO, 0)] * padn)
            mask.append([1] * n + [0] * padn)
            pos.append(list(range(n)) + [0] * padn)
    return plan_out, mask, pos, lengths


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

    # ---- reference llava_arch.py:189-573 -----------------------------------------------------------------------
    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images, regions=None):
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
        vis_chunks: List[torch.Tensor] = []
        blocks: List[Optional[List[Tuple[int, int]]]] = [None] * len(images)
        reg_rows: List[Optional[List[int]]] = [None] * len(images)
        nvis = 0
        region_buf = None
        if image_idx:
            if image_tower is None:
                raise ValueError("images were passed but the model has no image tower")
            batch = torch.stack([images[i] for i in image_idx]).to(dev)
            rb = [regions[i] for i in image_idx] if use_regions else None         # :241
            feats, region_buf = self.encode_images(batch, rb)
            P = feats.shape[1]
            vis_chunks.append(feats.reshape(-1, feats.shape[-1]))
            for j, i in enumerate(image_idx):
                blocks[i] = [(nvis + j * P, P)]
                reg_rows[i] = [j]
            nvis += feats.shape[0] * P
            region_buf = region_buf.reshape(-1, region_buf.shape[-1]) if use_regions else None
        if video_idx:
            if video_tower is None:
                raise ValueError("videos were passed but the model has no video tower")
            batch = torch.stack([images[i] for i in video_idx]).to(dev)
            feats = self.encode_videos(batch)                                     # [b, T, P, H]
            b, T, P, H = feats.shape
            vis_chunks.append(feats.reshape(-1, H))
            for j, i in enumerate(video_idx):
                blocks[i] = [(nvis + (j * T + t) * P, P) for t in range(T)]       # each frame = one "image" (:255-258)
                reg_rows[i] = [-1] * T
            nvis += b * T * P
        flat_blocks = [blk for bl in blocks for blk in bl]
        flat_regs = [r for rl in reg_rows for r in rl] if use_regions else None
        vis = torch.cat(vis_chunks, 0) if len(vis_chunks) > 1 else vis_chunks[0]

        ids_host = input_ids.tolist()
        am_host = None if attention_mask is None else attention_mask.tolist()
        plan, mask, pos, lengths = build_splice_plan(
            ids_host, am_host, flat_blocks, flat_regs, getattr(self.config, "tokenizer_model_max_length", None),
            getattr(self.config, "tokenizer_padding_side", "right"))
        B, S = len(plan), len(plan[0])
        plan_t = torch.tensor(plan, dtype=torch.int32, device=dev).reshape(B * S, 2)
        embeds = ops.embed_splice(self.get_model().embed_tokens_weight, vis.contiguous(), region_buf, plan_t).view(B, S, -1)

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
