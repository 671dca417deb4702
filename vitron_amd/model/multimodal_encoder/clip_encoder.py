"""CLIPVisionTower -- mirror of the reference's vitron/model/multimodal_encoder/clip_encoder.py:7-78, the plain HF-CLIP image tower
`build_image_tower` picks for `openai*` / `laion*` tower names (reference multimodal_encoder/builder.py:12).

A transformers `CLIPVisionModel` checkpoint is the LanguageBind image tower without the temporal block: the same
CLIPVisionTransformer under the key prefix `vision_model.` (patch conv without bias, class / position embeddings, `pre_layrnorm`,
pre-LN encoder layers, quick_gelu for the OpenAI checkpoints), so it runs on the same kernels (vt_vit_forward) through the same
packer; what differs is the loader mapping and `feature_select`, which here also knows 'cls_patch' (clip_encoder.py:29-37).
There is no network: the tower name must resolve to a local directory in the transformers layout (config.json +
model.safetensors | pytorch_model.bin [+ preprocessor_config.json]), either as given or below `cache_dir`.
"""
from __future__ import annotations

import json
import os

import torch

from .languagebind import VisionConfig, _Tower


def _resolve_dir(name: str, cache_dir: str):
    for cand in (name, os.path.join(cache_dir or ".", name), os.path.join(cache_dir or ".", name.replace("/", "--"))):
        if os.path.isdir(cand):
            return cand
    return None


def _load_clip_dir(path):
    """(vision config dict, CLIPVisionTransformer state dict) of a transformers CLIPVisionModel / CLIPModel directory."""
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    vcfg = dict(cfg.get("vision_config") or cfg)
    vcfg.setdefault("hidden_act", "quick_gelu")      # CLIPVisionConfig's default (the OpenAI checkpoints)
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
    pref = "vision_model."
    if any(k.startswith(pref) for k in sd):          # transformers 4.x naming (CLIPVisionModel / CLIPModel); 5.x drops the prefix
        sd = {k[len(pref):]: v for k, v in sd.items() if k.startswith(pref)}
    sd = {k: v for k, v in sd.items() if k.startswith(("embeddings.", "pre_layrnorm.", "encoder.", "post_layernorm."))}
    if "embeddings.patch_embedding.weight" not in sd:
        raise KeyError(f"{path}: no CLIP vision tower tensors (not a CLIPVisionModel / CLIPModel checkpoint)")
    return vcfg, sd


class CLIPVisionTower(_Tower):
    """reference clip_encoder.py:7-78. forward(images [B,3,H,W] | list of [3,H,W]) -> [B, P(+1), hidden] in the input dtype."""
    kind = "image"
    select_features = ("patch", "cls_patch")

    def __init__(self, vision_tower, args, delay_load=False, cache_dir="./cache_dir"):
        self.vision_tower_name = vision_tower
        self.image_processor = None
        self._dir = _resolve_dir(str(vision_tower), cache_dir)
        super().__init__(self._dir or str(vision_tower), args, delay_load, cache_dir)

    def load_model(self, device=None):
        if self._dir is None:
            raise FileNotFoundError(f"CLIPVisionTower: `{self.vision_tower_name}` is not a local checkpoint directory (no network "
                                    "here: place the transformers checkpoint there or below cache_dir)")
        vcfg, sd = _load_clip_dir(self._dir)
        self.load_state(VisionConfig.from_dict(vcfg), sd, device)
        if os.path.exists(os.path.join(self._dir, "preprocessor_config.json")):
            from transformers import CLIPImageProcessor          # the reference's own processor line (clip_encoder.py:23), on the host
            self.image_processor = CLIPImageProcessor.from_pretrained(self._dir)

    def load_state(self, config, state_dict, device=None):
        sd = {(k[len("vision_model."):] if k.startswith("vision_model.") else k): v for k, v in state_dict.items()}
        super().load_state(config, sd, device)

    @torch.no_grad()
    def forward(self, images):
        if self.packed is None:
            raise RuntimeError("CLIPVisionTower is not on the GPU: call load_model()/load_state() and .to('cuda')")
        if type(images) is list:
            return [self._one(im.unsqueeze(0)) for im in images]
        return self._one(images)

    __call__ = forward

    def _one(self, x):
        dt = x.dtype if x.dtype in (torch.bfloat16, torch.float16, torch.float32) else torch.bfloat16
        if self.select_feature == "patch":
            return self.packed.forward(x.to(self.device)).to(dt)
        _, hidden = self.packed.forward(x.to(self.device), return_hidden=True)       # 'cls_patch': the whole hidden state
        return hidden.to(dt)
