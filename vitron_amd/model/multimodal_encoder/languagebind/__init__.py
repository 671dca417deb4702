"""LanguageBind tower wrappers -- mirror of the reference's
vitron/model/multimodal_encoder/languagebind/__init__.py:69-233 (LanguageBindImageTower / LanguageBindVideoTower).

`forward` = tower -> hidden_states[select_layer] -> drop CLS (feature_select), run as ONE vt_vit_forward call: the
library executes exactly the `n_layers + 1 + select_layer` encoder layers that hidden state needs (23 of 24 for
select_layer = -2) and never materialises the other hidden states or the pooled output the reference discards.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch

from .... import _lib, synth
from ....engine import PackedVit

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)   # reference image/processing_image.py:7
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)   # :8


class VisionConfig(SimpleNamespace):
    """The fields of the reference's CLIPVisionConfig that the hot path reads (configuration_video.py:181-232)."""

    @classmethod
    def from_dict(cls, d):
        base = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                    image_size=224, patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5, add_time_attn=False,
                    num_frames=1, lora_r=0, lora_alpha=16)
        base.update({k: v for k, v in d.items() if k in base})
        return cls(**base)

    def to_dict(self):
        return dict(self.__dict__)


def _load_dir_state(path):
    """(vision_config dict, vision_model.* state dict) from a LanguageBind checkpoint directory."""
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    vcfg = cfg.get("vision_config", cfg)
    sd = None
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
    pref = "vision_model."
    sd = {k[len(pref):]: v for k, v in sd.items() if k.startswith(pref)}
    sd = {k.replace("encoder.base_model.model.", "encoder."): v for k, v in sd.items()}  # peft-wrapped encoder (image tower)
    return vcfg, sd


class _Tower:
    kind = "image"
    select_features = ("patch",)          # LanguageBind towers: feature_select always drops CLS (languagebind/__init__.py:96-104)

    def __init__(self, tower_name, args, delay_load=False, cache_dir="./cache_dir"):
        self.is_loaded = False
        self.tower_name = tower_name
        self.select_layer = getattr(args, "mm_vision_select_layer", -2)
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        if self.select_feature not in self.select_features:
            raise ValueError(f"Unexpected select feature: {self.select_feature}")
        self.cache_dir = cache_dir
        self.cfg_only = None
        self.packed = None
        self._device = torch.device("cpu")
        self._sd = None
        self._dtype = None          # operand dtype (bf16 / fp16); None = the package default until .to(dtype=...) says otherwise
        if os.path.isdir(tower_name):
            with open(os.path.join(tower_name, "config.json")) as f:
                c = json.load(f)
            self.cfg_only = VisionConfig.from_dict(c.get("vision_config", c))
            if not delay_load:
                self.load_model()

    # ---- loading -------------------------------------------------------------------------------------------
    def load_model(self, device=None):
        vcfg, sd = _load_dir_state(self.tower_name)
        self.load_state(VisionConfig.from_dict(vcfg), sd, device)

    def load_state(self, config, state_dict, device=None):
        """Attach weights in the reference's CLIPVisionTransformer naming (peft LoRA adapters are merged)."""
        self.cfg_only = config if isinstance(config, VisionConfig) else VisionConfig.from_dict(dict(config))
        # (an IMAGE tower checkpoint with add_time_attn carries temporal_mlp.* / temporal_layer_norm2.* per layer -- reference
        # image/modeling_image.py:74-84,105-134; PackedVit packs them and vt_vit_forward runs the extra block. LanguageBind_Image ships
        # add_time_attn = false, so this is the uncommon branch.)
        self._sd = state_dict
        self.is_loaded = True
        if device is not None:
            self.to(device)

    def init_synthetic(self, config: dict, gen, device, w_std=0.02, b_std=0.0):
        self.load_state(VisionConfig.from_dict(config), synth.vit_state(config, gen, device, w_std, b_std), device)
        return self

    def to(self, device=None, dtype=None):
        """nn.Module.to as the reference calls it on the towers (`.to(device=device, dtype=torch.float16)`, builder.py:153,161):
        the dtype (bf16 / fp16) is the operand format the weights are packed in the first time they reach the GPU."""
        if dtype is not None:
            _lib.operand_of(dtype)
            if self.packed is not None and dtype != self.packed.dtype:
                raise RuntimeError(f"{type(self).__name__}: weights are already packed as {self.packed.dtype}; reload them to change the dtype")
            self._dtype = dtype
        if device is not None and self._sd is not None and torch.device(device).type == "cuda":
            self.packed = PackedVit(self._sd, self.config.to_dict(), device, self.select_layer, dtype=self._dtype)
            self._device = torch.device(device)
            self._sd = None  # the packed copy is the only one kept on the device
        return self

    def requires_grad_(self, flag=False):
        return self

    # ---- reference surface -------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x):
        if self.packed is None:
            raise RuntimeError(f"{type(self).__name__} is not on the GPU: call load_model()/load_state() and .to('cuda')")
        if type(x) is list:
            return [self.packed.forward(t.unsqueeze(0).to(self.device)).to(t.dtype if t.dtype != torch.float32 else self.dtype) for t in x]
        return self.packed.forward(x.to(self.device))

    __call__ = forward

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return self.packed.dtype if self.packed is not None else (self._dtype or _lib.torch_dtype())

    @property
    def device(self):
        return self._device

    @property
    def config(self):
        return self.cfg_only

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2


class LanguageBindImageTower(_Tower):
    """reference languagebind/__init__.py:69-148. forward(images [B,3,H,W]) -> [B, P, hidden] (CLS dropped)."""
    kind = "image"

    def __init__(self, image_tower, args, delay_load=False, cache_dir="./cache_dir"):
        super().__init__(image_tower, args, delay_load, cache_dir)
        self.image_tower_name = image_tower
        self.image_processor = None


class LanguageBindVideoTower(_Tower):
    """reference languagebind/__init__.py:151-233. forward(videos [B,3,T,H,W]) -> [B, T, P, hidden]."""
    kind = "video"

    def __init__(self, video_tower, args, delay_load=False, cache_dir="./cache_dir"):
        super().__init__(video_tower, args, delay_load, cache_dir)
        self.video_tower_name = video_tower
        self.video_processor = None
