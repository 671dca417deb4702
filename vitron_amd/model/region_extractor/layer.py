"""RegionExtractor -- mirror of the reference's vitron/model/region_extractor/layer.py:58-130 on HIP kernels."""
from __future__ import annotations

import torch

from ... import _lib, synth
from ...engine import PackedRegion


class RegionExtractor:
    def __init__(self, in_dim=1024, out_dim=4096, patch_size=14, image_size=224):
        self.in_dim, self.out_dim = in_dim, out_dim
        self.image_size, self.patch_size = image_size, patch_size
        self._sd = None
        self.packed = None
        self._dtype = None

    KEYS = [f"region_linear.layers.{i}.{p}" for i in range(3) for p in ("weight", "bias")] + \
           [f"loc_encoder.loc_encoder.{i}.{p}" for i in (0, 2) for p in ("weight", "bias")]

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.KEYS if k not in sd]
        if missing and strict:
            raise KeyError(f"region_extractor: missing keys {missing}")
        self._sd = {k: sd[k] for k in self.KEYS if k in sd}
        self.packed = None
        return missing, [k for k in sd if k not in self.KEYS]

    def state_dict(self):
        return dict(self._sd or {})

    def to(self, device=None, dtype=None):
        if dtype is not None:
            _lib.operand_of(dtype)
            if self.packed is not None and dtype != self.packed.dtype:
                device = device if device is not None else self.packed.device
                self.packed = None          # the state dict is kept: repack below
            self._dtype = dtype
        if device is not None and self._sd is not None and torch.device(device).type == "cuda" and self.packed is None:
            self.packed = PackedRegion(self._sd, device, self.image_size, self.patch_size, dtype=self._dtype)
        return self

    def init_synthetic(self, gen, device, w_std=0.02, b_std=0.0):
        self.load_state_dict(synth.region_state(self.in_dim, self.out_dim, gen, device, w_std, b_std))
        return self.to(device)

    def __call__(self, feats, regions):
        """feats [B, S, C] patch features, regions: list of B boxes [x1,y1,x2,y2] -> [B, 1, out_dim]."""
        if self.packed is None:
            raise RuntimeError("region_extractor has no packed weights: load a state dict and move it to the GPU first")
        if feats.size(0) != len(regions):   # reference prints and continues (layer.py:101-107)
            print(f"region_extractor: {feats.size(0)} feature maps vs {len(regions)} regions")
        return self.packed.forward(feats, regions)

    forward = __call__
