"""mm_projector factory -- mirror of the reference's vitron/model/multimodal_projector/builder.py:33-51."""
from __future__ import annotations

import re

import torch

from ... import _lib, synth
from ...engine import PackedProjector


class IdentityMap:
    """reference builder.py:6-15."""

    def __call__(self, x, *args, **kwargs):
        return x

    forward = __call__

    @property
    def config(self):
        return {"mm_projector_type": "identity"}


class VisionProjector:
    """Linear / mlpNx_gelu projector running on the MFMA GEMM (bias + exact-erf GELU fused in the epilogue).
    State-dict keys are nn.Sequential's ('0.weight', '0.bias', '2.weight', '2.bias', '4.weight', ...) or nn.Linear's."""

    def __init__(self, mm_hidden_size: int, hidden_size: int, depth: int = 2):
        if depth < 1:
            raise ValueError(f"mlp{depth}x_gelu: the projector needs at least one Linear layer")
        self.mm_hidden_size, self.hidden_size, self.depth = mm_hidden_size, hidden_size, depth
        self._sd = None
        self.packed = None
        self._dtype = None

    def load_state_dict(self, sd, strict=True):
        # nn.Sequential(Linear, GELU, Linear, GELU, Linear, ...): the Linear layers sit at the even indices (builder.py:41-45);
        # 'linear' is a bare nn.Linear ('weight', 'bias') while 'mlp1x_gelu' is nn.Sequential(Linear) ('0.weight', '0.bias')
        keys = [f"{2 * i}.{n}" for i in range(self.depth) for n in ("weight", "bias")]
        if self.depth == 1 and "0.weight" not in sd:
            keys = ["weight", "bias"]
        missing = [k for k in keys if k not in sd]
        if missing and strict:
            raise KeyError(f"mm_projector: missing keys {missing}")
        self._sd = {k: sd[k] for k in keys if k in sd}
        self.packed = None
        return missing, [k for k in sd if k not in keys]

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
            self.packed = PackedProjector(self._sd, device, dtype=self._dtype, depth=self.depth)
        return self

    def init_synthetic(self, gen, device, w_std=0.02, b_std=0.0):
        sd = synth.projector_state(self.mm_hidden_size, self.hidden_size, gen, device, w_std, b_std)
        if self.depth == 1:
            sd = {"weight": sd["0.weight"], "bias": sd["0.bias"]}
        for i in range(2, self.depth):        # mlpNx_gelu, N > 2: further hidden -> hidden layers
            extra = synth.projector_state(self.hidden_size, self.hidden_size, gen, device, w_std, b_std)
            sd[f"{2 * i}.weight"], sd[f"{2 * i}.bias"] = extra["2.weight"], extra["2.bias"]
        self.load_state_dict(sd)
        return self.to(device)

    def __call__(self, x):
        if self.packed is None:
            raise RuntimeError("mm_projector has no packed weights: load a state dict and move it to the GPU first")
        return self.packed.forward(x)

    forward = __call__


def build_vision_projector(config, delay_load=False, **kwargs):
    projector_type = getattr(config, "mm_projector_type", "linear")
    if projector_type == "linear":
        return VisionProjector(config.mm_hidden_size, config.hidden_size, 1)
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if m:
        return VisionProjector(config.mm_hidden_size, config.hidden_size, int(m.group(1)))
    if projector_type == "identity":
        return IdentityMap()
    raise ValueError(f"Unknown projector type: {projector_type}")
