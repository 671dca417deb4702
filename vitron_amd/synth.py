"""Synthetic (seeded, random-init) weights in the reference's parameter naming.

No checkpoints exist offline, so benchmarks, smoke tests and parity tests run on random-init
weights of the right architecture: Linear / Conv / embeddings ~ N(0, w_std^2), biases ~ N(0, b_std^2) (0 by
default, as SURVEY.md 8(d) specifies), norm gains 1 (+ optional jitter so tests exercise gamma/beta).
Every tensor is rounded to bf16 once, so an fp32 oracle fed `.float()` copies sees exactly the kernel's weights.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

SD = Dict[str, torch.Tensor]

# canonical full-size configurations (ViT-L/14 towers, Vicuna-7B)
VIT_L14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
               patch_size=14, image_size=224, hidden_act="gelu", layer_norm_eps=1e-5, add_time_attn=False, num_frames=1)
VICUNA_7B = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                 vocab_size=32000, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=4096)


def _n(gen, shape, std, device, dtype=torch.bfloat16):
    if std == 0.0:
        return torch.zeros(shape, device=device, dtype=dtype)
    return (torch.randn(shape, generator=gen, device=device, dtype=torch.float32) * std).to(dtype)


def _gain(gen, n, jitter, device, dtype=torch.bfloat16):
    g = torch.ones(n, device=device, dtype=torch.float32)
    if jitter:
        g = g + torch.randn(n, generator=gen, device=device, dtype=torch.float32) * jitter
    return g.to(dtype)


def make_generator(seed: int, device="cpu") -> torch.Generator:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def vit_state(cfg: dict, gen, device="cpu", w_std=0.02, b_std=0.0, ln_jitter=0.0, attn_std: Optional[float] = None) -> SD:
    """CLIPVisionTransformer parameters (reference modeling_video.py:596-608 + CLIPEncoderLayer :65-84)."""
    D, I, P = cfg["hidden_size"], cfg["intermediate_size"], cfg["patch_size"]
    G = cfg["image_size"] // P
    a_std = w_std if attn_std is None else attn_std
    sd = {
        "embeddings.class_embedding": _n(gen, (D,), w_std, device),
        "embeddings.patch_embedding.weight": _n(gen, (D, 3, P, P), w_std, device),
        "embeddings.position_embedding.weight": _n(gen, (G * G + 1, D), w_std, device),
        "pre_layrnorm.weight": _gain(gen, D, ln_jitter, device),
        "pre_layrnorm.bias": _n(gen, (D,), ln_jitter, device),
        "post_layernorm.weight": _gain(gen, D, ln_jitter, device),
        "post_layernorm.bias": _n(gen, (D,), ln_jitter, device),
    }
    for l in range(cfg["num_hidden_layers"]):
        p = f"encoder.layers.{l}."
        blocks = ["self_attn."]
        if cfg.get("add_time_attn", False):
            blocks.append("temporal_attn.")
            sd[p + "temporal_embedding"] = _n(gen, (1, cfg["num_frames"], D), w_std, device)
            sd[p + "temporal_layer_norm1.weight"] = _gain(gen, D, ln_jitter, device)
            sd[p + "temporal_layer_norm1.bias"] = _n(gen, (D,), ln_jitter, device)
        for blk in blocks:
            for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                std = a_std if proj in ("q_proj", "k_proj") else w_std
                sd[p + blk + proj + ".weight"] = _n(gen, (D, D), std, device)
                sd[p + blk + proj + ".bias"] = _n(gen, (D,), b_std, device)
        for ln in ("layer_norm1", "layer_norm2"):
            sd[p + ln + ".weight"] = _gain(gen, D, ln_jitter, device)
            sd[p + ln + ".bias"] = _n(gen, (D,), ln_jitter, device)
        sd[p + "mlp.fc1.weight"] = _n(gen, (I, D), w_std, device)
        sd[p + "mlp.fc1.bias"] = _n(gen, (I,), b_std, device)
        sd[p + "mlp.fc2.weight"] = _n(gen, (D, I), w_std, device)
        sd[p + "mlp.fc2.bias"] = _n(gen, (D,), b_std, device)
    return sd


def projector_state(mm_hidden: int, hidden: int, gen, device="cpu", w_std=0.02, b_std=0.0) -> SD:
    """mlp2x_gelu: nn.Sequential(Linear, GELU, Linear) (reference multimodal_projector/builder.py:40-46)."""
    return {"0.weight": _n(gen, (hidden, mm_hidden), w_std, device), "0.bias": _n(gen, (hidden,), b_std, device),
            "2.weight": _n(gen, (hidden, hidden), w_std, device), "2.bias": _n(gen, (hidden,), b_std, device)}


def region_state(in_dim: int, out_dim: int, gen, device="cpu", w_std=0.02, b_std=0.0) -> SD:
    """RegionExtractor parameters (reference region_extractor/layer.py:58-75)."""
    dims = [(out_dim, in_dim), (out_dim, out_dim), (out_dim, out_dim)]
    sd = {}
    for i, (o, n) in enumerate(dims):
        sd[f"region_linear.layers.{i}.weight"] = _n(gen, (o, n), w_std, device)
        sd[f"region_linear.layers.{i}.bias"] = _n(gen, (o,), b_std, device)
    sd["loc_encoder.loc_encoder.0.weight"] = _n(gen, (out_dim // 2, 4), w_std, device)
    sd["loc_encoder.loc_encoder.0.bias"] = _n(gen, (out_dim // 2,), b_std, device)
    sd["loc_encoder.loc_encoder.2.weight"] = _n(gen, (out_dim, out_dim // 2), w_std, device)
    sd["loc_encoder.loc_encoder.2.bias"] = _n(gen, (out_dim,), b_std, device)
    return sd


def llama_state(cfg: dict, gen, device="cpu", w_std=0.02, ln_jitter=0.0, attn_std: Optional[float] = None) -> SD:
    """LlamaForCausalLM parameters (transformers naming, as loaded by reference builder.py:57)."""
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    a_std = w_std if attn_std is None else attn_std
    sd = {"model.embed_tokens.weight": _n(gen, (V, H), w_std, device),
          "model.norm.weight": _gain(gen, H, ln_jitter, device),
          "lm_head.weight": _n(gen, (V, H), w_std, device)}
    for l in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{l}."
        sd[p + "input_layernorm.weight"] = _gain(gen, H, ln_jitter, device)
        sd[p + "post_attention_layernorm.weight"] = _gain(gen, H, ln_jitter, device)
        sd[p + "self_attn.q_proj.weight"] = _n(gen, (H, H), a_std, device)
        sd[p + "self_attn.k_proj.weight"] = _n(gen, (H, H), a_std, device)
        sd[p + "self_attn.v_proj.weight"] = _n(gen, (H, H), w_std, device)
        sd[p + "self_attn.o_proj.weight"] = _n(gen, (H, H), w_std, device)
        sd[p + "mlp.gate_proj.weight"] = _n(gen, (I, H), w_std, device)
        sd[p + "mlp.up_proj.weight"] = _n(gen, (I, H), w_std, device)
        sd[p + "mlp.down_proj.weight"] = _n(gen, (H, I), w_std, device)
    return sd


def checksum(sd: SD) -> float:
    """Order-independent fingerprint of a state dict (guards seeded fixtures against RNG drift)."""
    tot = 0.0
    for k in sorted(sd):
        t = sd[k].float()
        tot += float(t.double().abs().sum()) + 1e-3 * float(t.double().sum())
    return tot
