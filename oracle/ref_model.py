"""Assemble the REFERENCE's own model objects (through oracle/ref_shim.py) from seeded synthetic state dicts.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (like everything under oracle/): used by tests/golden/make_golden*.py to generate the
reference's outputs in the build container and by bench.py's `cpu_baseline` leg (kind "reference") to time the reference's CPU path
on the GPU box's host cores. The product (vitron_amd/) never imports this module.

What is built is the reference's code, unmodified: `LlavaLlamaForCausalLM` (vitron/model/language_model/llava_llama.py:40) in fp32
with eager attention, its LanguageBind tower wrappers (multimodal_encoder/languagebind/__init__.py:69-233) around the reference's
`CLIPVisionTransformer` (video/modeling_video.py:596, image/modeling_image.py), `build_vision_projector`
(multimodal_projector/builder.py:33) and `RegionExtractor` (region_extractor/layer.py:58). No checkpoints exist offline, so the
towers are attached directly instead of going through `load_model`.
"""
from __future__ import annotations

import contextlib
import io
import sys
import types

import torch
import torch.nn as nn


def f32(sd):
    return {k: v.float() for k, v in sd.items()}


def build_vit(ns, cfg: dict, sd, image_file: bool = False):
    """The reference's CLIPVisionTransformer for a tower config (video file when add_time_attn, image file otherwise; image_file=True
    forces the image file's class -- its add_time_attn variant carries a temporal MLP per layer, image/modeling_image.py:83-84)."""
    if cfg["add_time_attn"] and not image_file:
        cv, mv = ns.configuration_video, ns.modeling_video
    else:
        cv, mv = ns.configuration_image, ns.modeling_image
    c = cv.CLIPVisionConfig(hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                            num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
                            image_size=cfg["image_size"], patch_size=cfg["patch_size"], hidden_act=cfg["hidden_act"],
                            layer_norm_eps=cfg["layer_norm_eps"], add_time_attn=cfg["add_time_attn"],
                            num_frames=cfg["num_frames"])
    m = mv.CLIPVisionTransformer(c).eval()
    missing, unexpected = m.load_state_dict(f32(sd), strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in k for k in missing), missing
    return m


def build_decoder(ns, c: dict, state, max_positions=8192):
    """The reference's LlavaLlamaForCausalLM (no towers yet), fp32, eager attention; `state` maps parameter names to tensors (any
    dtype / device; consumed entry by entry so that a 7B state dict never exists twice in host memory)."""
    ll = ns.llava_llama
    cfg = ll.LlavaConfig(hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"], num_hidden_layers=c["num_hidden_layers"],
                         num_attention_heads=c["num_attention_heads"], num_key_value_heads=c["num_attention_heads"],
                         vocab_size=c["vocab_size"], rms_norm_eps=c["rms_norm_eps"], max_position_embeddings=max_positions,
                         rope_theta=c["rope_theta"], tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    cfg.pretraining_tp = 1
    try:
        from transformers.modeling_utils import no_init_weights
        init_ctx = no_init_weights()
    except Exception:  # noqa: BLE001
        init_ctx = contextlib.nullcontext()
    with contextlib.redirect_stdout(io.StringIO()), init_ctx:     # the reference prints the whole config in __init__
        model = ll.LlavaLlamaForCausalLM(cfg).eval()
    params = dict(model.named_parameters())
    with torch.no_grad():
        for k in list(state):
            params[k].copy_(state.pop(k).to("cpu", torch.float32))
    model.config.tokenizer_model_max_length = None
    model.config.tokenizer_padding_side = "right"
    return model


def attach(ns, model, vcfg: dict, vsd, psd=None, rsd=None, mm_hidden=1024, hidden=4096):
    """Attach a tower of config `vcfg` (video tower when add_time_attn, else image tower; the other slot is cleared), and -- when
    given -- the mlp2x_gelu projector and the RegionExtractor (reference default: 224 canvas, region_extractor/builder.py:5)."""
    lb = sys.modules["vitron.model.multimodal_encoder.languagebind"]
    video = bool(vcfg["add_time_attn"])
    cls, attr = (lb.LanguageBindVideoTower, "video_tower") if video else (lb.LanguageBindImageTower, "image_tower")
    t = cls.__new__(cls)
    nn.Module.__init__(t)
    t.is_loaded, t.select_layer, t.select_feature = True, -2, "patch"
    setattr(t, attr, build_vit(ns, vcfg, vsd))
    model.model.video_tower = t if video else None
    model.model.image_tower = None if video else t
    if psd is not None:
        pcfg = types.SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=mm_hidden, hidden_size=hidden)
        model.model.mm_projector = ns.projector_builder.build_vision_projector(pcfg).eval()
        model.model.mm_projector.load_state_dict(f32(psd))
    if rsd is not None:
        model.model.region_extractor = ns.region_layer.RegionExtractor(mm_hidden, hidden).eval()
        model.model.region_extractor.load_state_dict(f32(rsd))
    return model


def prefill(model, ids, pixels, regions=None):
    """The reference's prefill exactly as forward() runs it (llava_llama.py:73-102): prepare_inputs_labels_for_multimodal, then the
    decoder over ALL positions. Returns (logits [1, S, V], embeds [1, S, H], seconds of the glue + towers, seconds of the decoder)."""
    import time
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        ta = time.perf_counter()
        (_, _, _, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [pixels], regions)
        tb = time.perf_counter()
        logits = model(inputs_embeds=embeds, use_cache=False).logits
        tc = time.perf_counter()
    return logits, embeds, tb - ta, tc - tb
