"""load_pretrained_model -- mirror of the reference's vitron/model/builder.py:27-171 (same arguments, same 4-tuple).

Differences that are inherent to this implementation:
  * weights end up packed for the HIP kernels (vitron_amd.engine) in the operand dtype `torch_dtype`: a real checkpoint loads
    as fp16 exactly like the reference (`kwargs['torch_dtype'] = torch.float16`, builder.py:47; towers :153,161) unless the
    caller passes torch_dtype=torch.bfloat16; synthetic models default to bf16 (the benchmark's dtype);
  * load_8bit / load_4bit (bitsandbytes, builder.py:36-45) are rejected: there is no quantised kernel path;
  * device must be a GPU: there is no CPU execution path;
  * `model_path="synthetic"` (or a path that does not exist, with `synthetic_ok=True`) builds random-init weights of
    the configured architecture -- the only thing available offline (checkpoints/ holds only download scripts).
LoRA checkpoints follow the reference's layout (builder.py:53-86): base weights from `model_base`, projector /
region_extractor from `non_lora_trainables.bin`, adapters from `adapter_model.bin|safetensors`, merged here as
W += (lora_alpha / r) * B @ A. A `model_base` WITHOUT 'lora' in the name is the projector-only layout (builder.py:87-103):
language model from `model_base`, config from `model_path`, `mm_projector.bin` laid over it.
"""
from __future__ import annotations

import glob
import json
import os

import torch

from .language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM

from ..constants import (DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_VID_END_TOKEN,
                         DEFAULT_VID_START_TOKEN, DEFAULT_VIDEO_PATCH_TOKEN)


def _load_weight_files(path):
    sd = {}
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if files:
        from safetensors.torch import load_file
        for f in files:
            if "adapter" in os.path.basename(f):
                continue
            sd.update(load_file(f))
        return sd
    for f in sorted(glob.glob(os.path.join(path, "pytorch_model*.bin"))):
        sd.update(torch.load(f, map_location="cpu"))
    return sd


def _merge_llm_lora(base_sd, adapter_sd, lora_alpha, r=None):
    """peft naming: base_model.model.<name>.lora_A.weight / lora_B.weight (optionally '.default')."""
    out = dict(base_sd)
    for k, a in adapter_sd.items():
        if "lora_A" not in k:
            continue
        kb = k.replace("lora_A", "lora_B")
        name = k.split(".lora_A")[0].replace("base_model.model.", "", 1) + ".weight"
        if name not in out or kb not in adapter_sd:
            continue
        b = adapter_sd[kb]
        rr = a.shape[0] if r is None else r
        out[name] = (out[name].float() + (lora_alpha / rr) * (b.float() @ a.float())).to(out[name].dtype)
    return out


def _add_special_tokens(tokenizer, cfg):
    """Special tokens of reference builder.py:136-145; returns len(tokenizer) for resize_token_embeddings (:146), or None when
    the caller's tokenizer object cannot add tokens."""
    if tokenizer is None or not hasattr(tokenizer, "add_tokens"):
        return None
    if getattr(cfg, "mm_use_im_patch_token", True):
        tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
        tokenizer.add_tokens([DEFAULT_VIDEO_PATCH_TOKEN], special_tokens=True)
    if getattr(cfg, "mm_use_im_start_end", False):
        tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
        tokenizer.add_tokens([DEFAULT_VID_START_TOKEN, DEFAULT_VID_END_TOKEN], special_tokens=True)
    return len(tokenizer)


def _processors(model):
    """{'image': processor, 'video': processor} like the reference's builder (builder.py:149-171): device-side processors
    (vitron_amd.processing) exposing what app.py / mm_utils read -- preprocess(...)['pixel_values'], crop_size, image_mean."""
    from ..processing import LanguageBindImageProcessor, LanguageBindVideoProcessor
    proc = {"image": None, "video": None}
    it, vt = model.get_image_tower(), model.get_video_tower()
    if it is not None and it.config is not None:
        it.image_processor = it.image_processor or LanguageBindImageProcessor(image_size=it.config.image_size, device=model.device, dtype=model.dtype)
        proc["image"] = it.image_processor
    if vt is not None and vt.config is not None:
        vt.video_processor = vt.video_processor or LanguageBindVideoProcessor(image_size=vt.config.image_size,
                                                                              num_frames=vt.config.num_frames, device=model.device,
                                                                              dtype=model.dtype)
        proc["video"] = vt.video_processor
    return proc


def load_pretrained_model(model_path, model_base, model_name, load_8bit=False, load_4bit=False, device_map="auto",
                          device="cuda", **kwargs):
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes 8/4-bit loading has no MI355X kernel path in vitron_amd")
    if torch.device(device).type != "cuda":
        raise RuntimeError("vitron_amd runs on the GPU only (device must be 'cuda'); there is no CPU path")
    synthetic = kwargs.pop("synthetic", None)
    tokenizer = kwargs.pop("tokenizer", None)
    torch_dtype = kwargs.pop("torch_dtype", None)
    if model_path == "synthetic" or synthetic is not None:
        spec = synthetic or {}
        cfg = LlavaConfig(**spec.get("llm", {}), mm_hidden_size=spec.get("image", spec.get("video", {})).get("hidden_size", 1024))
        model = LlavaLlamaForCausalLM(cfg)
        model.init_synthetic(device, seed=spec.get("seed", 1234), vit_image=spec.get("image"), vit_video=spec.get("video"),
                             w_std=spec.get("w_std", 0.02), resize_for=lambda: _add_special_tokens(tokenizer, cfg),
                             dtype=torch_dtype if torch_dtype is not None else spec.get("dtype"))
        context_len = getattr(cfg, "max_sequence_length", 2048)
        return tokenizer, model, _processors(model), context_len

    with open(os.path.join(model_path, "config.json")) as f:
        cfg = LlavaConfig(**json.load(f))
    if "lora" in model_name.lower() and model_base is None:
        raise ValueError("LoRA checkpoints need `model_base` (reference builder.py:51-52)")
    if tokenizer is None:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(model_base or model_path, use_fast=False)
    model = LlavaLlamaForCausalLM(cfg)
    if "lora" in model_name.lower() and model_base is not None:
        sd = _load_weight_files(model_base)
        nl = os.path.join(model_path, "non_lora_trainables.bin")
        if os.path.exists(nl):
            extra = torch.load(nl, map_location="cpu")
            extra = {(k[11:] if k.startswith("base_model.") else k): v for k, v in extra.items()}
            if any(k.startswith("model.model.") for k in extra):
                extra = {(k[6:] if k.startswith("model.") else k): v for k, v in extra.items()}
            sd.update(extra)
        ad = {}
        for f in ("adapter_model.safetensors", "adapter_model.bin"):
            p = os.path.join(model_path, f)
            if os.path.exists(p):
                if f.endswith(".bin"):
                    ad = torch.load(p, map_location="cpu")
                else:
                    from safetensors.torch import load_file
                    ad = load_file(p)
                break
        acfg = {}
        if os.path.exists(os.path.join(model_path, "adapter_config.json")):
            with open(os.path.join(model_path, "adapter_config.json")) as f:
                acfg = json.load(f)
        sd = _merge_llm_lora(sd, ad, float(acfg.get("lora_alpha", 16)), acfg.get("r"))
    elif model_base is not None:
        # "this may be mm projector only" (reference builder.py:87-103): the language model comes from `model_base` under the CONFIG of
        # `model_path`, and model_path/mm_projector.bin (keys model.mm_projector.*, possibly model.region_extractor.*) is laid over it
        # non-strictly -- what a projector-pretraining stage leaves behind
        sd = _load_weight_files(model_base)
        mp = os.path.join(model_path, "mm_projector.bin")
        if not os.path.exists(mp):
            raise FileNotFoundError(f"{mp}: `model_base` given without 'lora' in model_name means a projector-only checkpoint "
                                    "(reference builder.py:100: torch.load(os.path.join(model_path, 'mm_projector.bin')))")
        extra = torch.load(mp, map_location="cpu")
        sd.update({k: v.to(torch.float16) for k, v in extra.items()})          # :101 casts the projector weights to fp16
    else:
        sd = _load_weight_files(model_path)
    # a LoRA adapter's config may name more vocabulary rows than the base checkpoint holds: the reference re-creates lm_head /
    # embed_tokens at config size (builder.py:57-61, uninitialised rows that non_lora_trainables.bin may fill) -- zero rows here
    for key in ("model.embed_tokens.weight", "lm_head.weight"):
        if key in sd and sd[key].shape[0] < cfg.vocab_size:
            w = sd[key]
            sd[key] = torch.cat([w, torch.zeros((cfg.vocab_size - w.shape[0], w.shape[1]), dtype=w.dtype)], 0)
    model.load_state_dict(sd, strict=False)
    n_tok = _add_special_tokens(tokenizer, cfg)
    if n_tok is not None:
        model.resize_token_embeddings(n_tok)
    # towers (reference builder.py:149-163): LanguageBind checkpoint directories named by the config
    for get in (model.get_image_tower, model.get_video_tower):
        t = get()
        if t is not None and not t.is_loaded:
            t.load_model()
    # reference builder.py:47 loads the language model with torch_dtype=float16 and moves the towers with
    # .to(device=device, dtype=torch.float16) (:153,161): fp16 is the default for a real checkpoint here too
    model.to(device, dtype=torch_dtype if torch_dtype is not None else torch.float16)
    context_len = getattr(cfg, "max_sequence_length", 2048)   # reference builder.py:166-169
    return tokenizer, model, _processors(model), context_len
