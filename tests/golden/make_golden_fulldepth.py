"""Full-depth golden of the BENCHMARK workload from the REFERENCE's own code, and the reference's CPU time for it.

    python tests/golden/make_golden_fulldepth.py        # build container only: needs /root/reference, ~45 GB of RAM, ~15 min

Runs the reference's LlavaLlamaForCausalLM (32-layer Vicuna-7B shape, fp32, eager attention) with its LanguageBind video
tower (ViT-L/14, 336 px, 24 layers, temporal attention over 8 frames) and mlp2x_gelu projector attached -- through the
reference's own prepare_inputs_labels_for_multimodal + forward -- on BASELINE configs[2]: one 8-frame 336 x 336 clip + a
512-token prompt (S = 5120). Weights come from vitron_amd.synth.HashGenerator, a counter-based stream that is bit-identical on
CPU and GPU, so tests/test_gpu_parity_fulldepth.py can rebuild the very same 7B weights on the GPU box in seconds.

Writes
  tests/golden/fulldepth_c3.npz   last-position logits, top-5 ids of every position, projections / sample rows of the logits,
                                  of the final hidden state and of the spliced input embeddings (visual tokens included)
  profiles/r2_cpu_reference.json  wall-clock of the reference on this container's host cores (the measured CPU baseline
                                  bench.py's cpu_baseline cites next to its sampled estimate)
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from tests.golden import cases, make_golden  # noqa: E402
from vitron_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 20260926
FRAMES, IMAGE, TEXT = 8, 336, 512


def c3_inputs():
    """The clip and the prompt ids of the case (CPU generators: cheap to regenerate anywhere)."""
    clip = cases.pixels((3, FRAMES, IMAGE, IMAGE), SEED + 1)
    g = torch.Generator().manual_seed(SEED + 2)
    text = torch.randint(3, 32000, (TEXT - 1,), generator=g)
    ids = torch.cat([torch.tensor([1]), torch.full((FRAMES,), -200), text]).unsqueeze(0)
    return clip, ids


def c3_weights(device="cpu"):
    """(llama state dict, video tower state dict, projector state dict), bf16, from the device-independent hash streams."""
    vcfg = dict(synth.VIT_L14, image_size=IMAGE, add_time_attn=True, num_frames=FRAMES)
    lsd = synth.llama_state(synth.VICUNA_7B, synth.HashGenerator(SEED + 10), device)
    vsd = synth.vit_state(vcfg, synth.HashGenerator(SEED + 11), device)
    psd = synth.projector_state(1024, 4096, synth.HashGenerator(SEED + 12), device)
    return lsd, vsd, psd, vcfg


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    ns = ref_shim.install()
    ll, lb = ns.llava_llama, sys.modules["vitron.model.multimodal_encoder.languagebind"]
    t0 = time.time()
    lsd, vsd, psd, vcfg = c3_weights()
    print(f"weights: {time.time() - t0:.0f}s", flush=True)
    c = synth.VICUNA_7B
    cfg = ll.LlavaConfig(hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"], num_hidden_layers=c["num_hidden_layers"],
                         num_attention_heads=c["num_attention_heads"], num_key_value_heads=c["num_attention_heads"],
                         vocab_size=c["vocab_size"], rms_norm_eps=c["rms_norm_eps"], max_position_embeddings=8192,
                         rope_theta=c["rope_theta"], tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    cfg.pretraining_tp = 1
    try:
        from transformers.modeling_utils import no_init_weights
        init_ctx = no_init_weights()
    except Exception:  # noqa: BLE001
        init_ctx = contextlib.nullcontext()
    with contextlib.redirect_stdout(io.StringIO()), init_ctx:
        model = ll.LlavaLlamaForCausalLM(cfg).eval()
    params = dict(model.named_parameters())
    with torch.no_grad():
        for k in list(lsd):
            params[k].copy_(lsd.pop(k).float())
    print(f"decoder built: {time.time() - t0:.0f}s", flush=True)
    t = lb.LanguageBindVideoTower.__new__(lb.LanguageBindVideoTower)
    nn.Module.__init__(t)
    t.is_loaded, t.select_layer, t.select_feature = True, -2, "patch"
    t.video_tower = make_golden.build_ref_vit(ns, vcfg, vsd)
    model.model.video_tower = t
    pcfg = types.SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=1024, hidden_size=4096)
    model.model.mm_projector = ns.projector_builder.build_vision_projector(pcfg).eval()
    model.model.mm_projector.load_state_dict(make_golden.f32(psd))
    model.config.tokenizer_model_max_length = None
    model.config.tokenizer_padding_side = "right"
    clip, ids = c3_inputs()
    grabbed = {}
    hook = model.model.layers[-1].register_forward_hook(
        lambda mod, args, res: grabbed.__setitem__("h", (res[0] if isinstance(res, tuple) else res).detach()))
    print(f"model ready: {time.time() - t0:.0f}s; running the reference on C3 (S = {FRAMES * 576 + TEXT}) ...", flush=True)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        ta = time.perf_counter()
        (_, pos, mask, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [clip], None)
        tb = time.perf_counter()
        logits = model(inputs_embeds=embeds, use_cache=False).logits
        tc = time.perf_counter()
    hook.remove()
    S = embeds.shape[1]
    assert S == FRAMES * 576 + TEXT
    print(f"reference: towers+projector+splice {tb - ta:.1f}s, decoder {tc - tb:.1f}s", flush=True)
    out = {}
    lg = logits[0].float()
    out["last_logits"] = lg[-1].numpy()
    make_golden._compact(lg, "logits", out, top5=True, nrows=4)
    make_golden._compact(grabbed["h"].reshape(S, -1).float(), "hidden", out)
    make_golden._compact(embeds[0].float(), "embeds", out)
    out["S"] = np.int64(S)
    np.savez_compressed(os.path.join(OUT, "fulldepth_c3.npz"), **out)
    total = tc - ta
    rep = {"what": "the REFERENCE's own modules (oracle/ref_shim.py import shim, nothing under /root/reference modified): "
                   "LanguageBind video tower (24 layers; hidden_states[-2] selected) + mlp2x_gelu projector + "
                   "prepare_inputs_labels_for_multimodal + LlavaLlamaForCausalLM.forward (32 layers, logits of all positions), fp32, "
                   "eager attention, full depth, BASELINE configs[2] (8-frame 336x336 clip + 512-token prompt, S=5120), run once",
           "where": "build container (no GPU): host CPU of this container, not of the GPU box",
           "cores": os.cpu_count(), "torch_threads": torch.get_num_threads(),
           "seconds_towers_projector_splice": tb - ta, "seconds_decoder": tc - tb, "seconds_total": total,
           "tokens_per_s": S / total, "kind": "reference", "script": "tests/golden/make_golden_fulldepth.py"}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r2_cpu_reference.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep), flush=True)
    print("fulldepth_c3.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
