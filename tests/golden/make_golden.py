"""Generate tests/golden/*.npz from the REFERENCE's own modules (run in the build container only).

    python tests/golden/make_golden.py            # needs /root/reference (read-only) and oracle/ref_shim.py

The reference ships no golden vectors for this path, so these fixtures are what pins the oracle:
each file holds the OUTPUTS the reference's code produced for seeded synthetic weights/inputs. The weights are
not stored: they are regenerated from the seed by vitron_amd.synth (a checksum of the regenerated state dict is
stored and verified by the tests). All tensors fp32, CPU, eager attention.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from vitron_amd import synth  # noqa: E402
from tests.golden import cases  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF = getattr(ref_shim, "REF", "/root/reference")


def f32(sd):
    return {k: v.float() for k, v in sd.items()}


def build_ref_vit(ns, cfg, sd):
    # video tower: video/modeling_video.py; image tower: image/modeling_image.py (no 'b t n c' view of hidden states; its add_time_attn
    # variant -- cfg["temporal_mlp"] -- has a temporal MLP behind the temporal attention)
    from oracle import ref_model
    if cfg["add_time_attn"] and cfg.get("temporal_mlp", False):
        return ref_model.build_vit(ns, cfg, sd, image_file=True)
    return ref_model.build_vit(ns, cfg, sd)


def gen_vit(ns):
    out = {}
    for name, cfg, shape in (("video", cases.VIT_VIDEO, (2, 3, cases.VIT_VIDEO["num_frames"], 56, 56)),
                             ("image", cases.VIT_IMAGE, (3, 3, 56, 56))):
        sd = synth.vit_state(cfg, synth.make_generator(cases.SEED_VIT), **cases.VIT_INIT)
        m = build_ref_vit(ns, cfg, sd)
        x = cases.pixels(shape, cases.SEED_PIX)
        with torch.no_grad():
            o = m(x, output_hidden_states=True)
        hs = o.hidden_states
        out[f"{name}_checksum"] = np.float64(synth.checksum(sd))
        for i, h in enumerate(hs):
            out[f"{name}_hidden_{i}"] = h.reshape(-1, h.shape[-2], h.shape[-1]).numpy()
    np.savez_compressed(os.path.join(OUT, "vit.npz"), **out)
    print("vit.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


def gen_vit_b(ns):
    out = {}
    for name, (cfg, shape) in cases.VIT_B_CASES.items():
        sd = synth.vit_state(cfg, synth.make_generator(cases.SEED_VIT + 1), **cases.VIT_INIT)
        m = build_ref_vit(ns, cfg, sd)
        x = cases.pixels(shape, cases.SEED_PIX + 9)
        with torch.no_grad():
            hs = m(x, output_hidden_states=True).hidden_states
        out[f"{name}_checksum"] = np.float64(synth.checksum(sd))
        for i, h in enumerate(hs):
            out[f"{name}_hidden_{i}"] = h.reshape(-1, h.shape[-2], h.shape[-1]).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "vit_b.npz"), **out)
    print("vit_b.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


def gen_vit_tmlp(ns):
    """The IMAGE tower file's CLIPVisionTransformer with add_time_attn = True (temporal attention + temporal MLP per layer)."""
    out = {}
    for name, (cfg, shape) in cases.VIT_TMLP_CASES.items():
        sd = synth.vit_state(cfg, synth.make_generator(cases.SEED_VIT + 2), **cases.VIT_INIT)
        m = build_ref_vit(ns, cfg, sd)
        assert hasattr(m.encoder.layers[0], "temporal_mlp")
        x = cases.pixels(shape, cases.SEED_PIX + 2)
        with torch.no_grad():
            hs = m(x, output_hidden_states=True).hidden_states
        out[f"{name}_checksum"] = np.float64(synth.checksum(sd))
        for i, h in enumerate(hs):
            out[f"{name}_hidden_{i}"] = h.reshape(-1, h.shape[-2], h.shape[-1]).numpy()
    np.savez_compressed(os.path.join(OUT, "vit_tmlp.npz"), **out)
    print("vit_tmlp.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


def gen_region_projector(ns):
    out = {}
    # region extractor at the reference-native geometry (224 canvas, 16x16 grid) and on a 4x4 grid
    for tag, (cin, cout, G) in cases.REGION_CASES.items():
        sd = synth.region_state(cin, cout, synth.make_generator(cases.SEED_REGION), **cases.MLP_INIT)
        m = ns.region_layer.RegionExtractor(cin, cout).eval()
        m.load_state_dict(f32(sd))
        feats = cases.features((len(cases.BOXES), G * G, cin), cases.SEED_FEATS)
        with torch.no_grad():
            r = m(feats, cases.BOXES)
            # the integer side: bbox -> canvas -> bilinear -> >0  (layer.py:27-35,77-85)
            canvas = m.transform_bbox_2_mask(cases.BOXES, m.image_size, feats.device, feats.dtype).unsqueeze(1)
            grid = torch.nn.functional.interpolate(canvas, size=(G, G), mode="bilinear", align_corners=False)
        out[f"region_{tag}_checksum"] = np.float64(synth.checksum(sd))
        out[f"region_{tag}_out"] = r.numpy()
        out[f"region_{tag}_cells"] = (grid > 0).reshape(len(cases.BOXES), -1).numpy().astype(np.int32)
    sd = synth.projector_state(cases.MM_HIDDEN, cases.LLM["hidden_size"], synth.make_generator(cases.SEED_PROJ), **cases.MLP_INIT)
    cfg = types.SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=cases.MM_HIDDEN, hidden_size=cases.LLM["hidden_size"])
    pm = ns.projector_builder.build_vision_projector(cfg).eval()
    pm.load_state_dict(f32(sd))
    x = cases.features((37, cases.MM_HIDDEN), cases.SEED_FEATS + 1)
    with torch.no_grad():
        out["projector_out"] = pm(x).numpy()
    out["projector_checksum"] = np.float64(synth.checksum(sd))
    np.savez_compressed(os.path.join(OUT, "region_projector.npz"), **out)
    print("region_projector.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


class _StubTok:  # the stub tokenizer of SURVEY.md Appendix B: ids = [BOS] + [100 + ord(c)]
    bos_token_id = 1

    def __call__(self, text):
        return types.SimpleNamespace(input_ids=[1] + [100 + ord(c) for c in text])


def gen_mm_utils(ns):
    mu = ns.mm_utils
    tok = _StubTok()
    out = {}
    for i, p in enumerate(cases.PROMPTS):
        out[f"image_token_{i}"] = np.array(mu.tokenizer_image_token(p, tok), dtype=np.int64)
        out[f"region_token_{i}"] = np.array(mu.tokenizer_image_region_token(p, tok), dtype=np.int64)
    for i, (r, isz, tsz) in enumerate(cases.REGION_RESCALE):
        out[f"preprocess_region_{i}"] = np.array(mu.preprocess_region(r, isz, tsz), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "mm_utils.npz"), **out)
    print("mm_utils.npz", {k: v.tolist() for k, v in out.items()})


def gen_output_parser():
    """The reference's reply parser lives in app.py next to the Gradio UI; importing app.py would start loading models, so its
    five small functions are compiled straight from the source text (ast), nothing else of the file runs."""
    import ast
    import json
    import re
    src = open(os.path.join(REF, "app.py")).read()
    want = {"find_module_content", "find_instruction_content", "find_region_instrction_content", "remove_special_tags", "parse_model_output"}
    tree = ast.parse(src)
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert {f.name for f in fns} == want
    scope = {"re": re}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "app.py", "exec"), scope)
    out = [{"text": t, "parsed": list(scope["parse_model_output"](t))} for t in cases.MODEL_OUTPUTS]
    with open(os.path.join(OUT, "output_parser.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("output_parser.json", out)


def build_ref_llava(ns):
    """The reference's LlavaLlamaForCausalLM with tiny towers attached."""
    ll, lb = ns.llava_llama, sys.modules["vitron.model.multimodal_encoder.languagebind"]
    L = cases.LLM
    cfg = ll.LlavaConfig(hidden_size=L["hidden_size"], intermediate_size=L["intermediate_size"],
                         num_hidden_layers=L["num_hidden_layers"], num_attention_heads=L["num_attention_heads"],
                         num_key_value_heads=L["num_attention_heads"], vocab_size=L["vocab_size"],
                         rms_norm_eps=L["rms_norm_eps"], max_position_embeddings=L["max_position_embeddings"],
                         rope_theta=L["rope_theta"], tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    cfg.pretraining_tp = 1
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):  # the reference prints the whole config in __init__
        model = ll.LlavaLlamaForCausalLM(cfg).eval()
    lsd = synth.llama_state(L, synth.make_generator(cases.SEED_LLM), **cases.LLM_INIT)
    missing, unexpected = model.load_state_dict(f32(lsd), strict=False)
    assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)

    def tower(cls, attr, vit_cfg):
        sd = synth.vit_state(vit_cfg, synth.make_generator(cases.SEED_VIT), **cases.VIT_INIT)
        t = cls.__new__(cls)
        nn.Module.__init__(t)
        t.is_loaded = True
        t.select_layer = -2
        t.select_feature = "patch"
        setattr(t, attr, build_ref_vit(ns, vit_cfg, sd))
        return t

    model.model.image_tower = tower(lb.LanguageBindImageTower, "image_tower", cases.VIT_IMAGE)
    model.model.video_tower = tower(lb.LanguageBindVideoTower, "video_tower", cases.VIT_VIDEO)
    pcfg = types.SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=cases.MM_HIDDEN, hidden_size=L["hidden_size"])
    model.model.mm_projector = ns.projector_builder.build_vision_projector(pcfg).eval()
    model.model.mm_projector.load_state_dict(f32(synth.projector_state(cases.MM_HIDDEN, L["hidden_size"], synth.make_generator(cases.SEED_PROJ), **cases.MLP_INIT)))
    # the reference builds RegionExtractor(mm_hidden, hidden) with its default 224 canvas (region_extractor/builder.py:5)
    model.model.region_extractor = ns.region_layer.RegionExtractor(cases.MM_HIDDEN, L["hidden_size"]).eval()
    model.model.region_extractor.load_state_dict(f32(synth.region_state(cases.MM_HIDDEN, L["hidden_size"], synth.make_generator(cases.SEED_REGION), **cases.MLP_INIT)))
    return model


def gen_glue(ns):
    import io
    import contextlib
    model = build_ref_llava(ns)
    out = {}
    for name, case in cases.glue_cases().items():
        model.config.tokenizer_model_max_length = case.get("max_length")
        model.config.tokenizer_padding_side = case.get("padding_side", "right")
        ids, am = case["input_ids"], case["attention_mask"]
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            (_, pos, mask, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(
                ids, None, am, None, None, case["images"], case["regions"])
            logits = model(input_ids=ids, attention_mask=am, images=case["images"], regions=case["regions"]).logits
        out[f"{name}_embeds"] = embeds.numpy()
        out[f"{name}_logits"] = logits.float().numpy()
        out[f"{name}_mask"] = (mask if mask is not None else torch.ones(embeds.shape[:2])).numpy().astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "glue_llm.npz"), **out)
    print("glue_llm.npz", {k: v.shape for k, v in out.items()})


def gen_glue_random(ns):
    import contextlib
    import io
    model = build_ref_llava(ns)
    proj = cases.glue_projection(cases.LLM["hidden_size"])
    out = {}
    for name, case in cases.random_glue_cases().items():
        model.config.tokenizer_model_max_length = case.get("max_length")
        model.config.tokenizer_padding_side = case.get("padding_side", "right")
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            (_, pos, mask, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(
                case["input_ids"], None, case["attention_mask"], None, None, case["images"], case["regions"])
        out[f"{name}_proj"] = (embeds.double() @ proj).numpy()
        out[f"{name}_mask"] = (mask if mask is not None else torch.ones(embeds.shape[:2])).numpy().astype(np.int32)
        if pos is not None:
            out[f"{name}_pos"] = pos.numpy().astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "glue_random.npz"), **out)
    print("glue_random.npz", {k: v.shape for k, v in out.items() if k.endswith("_mask")})


def gen_state_dict_keys(ns):
    """Parameter names and shapes of the reference's OWN LlavaLlamaForCausalLM.state_dict() (tiny config, towers / projector /
    region extractor attached as the reference attaches them) and of its tower modules: what a checkpoint written by the
    reference's save path contains. The loader test writes its synthetic checkpoint with THESE names, so a naming mistake cannot
    be shared by the test's writer and the loader under test. Also renders the conversation prompts of the acceptance test."""
    import importlib
    import json
    model = build_ref_llava(ns)
    keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    conv = importlib.import_module("vitron.conversation")
    prompts = {}
    for name, inp in cases.ACCEPTANCE_USER_TURNS.items():
        c = conv.conv_templates["llava_v1"].copy()
        c.append_message(c.roles[0], inp)
        c.append_message(c.roles[1], None)
        prompts[name] = {"prompt": c.get_prompt(), "stop_str": c.sep if c.sep_style != conv.SeparatorStyle.TWO else c.sep2}
    with open(os.path.join(OUT, "ref_state_dict_keys.json"), "w") as f:
        json.dump({"llava_state_dict": keys, "acceptance_prompts": prompts}, f, indent=1)
    print("ref_state_dict_keys.json", len(keys), "keys;", {k: v["prompt"][-60:] for k, v in prompts.items()})


def _compact(t, tag, out, top5=False, nrows=cases.FW_ROWS):
    """Compact pin of a big [rows, dim] tensor: projections of every row on fixed directions + a few whole rows (+ top-5 ids)."""
    t = t.reshape(-1, t.shape[-1])
    out[f"{tag}_proj"] = (t.double() @ cases.fw_directions(t.shape[-1])).numpy()
    rows = cases.fw_rows(t.shape[0], nrows)
    out[f"{tag}_rows"] = t[rows].float().numpy()
    out[f"{tag}_rowidx"] = np.array(rows, dtype=np.int64)
    out[f"{tag}_norm"] = np.float64(t.double().norm())
    if top5:
        out[f"{tag}_top5"] = t.float().topk(5, dim=-1).indices.numpy().astype(np.int32)


FW_SPECS = {
    # outfile, decoder cases, (tower name, pixel shape, temporal) list, image size, projector rows, (canvas, grid, boxes) list
    "336": dict(out="fullwidth.npz", llama=cases.FW_LLAMA, image=336, proj_rows=cases.FW_PROJ_ROWS,
                towers=(("video336", cases.FW_VIDEO_SHAPE, True), ("image336", cases.FW_IMAGE_SHAPE, False)),
                regions=((224, 24, cases.FW_BOXES_224), (336, 24, cases.FW_BOXES_336))),
    # the reference-native 224 px shapes (round 5): N = 257, 2056 tower rows per clip, G = 16, S = 768 / 2560
    "224": dict(out="fullwidth_224.npz", llama=cases.FW224_LLAMA, image=224, proj_rows=cases.FW224_PROJ_ROWS,
                towers=(("video224", cases.FW224_VIDEO_SHAPE, True), ("image224", cases.FW224_IMAGE_SHAPE, False)),
                regions=((224, 16, cases.FW224_BOXES),)),
}


def gen_fullwidth(ns, which="336"):
    """The reference's own modules at the BASELINE widths (H=4096 / I=11008 / 32 heads; ViT-L/14, T = 8; projector 1024 -> 4096;
    RegionExtractor(1024, 4096)), weights drawn exactly as bench.py draws them. which = "336": the BASELINE image size (24 x 24
    grid); "224": the size the reference's processors are hard-wired to (16 x 16 grid, 257 tokens per frame)."""
    import contextlib
    import io
    import time
    from vitron_amd.synth import VICUNA_7B, VIT_L14
    spec = FW_SPECS[which]
    out = {}
    t0 = time.time()
    # ---- decoder: LlavaLlamaForCausalLM.forward(inputs_embeds=...) ----------------------------------------------------
    ll = ns.llava_llama
    for name, (S, L) in spec["llama"].items():
        c = dict(VICUNA_7B, num_hidden_layers=L)
        sd = synth.llama_state(c, synth.make_generator(cases.FW_SEED + L), **cases.FW_INIT)
        cfg = ll.LlavaConfig(hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"], num_hidden_layers=L,
                             num_attention_heads=c["num_attention_heads"], num_key_value_heads=c["num_attention_heads"],
                             vocab_size=c["vocab_size"], rms_norm_eps=c["rms_norm_eps"], max_position_embeddings=4096,
                             rope_theta=c["rope_theta"], tie_word_embeddings=False)
        cfg._attn_implementation = "eager"
        cfg.pretraining_tp = 1
        with contextlib.redirect_stdout(io.StringIO()):
            model = ll.LlavaLlamaForCausalLM(cfg).eval()
        missing, unexpected = model.load_state_dict(f32(sd), strict=False)
        assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
        x = cases.fw_llama_embeds(S, cases.FW_SEED + S)
        grabbed = {}
        hook = model.model.layers[-1].register_forward_hook(     # residual stream behind the last decoder layer (before model.norm)
            lambda mod, args, res: grabbed.__setitem__("h", (res[0] if isinstance(res, tuple) else res).detach()))
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            o = model(inputs_embeds=x.unsqueeze(0), use_cache=False)
        hook.remove()
        out[f"llama_{name}_checksum"] = np.float64(synth.checksum(sd))
        _compact(o.logits[0].float(), f"llama_{name}_logits", out, top5=True, nrows=4)
        _compact(grabbed["h"].reshape(S, -1).float(), f"llama_{name}_hidden", out)
        del model, sd
        print(f"llama {name}: {time.time() - t0:.1f}s", flush=True)
    # ---- towers: CLIPVisionTransformer (video / image) ---------------------------------------------------------------------
    for name, shape, time_attn in spec["towers"]:
        c = dict(VIT_L14, image_size=spec["image"], add_time_attn=time_attn, num_frames=8 if time_attn else 1,
                 num_hidden_layers=cases.FW_VIT_LAYERS)
        sd = synth.vit_state(c, synth.make_generator(cases.FW_SEED + 7), **cases.FW_INIT)
        m = build_ref_vit(ns, c, sd)
        x = cases.pixels(shape, cases.FW_SEED + 8)
        with torch.no_grad():
            hs = m(x, output_hidden_states=True).hidden_states
        out[f"vit_{name}_checksum"] = np.float64(synth.checksum(sd))
        for i, h in enumerate(hs):
            _compact(h.reshape(-1, h.shape[-1]).float(), f"vit_{name}_hidden_{i}", out)
        print(f"vit {name}: {time.time() - t0:.1f}s", flush=True)
    # ---- projector ------------------------------------------------------------------------------------------------------------
    sd = synth.projector_state(1024, 4096, synth.make_generator(cases.FW_SEED + 9), **cases.FW_INIT)
    cfg = types.SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=1024, hidden_size=4096)
    pm = ns.projector_builder.build_vision_projector(cfg).eval()
    pm.load_state_dict(f32(sd))
    x = cases.features((spec["proj_rows"], 1024), cases.FW_SEED + 10)
    with torch.no_grad():
        _compact(pm(x), "projector", out)
    out["projector_checksum"] = np.float64(synth.checksum(sd))
    # ---- region extractor: RegionExtractor(1024, 4096, image_size=canvas) on the tower's G x G grid ------------------------------
    sd = synth.region_state(1024, 4096, synth.make_generator(cases.FW_SEED + 11), **cases.FW_INIT)
    for canvas, G, boxes in spec["regions"]:
        m = ns.region_layer.RegionExtractor(1024, 4096, image_size=canvas).eval()
        m.load_state_dict(f32(sd))
        feats = cases.features((len(boxes), G * G, 1024), cases.FW_SEED + 12)
        with torch.no_grad():
            r = m(feats, boxes)
            cv = m.transform_bbox_2_mask(boxes, m.image_size, feats.device, feats.dtype).unsqueeze(1)
            grid = torch.nn.functional.interpolate(cv, size=(G, G), mode="bilinear", align_corners=False)
        out[f"region_c{canvas}_out"] = r[:, 0].numpy()
        out[f"region_c{canvas}_cells"] = (grid > 0).reshape(len(boxes), -1).numpy().astype(np.int32)
    out["region_checksum"] = np.float64(synth.checksum(sd))
    np.savez_compressed(os.path.join(OUT, spec["out"]), **out)
    print(spec["out"], {k: getattr(v, "shape", v) for k, v in out.items()}, f"{time.time() - t0:.1f}s")


def gen_fullwidth_224(ns):
    gen_fullwidth(ns, "224")


GREEDY_TINY = ("image_region", "video", "text_only", "video_image_trunc")   # the batch-1 glue cases (what app.py / inference_image.py run)
GREEDY_STEPS_TINY, GREEDY_STEPS_7B = 12, 8


def _greedy_over_reference_forward(model, embeds, embed_table, steps):
    """Hand-written greedy loop over the REFERENCE's own forward (its generate() cannot run under transformers 5.x,
    llava_arch.py:198 subscripts the cache object). Cache-free: every step re-runs LlavaLlamaForCausalLM.forward
    (llava_llama.py:57-102) on the grown embedding sequence [prompt rows ; embed_tokens(generated ids)] -- for one unpadded
    sequence that is what the cached loop computes with the decode-step fix-up of llava_arch.py:196-205 (mask of ones, position =
    length - 1), in fp32. Returns per step: arg-max id, the whole last-position logits row."""
    import contextlib
    import io
    cur = embeds
    ids, rows = [], []
    for _ in range(steps):
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            lg = model(inputs_embeds=cur, use_cache=False).logits[0, -1].float()
        nxt = int(lg.argmax())
        ids.append(nxt)
        rows.append(lg)
        cur = torch.cat([cur, embed_table[nxt].view(1, 1, -1).to(cur.dtype)], dim=1)
    return ids, torch.stack(rows)


def gen_greedy(ns):
    """Greedy token ids pinned on the REFERENCE: ids, top-2 margin and logits scale of every step, so that a GPU
    test can demand the exact id wherever the reference's margin exceeds the bf16 noise bound -- and must report where it does not.
      * the batch-1 glue cases at tiny widths through the reference's prepare_inputs_labels_for_multimodal + forward (whole logits
        rows stored: vocabulary 512);
      * one Vicuna-7B-width case (H = 4096 / I = 11008 / 32 heads, 2 layers, 1088 prompt rows = fullwidth.npz's s1088_l2 weights and
        rows): ids, top-5 ids / values, row rms and the projections on the fixed directions of every step."""
    import contextlib
    import io
    import time
    from vitron_amd.synth import VICUNA_7B
    out = {}
    t0 = time.time()
    model = build_ref_llava(ns)
    table = model.get_model().embed_tokens.weight.detach()
    for name in GREEDY_TINY:
        case = cases.glue_cases()[name]
        model.config.tokenizer_model_max_length = case.get("max_length")
        model.config.tokenizer_padding_side = case.get("padding_side", "right")
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            (_, _, _, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(
                case["input_ids"], None, case["attention_mask"], None, None, case["images"], case["regions"])
        ids, rows = _greedy_over_reference_forward(model, embeds, table, GREEDY_STEPS_TINY)
        top2 = rows.topk(2, dim=-1).values
        out[f"{name}_ids"] = np.array(ids, dtype=np.int64)
        out[f"{name}_margin"] = (top2[:, 0] - top2[:, 1]).numpy()
        out[f"{name}_logits"] = rows.numpy()
    del model
    print(f"greedy tiny: {time.time() - t0:.1f}s", {k: v.tolist() for k, v in out.items() if k.endswith("_ids")}, flush=True)
    # ---- 7B width ---------------------------------------------------------------------------------------------------------
    ll = ns.llava_llama
    name = "s1088_l2"
    S, L = cases.FW_LLAMA[name]
    c = dict(VICUNA_7B, num_hidden_layers=L)
    sd = synth.llama_state(c, synth.make_generator(cases.FW_SEED + L), **cases.FW_INIT)
    cfg = ll.LlavaConfig(hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"], num_hidden_layers=L,
                         num_attention_heads=c["num_attention_heads"], num_key_value_heads=c["num_attention_heads"],
                         vocab_size=c["vocab_size"], rms_norm_eps=c["rms_norm_eps"], max_position_embeddings=4096,
                         rope_theta=c["rope_theta"], tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    cfg.pretraining_tp = 1
    with contextlib.redirect_stdout(io.StringIO()):
        model = ll.LlavaLlamaForCausalLM(cfg).eval()
    missing, unexpected = model.load_state_dict(f32(sd), strict=False)
    assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    x = cases.fw_llama_embeds(S, cases.FW_SEED + S).unsqueeze(0)
    ids, rows = _greedy_over_reference_forward(model, x, sd["model.embed_tokens.weight"].float(), GREEDY_STEPS_7B)
    top5 = rows.topk(5, dim=-1)
    out[f"llama_{name}_checksum"] = np.float64(synth.checksum(sd))
    out[f"llama_{name}_ids"] = np.array(ids, dtype=np.int64)
    out[f"llama_{name}_margin"] = (top5.values[:, 0] - top5.values[:, 1]).numpy()
    out[f"llama_{name}_top5_ids"] = top5.indices.numpy().astype(np.int32)
    out[f"llama_{name}_top5_vals"] = top5.values.numpy()
    out[f"llama_{name}_rms"] = rows.double().pow(2).mean(-1).sqrt().numpy()
    out[f"llama_{name}_proj"] = (rows.double() @ cases.fw_directions(rows.shape[-1])).numpy()
    np.savez_compressed(os.path.join(OUT, "greedy.npz"), **out)
    print("greedy.npz", {k: (v.tolist() if v.size <= 16 else v.shape) for k, v in out.items() if not k.endswith("_logits")},
          f"{time.time() - t0:.1f}s")


def _padded_batch_greedy_hf431(model, embeds, spliced_mask, position_ids, ids_mask, embed_table, steps):
    """The reference's B > 1 generate(), restated step by step over the REFERENCE's own forward (with its KV cache):
      * prefill: LlavaLlamaForCausalLM.forward on the spliced, right-padded batch with the spliced mask / positions that
        prepare_inputs_labels_for_multimodal returned (llava_arch.py:520-573);
      * next token of every sample = arg-max of the LAST column of the logits (transformers 4.31 GenerationMixin.greedy_search:
        `outputs.logits[:, -1, :]` -- for a right-padded shorter sample that is a PAD row, not its last valid position);
      * every later step: GenerationMixin appends one column of ones to the mask it was GIVEN (the input_ids-length mask, NOT the
        spliced one), and the reference's fix-up (llava_arch.py:196-205) pads that with ones up to past_len + 1 and sets
        position_ids = sum(mask) - 1. So the pad rows of the spliced batch are attended, and the zeros of the ids-length mask land
        on whatever spliced rows share their index.
    transformers 4.31 itself is not in this image (5.x is): the two GenerationMixin rules above are restated from its published
    greedy_search / _update_model_kwargs_for_generation; the arithmetic of every step is the reference's own forward."""
    import contextlib
    import io
    B = embeds.shape[0]
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        out = model(inputs_embeds=embeds, attention_mask=spliced_mask, position_ids=position_ids, use_cache=True)
    past = out.past_key_values
    nxt = out.logits[:, -1].float().argmax(-1)
    ids = [nxt]
    mask = ids_mask.clone()
    past_len = embeds.shape[1]
    for _ in range(steps - 1):
        mask = torch.cat([mask, torch.ones((B, 1), dtype=mask.dtype)], dim=1)                 # _update_model_kwargs_for_generation
        m = torch.cat([mask, torch.ones((B, past_len + 1 - mask.shape[1]), dtype=mask.dtype)], dim=1)   # llava_arch.py:197-203
        pos = m.sum(1, keepdim=True) - 1                                                       # :204
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            out = model(inputs_embeds=embed_table[nxt].unsqueeze(1), attention_mask=m, position_ids=pos, past_key_values=past,
                        use_cache=True)
        past = out.past_key_values
        past_len += 1
        nxt = out.logits[:, -1].float().argmax(-1)
        ids.append(nxt)
    return torch.stack(ids, dim=1)


def gen_greedy_batch(ns):
    """What B > 1 generate() returns in the reference, next to what each sample returns ALONE (batch 1), for the right-padded
    `batch_pad` glue case: this repo packs the batch, so it returns the batch-1 rows; the fixture keeps both so that a GPU test can
    assert the first and REPORT the distance to the second."""
    import contextlib
    import io
    model = build_ref_llava(ns)
    table = model.get_model().embed_tokens.weight.detach()
    case = cases.glue_cases()["batch_pad"]
    model.config.tokenizer_model_max_length = None
    model.config.tokenizer_padding_side = "right"
    out = {}
    # transformers 4.31 LlamaForCausalLM.prepare_inputs_for_generation hands the prefill position_ids = cumsum(mask) - 1 (1 at the
    # pads), so prepare_inputs_labels_for_multimodal returns ITS OWN spliced positions (llava_arch.py:535-566: arange over the valid
    # rows, 0 at the pads) instead of None
    am = case["attention_mask"]
    pos_in = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 1)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        (_, pos, mask, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(
            case["input_ids"], pos_in, am, None, None, case["images"], case["regions"])
    B = embeds.shape[0]
    assert pos is not None
    batch_ids = _padded_batch_greedy_hf431(model, embeds, mask, pos, case["attention_mask"], table, GREEDY_STEPS_TINY)
    out["batch_pad_padded_ids"] = batch_ids.numpy().astype(np.int64)
    out["batch_pad_spliced_lengths"] = mask.long().sum(1).numpy()
    for b in range(B):
        n = int(case["attention_mask"][b].sum())
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            (_, _, _, _, e1, _) = model.prepare_inputs_labels_for_multimodal(
                case["input_ids"][b:b + 1, :n], None, None, None, None, [case["images"][b]], [case["regions"][b]])
        ids, rows = _greedy_over_reference_forward(model, e1, table, GREEDY_STEPS_TINY)
        top2 = rows.topk(2, dim=-1).values
        out[f"batch_pad_alone{b}_ids"] = np.array(ids, dtype=np.int64)
        out[f"batch_pad_alone{b}_margin"] = (top2[:, 0] - top2[:, 1]).numpy()
        out[f"batch_pad_alone{b}_rms"] = rows.double().pow(2).mean(-1).sqrt().numpy()
    np.savez_compressed(os.path.join(OUT, "greedy_batch.npz"), **out)
    print("greedy_batch.npz", {k: v.tolist() for k, v in out.items()})


def gen_f1(ns):
    """SURVEY.md 8(a) row F1's other branches, from the reference's own classes:
      * CLIPVisionTower (vitron/model/multimodal_encoder/clip_encoder.py:7-78) around a transformers CLIPVisionModel with seeded
        weights (no checkpoint / network: the model is attached the way load_model() would, :22-27): forward() for
        select_feature 'patch' and 'cls_patch', select_layer -2, batched and list inputs;
      * build_vision_projector with mm_projector_type = 'mlp3x_gelu' (multimodal_projector/builder.py:39-46)."""
    import importlib
    from transformers import CLIPVisionConfig, CLIPVisionModel
    ce = importlib.import_module("vitron.model.multimodal_encoder.clip_encoder")
    out = {}
    cfg = cases.CLIP_TOWER
    sd = synth.vit_state(cfg, synth.make_generator(cases.SEED_VIT + 2), **cases.VIT_INIT)
    hc = CLIPVisionConfig(hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                          num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
                          image_size=cfg["image_size"], patch_size=cfg["patch_size"], hidden_act=cfg["hidden_act"],
                          layer_norm_eps=cfg["layer_norm_eps"])
    hc._attn_implementation = "eager"
    vm = CLIPVisionModel(hc).eval()
    # transformers 4.31 (the reference's pin) names these tensors `vision_model.*`; the 5.x build installed here drops the prefix
    pref = "vision_model." if any(k.startswith("vision_model.") for k in vm.state_dict()) else ""
    missing, unexpected = vm.load_state_dict({pref + k: v.float() for k, v in sd.items()}, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    x = cases.pixels(cases.CLIP_TOWER_SHAPE, cases.SEED_PIX + 21)
    for feat in ("patch", "cls_patch"):
        t = ce.CLIPVisionTower.__new__(ce.CLIPVisionTower)
        nn.Module.__init__(t)
        t.is_loaded, t.select_layer, t.select_feature, t.vision_tower = True, -2, feat, vm
        with torch.no_grad():
            out[f"clip_{feat}"] = t(x).numpy()
            lst = t([x[0], x[2]])
        out[f"clip_{feat}_list0"], out[f"clip_{feat}_list1"] = lst[0].numpy(), lst[1].numpy()
    out["clip_checksum"] = np.float64(synth.checksum(sd))
    H = cases.LLM["hidden_size"]
    g = synth.make_generator(cases.SEED_PROJ + 3)
    psd = synth.projector_state(cases.MM_HIDDEN, H, g, **cases.MLP_INIT)
    extra = synth.projector_state(H, H, g, **cases.MLP_INIT)
    psd["4.weight"], psd["4.bias"] = extra["2.weight"], extra["2.bias"]
    pcfg = types.SimpleNamespace(mm_projector_type="mlp3x_gelu", mm_hidden_size=cases.MM_HIDDEN, hidden_size=H)
    pm = ns.projector_builder.build_vision_projector(pcfg).eval()
    pm.load_state_dict(f32(psd))
    xp = cases.features((cases.PROJ3_ROWS, cases.MM_HIDDEN), cases.SEED_FEATS + 5)
    with torch.no_grad():
        out["proj3_out"] = pm(xp).numpy()
    out["proj3_checksum"] = np.float64(synth.checksum(psd))
    np.savez_compressed(os.path.join(OUT, "f1.npz"), **out)
    print("f1.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    torch.set_num_threads(8)
    ns = ref_shim.install()
    if "--keys-only" in sys.argv:
        gen_state_dict_keys(ns)
        sys.exit(0)
    if "--fullwidth-only" in sys.argv:
        gen_fullwidth(ns)
        sys.exit(0)
    if "--vit-tmlp-only" in sys.argv:
        gen_vit_tmlp(ns)
        sys.exit(0)
    if "--fullwidth-224-only" in sys.argv:
        gen_fullwidth_224(ns)
        sys.exit(0)
    if "--greedy-only" in sys.argv:
        gen_greedy(ns)
        sys.exit(0)
    if "--greedy-batch-only" in sys.argv:
        gen_greedy_batch(ns)
        sys.exit(0)
    if "--f1-only" in sys.argv:
        gen_f1(ns)
        sys.exit(0)
    gen_mm_utils(ns)
    gen_output_parser()
    gen_vit(ns)
    gen_vit_b(ns)
    gen_vit_tmlp(ns)
    gen_region_projector(ns)
    gen_glue(ns)
    gen_glue_random(ns)
    gen_state_dict_keys(ns)
    gen_fullwidth(ns)
    gen_fullwidth_224(ns)
    gen_greedy(ns)
    gen_greedy_batch(ns)
    gen_f1(ns)
