"""Full-depth goldens of the BENCHMARK workloads from the REFERENCE's own code, and the reference's CPU time for them.

    python tests/golden/make_golden_fulldepth.py [case ...]     # build container only: needs /root/reference, ~45 GB of RAM

Runs the reference's LlavaLlamaForCausalLM (32-layer Vicuna-7B shape, fp32, eager attention) with its LanguageBind towers
(ViT-L/14, 24 layers; the video tower with temporal attention over 8 frames), mlp2x_gelu projector and RegionExtractor attached
-- through the reference's own prepare_inputs_labels_for_multimodal + forward -- on

  c3       BASELINE configs[2]: one 8-frame 336 x 336 clip + a 512-token prompt (S = 5120)                       (round 2)
  c3_224   the same at the image size the reference's processors are hard-wired to (224 px: N = 257, S = 2560)    (round 5)
  c2       BASELINE configs[1]: one 336 x 336 image + a 512-token prompt (S = 1088)                               (round 5)
  c2_224   the same at 224 px (S = 768)                                                                           (round 5)
  c5       BASELINE configs[4]: four (336 px image + box + prompt) samples, each ALONE (batch 1, what app.py and
           inference_image.py run) through prefill + 16 greedy steps of the reference's forward with its KV cache and the
           decode-step fix-up of llava_arch.py:196-205 (mask of ones, position = past length)                    (round 5)
  c5_224   the same at 224 px (G = 16: RegionExtractor exactly as the reference ships it, layer.py:60)           (round 5)

Weights come from vitron_amd.synth.HashGenerator, a counter-based stream that is bit-identical on CPU and GPU, so
tests/test_gpu_parity_fulldepth.py can rebuild the very same 7B weights on the GPU box in seconds.

Writes
  tests/golden/fulldepth_<case>.npz   last-position logits, top-5 ids of every position, projections / sample rows of the logits,
                                      of the final hidden state and of the spliced input embeddings (visual tokens included);
                                      c5*: per sample the greedy ids, top-5 ids / values, margins and projections of every step
  profiles/r5_cpu_reference.json      wall-clock of the reference on this container's host cores per case (the c3 entry of
                                      round 2 stays in profiles/r2_cpu_reference.json)
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_model, ref_shim  # noqa: E402
from tests.golden import cases, make_golden  # noqa: E402
from vitron_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 20260926
FRAMES, IMAGE, TEXT = 8, 336, 512
GREEDY_STEPS = 16
WIDE_TAIL_CASES = ("c3", "c2_224")   # cases whose pin also holds the last 64 logits rows whole (8 MB each)
# the four boxes of C5 in the reference's 224-pixel space (whole canvas, two interior boxes, a single 2 x 2-centre hit)
C5_BOXES = [[0, 0, 224, 224], [0, 58.9, 117.9, 117.9], [100, 20, 180, 200], [7, 7, 8, 8]]

CASES = {
    "c3": dict(kind="clip", image=336, seed=0),
    "c3_224": dict(kind="clip", image=224, seed=100),
    "c2": dict(kind="image", image=336, seed=200),
    "c2_224": dict(kind="image", image=224, seed=300),
    "c5": dict(kind="region", image=336, seed=400),
    "c5_224": dict(kind="region", image=224, seed=500),
}


def case_inputs(name):
    """Pixels, prompt ids (and boxes) of a case (CPU generators: cheap to regenerate anywhere).
    clip / image: (pixels, ids [1, L]); region: list of (image, ids [1, L], box) -- one entry per sample."""
    c = CASES[name]
    s = SEED + c["seed"]
    if c["kind"] == "region":
        g = torch.Generator().manual_seed(s + 2)
        out = []
        for b, box in enumerate(C5_BOXES):
            img = cases.pixels((3, c["image"], c["image"]), s + 20 + b)
            rnd = lambda k: torch.randint(3, 32000, (k,), generator=g)      # noqa: E731
            # app.py:525-534: ' ' + <image> + '\n' + <objs> + ' ' + user input
            ids = torch.cat([torch.tensor([1, -200]), rnd(6), torch.tensor([-300]), rnd(24)]).unsqueeze(0)
            out.append((img, ids, box))
        return out
    g = torch.Generator().manual_seed(s + 2)
    text = torch.randint(3, 32000, (TEXT - 1,), generator=g)
    if c["kind"] == "clip":
        pix = cases.pixels((3, FRAMES, c["image"], c["image"]), s + 1)
        ids = torch.cat([torch.tensor([1]), torch.full((FRAMES,), -200), text]).unsqueeze(0)
    else:
        pix = cases.pixels((3, c["image"], c["image"]), s + 1)
        ids = torch.cat([torch.tensor([1, -200]), text]).unsqueeze(0)
    return pix, ids


def tower_cfg(name):
    c = CASES[name]
    video = c["kind"] == "clip"
    return dict(synth.VIT_L14, image_size=c["image"], add_time_attn=video, num_frames=FRAMES if video else 1)


def llama_weights(device="cpu"):
    return synth.llama_state(synth.VICUNA_7B, synth.HashGenerator(SEED + 10), device)


def case_weights(name, device="cpu"):
    """(tower config, tower state dict, projector state dict, region extractor state dict), bf16, device-independent hash streams.
    The decoder (llama_weights), the projector and the region extractor are the same for every case; the towers differ by kind and
    position-table length."""
    vcfg = tower_cfg(name)
    vsd = synth.vit_state(vcfg, synth.HashGenerator(SEED + 11), device)
    psd = synth.projector_state(1024, 4096, synth.HashGenerator(SEED + 12), device)
    rsd = synth.region_state(1024, 4096, synth.HashGenerator(SEED + 13), device)
    return vcfg, vsd, psd, rsd


def c3_inputs():
    return case_inputs("c3")


def c3_weights(device="cpu"):
    """(llama state dict, video tower state dict, projector state dict, tower config) of case c3 (round-2 signature)."""
    vcfg, vsd, psd, _ = case_weights("c3", device)
    return llama_weights(device), vsd, psd, vcfg


def _attach_tower(ns, model, name):
    """The case's tower (image or video), the projector and the RegionExtractor on the reference's model (oracle/ref_model.py)."""
    vcfg, vsd, psd, rsd = case_weights(name)
    ref_model.attach(ns, model, vcfg, vsd, psd, rsd)


def _run_prefill_case(model, name):
    """clip / image cases: the reference's prepare_inputs_labels_for_multimodal + forward over all positions."""
    pix, ids = case_inputs(name)
    grabbed = {}
    hook = model.model.layers[-1].register_forward_hook(
        lambda mod, args, res: grabbed.__setitem__("h", (res[0] if isinstance(res, tuple) else res).detach()))
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        ta = time.perf_counter()
        (_, pos, mask, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [pix], None)
        tb = time.perf_counter()
        logits = model(inputs_embeds=embeds, use_cache=False).logits
        tc = time.perf_counter()
    hook.remove()
    S = embeds.shape[1]
    G2 = (CASES[name]["image"] // 14) ** 2
    assert S == (FRAMES if CASES[name]["kind"] == "clip" else 1) * G2 + TEXT, S
    out = {}
    lg = logits[0].float()
    out["last_logits"] = lg[-1].numpy()
    make_golden._compact(lg, "logits", out, top5=True, nrows=4)
    # round 6: a wider pin of the logits beside the first one (whose keys keep their round-5 values bit for bit) --
    # projections of EVERY row on 32 more directions, and for the headline case and the worst case the last 64 rows whole
    out["logits_proj32"] = (lg.double() @ cases.fw_directions(lg.shape[-1], n=32, seed=cases.FW_SEED + 1)).numpy()
    if name in WIDE_TAIL_CASES:
        out["logits_tail"] = lg[-64:].numpy()
    make_golden._compact(grabbed["h"].reshape(S, -1).float(), "hidden", out)
    make_golden._compact(embeds[0].float(), "embeds", out)
    out["S"] = np.int64(S)
    timing = {"S": int(S), "seconds_towers_projector_splice": tb - ta, "seconds_decoder": tc - tb, "seconds_total": tc - ta,
              "tokens_per_s": S / (tc - ta)}
    return out, timing


def _run_region_case(model, name, table):
    """c5*: every (image, box, prompt) sample alone: prefill through the reference's glue + forward (with its KV cache), then greedy
    steps through the reference's forward with the decode-step fix-up of llava_arch.py:196-205 restated (its own generate() cannot
    run under transformers 5.x: llava_arch.py:198 subscripts the cache object)."""
    out, t_prefill, t_decode, S_all = {}, 0.0, 0.0, []
    for b, (img, ids, box) in enumerate(case_inputs(name)):
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            ta = time.perf_counter()
            (_, _, _, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [img], [box])
            o = model(inputs_embeds=embeds, use_cache=True)
            tb = time.perf_counter()
            past, past_len = o.past_key_values, embeds.shape[1]
            rows = [o.logits[0, -1].float()]
            for _ in range(GREEDY_STEPS - 1):
                nxt = int(rows[-1].argmax())
                m = torch.ones((1, past_len + 1), dtype=torch.long)                # llava_arch.py:197-203
                pos = m.sum(1, keepdim=True) - 1                                   # :204
                o = model(inputs_embeds=table[nxt].view(1, 1, -1), attention_mask=m, position_ids=pos, past_key_values=past, use_cache=True)
                past, past_len = o.past_key_values, past_len + 1
                rows.append(o.logits[0, -1].float())
            tc = time.perf_counter()
        rows = torch.stack(rows)
        top5 = rows.topk(5, dim=-1)
        S_all.append(int(embeds.shape[1]))
        out[f"s{b}_S"] = np.int64(embeds.shape[1])
        out[f"s{b}_ids"] = rows.argmax(-1).numpy().astype(np.int64)
        out[f"s{b}_margin"] = (top5.values[:, 0] - top5.values[:, 1]).numpy()
        out[f"s{b}_top5_ids"] = top5.indices.numpy().astype(np.int32)
        out[f"s{b}_top5_vals"] = top5.values.numpy()
        out[f"s{b}_rms"] = rows.double().pow(2).mean(-1).sqrt().numpy()
        out[f"s{b}_proj"] = (rows.double() @ cases.fw_directions(rows.shape[-1])).numpy()
        out[f"s{b}_first_logits"] = rows[0].numpy()
        make_golden._compact(embeds[0].float(), f"s{b}_embeds", out)
        t_prefill += tb - ta
        t_decode += tc - tb
        print(f"  {name} sample {b}: S = {embeds.shape[1]}, ids {out[f's{b}_ids'].tolist()}, min margin {out[f's{b}_margin'].min():.4f}", flush=True)
    timing = {"samples": len(S_all), "S": S_all, "greedy_steps": GREEDY_STEPS, "seconds_prefill_all_samples": t_prefill,
              "seconds_decode_all_samples": t_decode, "seconds_per_decode_step_batch1": t_decode / (len(S_all) * (GREEDY_STEPS - 1))}
    return out, timing


def main(argv):
    names = [a for a in argv if a in CASES] or ["c3_224", "c2", "c2_224", "c5", "c5_224"]
    torch.set_num_threads(os.cpu_count() or 8)
    ns = ref_shim.install()
    t0 = time.time()
    lsd = llama_weights()
    print(f"weights: {time.time() - t0:.0f}s", flush=True)
    model = ref_model.build_decoder(ns, synth.VICUNA_7B, lsd)
    table = model.get_model().embed_tokens.weight.detach()
    print(f"decoder built: {time.time() - t0:.0f}s", flush=True)
    rep_path = os.path.join(ROOT, "profiles", "r2_cpu_reference.json" if names == ["c3"] else "r5_cpu_reference.json")
    rep = {}
    if os.path.exists(rep_path) and names != ["c3"]:
        with open(rep_path) as f:
            rep = json.load(f)
    for name in names:
        _attach_tower(ns, model, name)
        print(f"{name}: towers attached at {time.time() - t0:.0f}s; running the reference ...", flush=True)
        if CASES[name]["kind"] == "region":
            out, timing = _run_region_case(model, name, table)
        else:
            out, timing = _run_prefill_case(model, name)
        np.savez_compressed(os.path.join(OUT, f"fulldepth_{name}.npz"), **out)
        entry = {"what": "the REFERENCE's own modules (oracle/ref_shim.py import shim, nothing under /root/reference modified): LanguageBind "
                         "tower (24 layers; hidden_states[-2] selected) + mlp2x_gelu projector (+ RegionExtractor) + "
                         "prepare_inputs_labels_for_multimodal + LlavaLlamaForCausalLM.forward (32 layers), fp32, eager attention, full depth, run once",
                 "case": name, "image_size": CASES[name]["image"], "kind_of_case": CASES[name]["kind"],
                 "where": "build container (no GPU): host CPU of this container, not of the GPU box",
                 "cores": os.cpu_count(), "torch_threads": torch.get_num_threads(), "kind": "reference",
                 "script": "tests/golden/make_golden_fulldepth.py", **timing}
        print(json.dumps(entry), flush=True)
        if names == ["c3"]:
            rep = entry
        else:
            rep[name] = entry
        os.makedirs(os.path.dirname(rep_path), exist_ok=True)
        with open(rep_path, "w") as f:
            json.dump(rep, f, indent=1)
        print(f"fulldepth_{name}.npz", {k: getattr(v, "shape", v) for k, v in out.items()}, flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
