"""Full-depth parity of the BENCHMARK workloads against the REFERENCE ("at full depth report the value and
top-1 / top-5 agreement"), through the 23-layer tower, the projector, (the region extractor,) the splice and all 32 decoder layers
of vitron_amd, compared with tests/golden/fulldepth_<case>.npz: the outputs of the reference's OWN modules (fp32, eager, CPU) on
the same weights and inputs, written once in the build container by tests/golden/make_golden_fulldepth.py:

  c3 / c3_224   BASELINE configs[2]: one 8-frame clip + a 512-token prompt at 336 px (S = 5120) and at the reference-native 224 px
                (S = 2560: the only clip shape a real Vitron checkpoint runs, processing_video.py:50-51)
  c2 / c2_224   BASELINE configs[1]: one image + a 512-token prompt (S = 1088 / 768)
  c5 / c5_224   BASELINE configs[4]: four (image + box + prompt) samples, each alone through prefill + 16 greedy steps on the paged KV
                cache -- token ids held to the ids of the reference's loop (llava_arch.py:196-205, inference_image.py:52-61)

The 7B weights are not shipped: both sides draw them from vitron_amd.synth.HashGenerator, a counter-based stream that is
bit-identical on CPU and GPU. One model per operand build is constructed and shared by all cases (towers re-loaded per case).

bf16 storage puts a 32-layer chain ~1e-2 (rel-L2 of the logits) from an fp32 evaluation -- the emulating oracle shows the same
floor at reduced depth (tests/test_gpu_parity_fullwidth.py) -- so the asserted bounds are the measured floor with margin, and the
numbers themselves (printed, collected into profiles/ by VT_PARITY_REPORT) are the result.

Token ids (c5*): teacher-forced on the reference's ids, the device's arg-max must EQUAL the reference's id at every step whose
reference margin (top-1 minus top-2 logit) exceeds the noise bound 3 * sqrt(2) * tol * rms(logits row) (tol = the build's full-depth
logits distance); the remaining steps must pick one of the reference's near-tied candidates. The number of asserted / exempt steps
is printed and a floor is asserted. A free-running generate() per sample is compared up to the first undecidable step.
"""
import json
import os

import numpy as np
import pytest
import torch

from tests import fullwidth_util as FW
from tests.golden import cases
from tests.golden import make_golden_fulldepth as FD

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


# bounds = measurements x 1.5. bf16: profiles/r3_parity_fulldepth.json (embeddings 4.2e-3, logits 1.11e-2, hidden 1.12e-2, last position
# 1.01e-2, top-1 0.969, top-5 overlap 0.976 -- the whole residual is bf16 storage of 55 layers vs fp32). fp16 (round 4, the reference's own
# inference dtype; libvitron_hip_f16.so): profiles/r4_parity_fulldepth_fp16.json. Round 5: the same bounds hold the 224 px and the image-tower
# cases (profiles/r5_parity_fulldepth_*.json).
BOUNDS = {
    "bf16": dict(embeds=6.5e-3, last=1.6e-2, rows=1.7e-2, proj=1.8e-2, top1=0.95, top5=0.95),
    "fp16": dict(embeds=1.0e-3, last=2.4e-3, rows=2.6e-3, proj=2.7e-3, top1=0.985, top5=0.985),
}
# precise_qk (round 5): q / k and the norm output that feeds their projection travel as hi + lo operand pairs through the
# prefill. It takes 1-2 layer chains on token-sized rows below 1e-3 (tests/test_gpu_parity_fullwidth.py: 7.1e-4 / 9.3e-4, asserted), but at
# FULL depth behind real visual rows the q / k path is not what carries the distance (profiles/r5_parity_round_points_fulldepth.txt: the A
# operands of gate/up, o_proj and down_proj and the tower's own 5e-4 in the embeddings do): measured c3 1.41e-3 -> 1.41e-3 (last 1.31 -> 1.25),
# c2_224 1.63e-3 -> 1.50e-3 (last 1.67 -> 1.41). Asserted: never worse than the standard build's bounds, and the last position no worse than x 1.02.
BOUNDS["fp16-precise"] = dict(BOUNDS["fp16"])
# precise level 2 (round 5): EVERY GEMM A operand of the prefill an operand pair, from the towers' MLPs through the projector (whose output
# enters the residual stream unrounded) to the lm_head -- the verification mode in which north_star's 1e-3 against the reference's fp32
# logits is ASSERTED at full depth (what is left: the attention paths' single fp16 stores -- V^T, P, the towers' q / k / v / attention output)
# measured (profiles/r5_parity_fulldepth_precise.json): embeddings 0.7e-4 (image tower) / 1.9e-4 (video tower), logits over all rows 4.3e-4 .. 4.7e-4,
# last position 1.0e-4 .. 3.5e-4, top-1 99.9 .. 100 %; bounds = x 1.5, all inside north_star's 1e-3
# later in round 5 the towers' attention paths joined (pairs through the scores at head_dim 64, the temporal attention in fp32): embeddings
# 1.4e-5 .. 2.8e-5 in BOTH builds, logits 3.9e-4 .. 4.7e-4 (fp16 3.87 / 3.90 / 4.23 / 4.60, bf16 3.93 / 3.95 / 4.26 / 4.67), last position
# <= 3.4e-4, top-1 99.94 .. 100 %
BOUNDS["fp16-precise2"] = dict(embeds=5.0e-5, last=5.5e-4, rows=7.0e-4, proj=5.5e-4, top1=0.998, top5=0.998)
# the bf16 build (the benchmark dtype) in precise level 2: one operand pair carries 16 mantissa bits, v goes straight from fp32 into the pages'
# fp16, P is fp16 in both builds -- the same numbers as the fp16 build's (standard mode: 1.1 .. 1.3e-2), the same bounds
BOUNDS["bf16-precise2"] = dict(BOUNDS["fp16-precise2"])
# precise level 3 (round 6): the fp16 build with every decoder Linear of the prefill as ONE launch that adds the MX-FP4 product of the
# A operand's rounding remainder on the 4x-rate MX pipe (towers and projector in level 2, the lm_head's operand a 16-bit pair): the mode in which
# north_star's 1e-3 is met at ~1.25x of the standard step instead of 2.3x. The oracle's emulation (tests/parity_mx_fulldepth.py) puts c2_224 at
# 6.1e-4 over all rows / 6.6e-4 at the last position -- q, k, V^T and P keep their single 16-bit stores; asserted inside 1e-3 with what margin there is
# (the towers run their level 1 there: MLP operands and output features as pairs, attention paths standard -- embeddings 1.8e-4 instead of 2e-5)
BOUNDS["fp16-precise3"] = dict(embeds=3.0e-4, last=9.5e-4, rows=9.0e-4, proj=9.0e-4, top1=0.996, top5=0.996)
ID_TOL = {"bf16": 1.6e-2, "fp16": 2.4e-3}      # logits distance that sets the noise bound of the id comparison (= BOUNDS[op]["last"])
REPORT = {}


def _note(name, rep):
    REPORT[name] = rep
    print(f"[parity-fulldepth] {name}: " + json.dumps(rep), flush=True)
    out = os.environ.get("VT_PARITY_REPORT")
    if out:
        with open(out.replace(".json", "_fulldepth_all.json"), "w") as f:
            json.dump(REPORT, f, indent=1)


@pytest.fixture(scope="module", params=["bf16", "fp16"])
def full(request):
    """(operand name, dtype, model): the 32-layer Vicuna-7B-shaped model with both towers, projector and region extractor."""
    from vitron_amd import _lib, synth
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM
    op = request.param
    _lib.load(operand=op)
    odt = _lib.torch_dtype(op)
    dev = torch.device("cuda:0")
    model = LlavaLlamaForCausalLM(LlavaConfig(**synth.VICUNA_7B, mm_hidden_size=1024, mm_video_tower="fulldepth/LanguageBind_Video_merge",
                                              mm_image_tower="fulldepth/LanguageBind_Image", kv_prefix_reuse=False))
    vcfg, vsd, psd, rsd = FD.case_weights("c3", dev)
    model.get_video_tower().load_state(vcfg, vsd)
    icfg, isd, _, _ = FD.case_weights("c2", dev)
    model.get_image_tower().load_state(icfg, isd)
    sd = dict(FD.llama_weights(dev))
    sd.update({"model.mm_projector." + k: v for k, v in psd.items()})
    sd.update({"model.region_extractor." + k: v for k, v in rsd.items()})
    model.load_state_dict(sd)
    model.to(dev, dtype=odt)
    assert model.dtype == odt and model.get_video_tower().dtype == odt and model.get_image_tower().dtype == odt
    del vsd, isd, psd, rsd, sd
    yield op, odt, model
    model.reset_prefix_cache()
    del model
    torch.cuda.empty_cache()


def _load_tower(model, name, dev, odt, loaded={}):
    """(Re)load the tower the case uses with the case's own weights (the position table depends on the image size)."""
    key = (id(model), FD.CASES[name]["kind"] == "clip")
    if loaded.get(key) == FD.CASES[name]["image"]:
        return
    vcfg, vsd, _, _ = FD.case_weights(name, dev)
    tower = model.get_video_tower() if vcfg["add_time_attn"] else model.get_image_tower()
    tower.load_state(vcfg, vsd)
    tower.to(device=dev, dtype=odt)
    loaded[key] = FD.CASES[name]["image"]


@pytest.mark.parametrize("precise", [0, 1, 2, 3], ids=["standard", "precise_qk", "precise2", "precise3"])
@pytest.mark.parametrize("name", ["c3", "c3_224", "c2", "c2_224"])
def test_prefill_full_depth_vs_reference(full, name, precise):
    op, odt, model = full
    if precise in (1, 3) and op != "fp16":
        pytest.skip("precise_qk (level 1) and level 3 are asserted on the fp16 build (the reference's dtype) only; level 2 on both")
    dev = torch.device("cuda:0")
    _load_tower(model, name, dev, odt)          # (before set_precise: a re-packed tower starts in the standard mode)
    model.set_precise(precise)
    try:
        _prefill_case(op + {0: "", 1: "-precise", 2: "-precise2", 3: "-precise3"}[precise], odt, model, name, dev)
    finally:
        model.set_precise(0)


def _prefill_case(op, odt, model, name, dev):
    from vitron_amd.engine import SequenceState, llama_forward
    g = np.load(os.path.join(GOLD, f"fulldepth_{name}.npz"))
    _load_tower(model, name, dev, odt)
    pix, ids = FD.case_inputs(name)
    (_, _, _, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids.to(dev), None, None, None, None, [pix.to(dev).to(odt)], None,
                                                                         input_ids_host=ids)
    from vitron_amd.engine import pair_lo
    S = int(g["S"])
    assert embeds.shape[1] == S
    lo = pair_lo(embeds)                          # precise level 2: the spliced embeddings are an operand pair
    assert (lo is not None) == op.endswith(("precise2", "precise3"))
    e_val = embeds[0].float().cpu() + (lo[0].float().cpu() if lo is not None else 0.0)
    e_proj, e_rows = FW.vs_pin(e_val, g, "embeds")
    llama = model.get_model().llama
    model._ensure_kv((S + 63) // 64 + 2)
    seq = SequenceState()
    logits, hidden = llama_forward(llama, model.kv, [seq], embeds[0], [S], logit_rows=list(range(S)), return_hidden=True,
                                   embeds_lo=None if lo is None else lo[0])
    model.kv.release(seq.pages)
    logits, hidden = logits.float().cpu(), hidden.float().cpu()
    l_proj, l_rows = FW.vs_pin(logits, g, "logits")
    l_proj32, l_tail = FW.vs_wide_pin(logits, g)       # round 6: 32 more directions over all rows; the last 64 rows whole (c3, c2_224)
    h_proj, h_rows = FW.vs_pin(hidden, g, "hidden")
    last = FW.rel(logits[-1], g["last_logits"])
    top1, top5 = FW.topk_agreement(logits, g, "logits")
    last_top1 = int(logits[-1].argmax()) == int(np.argmax(g["last_logits"]))
    rep = {"operand": op, "case": name, "image_size": FD.CASES[name]["image"],
           "workload": f"full depth (23 ViT layers + projector + 32 decoder layers), S = {S}, vs the reference's fp32 output",
           "visual_plus_text_embeddings_rel_l2_rows": e_rows, "embeddings_rel_l2_proj": e_proj,
           "final_hidden_rel_l2_rows": h_rows, "final_hidden_rel_l2_proj": h_proj,
           "logits_rel_l2_rows": l_rows, "logits_rel_l2_proj": l_proj, "logits_rel_l2_proj32": l_proj32, "logits_rel_l2_last_64_rows": l_tail,
           "last_position_logits_rel_l2": last,
           "top1_agreement_all_positions": top1, "top5_overlap_all_positions": top5, "last_position_top1_equal": bool(last_top1)}
    _note(f"{name}_{op}", rep)
    out = os.environ.get("VT_PARITY_REPORT")
    if out and name == "c3":                     # the round-2 .. round-4 report files of the headline case keep their names
        with open(out.replace(".json", f"_fulldepth_{op}.json"), "w") as f:
            json.dump(rep, f, indent=1)
    b = BOUNDS[op]
    assert e_rows <= b["embeds"] and e_proj <= b["embeds"], (e_rows, e_proj)
    assert last <= b["last"] and l_rows <= b["rows"] and h_rows <= b["rows"] and l_proj <= b["proj"] and h_proj <= b["proj"], (last, l_rows, h_rows, l_proj, h_proj)
    assert l_proj32 is not None and l_proj32 <= b["proj"], l_proj32
    assert l_tail is None or l_tail <= b["rows"], l_tail
    assert (l_tail is not None) == (name in FD.WIDE_TAIL_CASES)
    assert top5 >= b["top5"] and top1 >= b["top1"], (top1, top5)
    # the greedy first token: equal wherever the reference's own top-2 margin exceeds the build's noise bound
    ll = np.sort(g["last_logits"])[::-1]
    bound = 3.0 * (2.0 ** 0.5) * BOUNDS[op]["last"] * float(np.sqrt(np.mean(g["last_logits"].astype(np.float64) ** 2)))
    assert last_top1 or (ll[0] - ll[1]) <= bound, (float(ll[0] - ll[1]), bound)


@pytest.mark.parametrize("precise", [0, 3], ids=["standard", "precise3"])
@pytest.mark.parametrize("name", ["c5", "c5_224"])
def test_region_prompt_greedy_full_depth_vs_reference(full, name, precise):
    """BASELINE configs[4] at FULL depth: (image + box + prompt) -> image tower -> region_extractor -> projector -> splice -> 32-layer
    prefill -> 16 greedy steps on the paged KV cache (device-resident decode state), each sample alone as app.py / inference_image.py
    run it, against the ids / top-5 values / projections of the reference's own loop."""
    from vitron_amd import ops
    from vitron_amd.engine import DecodeState, SequenceState, llama_forward
    op, odt, model = full
    if precise and op != "fp16":
        pytest.skip("precise level 3 is asserted on the fp16 build")
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(GOLD, f"fulldepth_{name}.npz"))
    _load_tower(model, name, dev, odt)
    llama = model.get_model().llama
    tol = ID_TOL[op]
    n = FD.GREEDY_STEPS
    # precise = 3 (round 6): the PREFILL of every sample in precise level 3 (region extractor and decode steps are the standard kernels): the first
    # logits row must be inside north_star's 1e-3 of the reference's, the ids as good as the standard mode's
    model.set_precise(precise)
    try:
        _region_greedy_case(op, odt, model, name, precise, g, llama, tol, n, dev)
    finally:
        model.set_precise(0)


def _region_greedy_case(op, odt, model, name, precise, g, llama, tol, n, dev):
    from vitron_amd import ops
    from vitron_amd.engine import DecodeState, SequenceState, llama_forward, pair_lo
    tot_asserted = tot_exempt = tot_equal = 0
    per_sample = []
    for b, (img, ids, box) in enumerate(FD.case_inputs(name)):
        (_, _, _, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids.to(dev), None, None, None, None, [img.to(dev).to(odt)], [box],
                                                                             input_ids_host=ids)
        S = int(g[f"s{b}_S"])
        assert embeds.shape[1] == S                                   # integer side: -200 -> G^2 rows, -300 -> 1 row
        e_proj, e_rows = FW.vs_pin(embeds[0].float().cpu(), g, f"s{b}_embeds")
        ref_ids = g[f"s{b}_ids"].tolist()
        model._ensure_kv((S + n + 63) // 64 + 2)
        seq = SequenceState()
        lo = pair_lo(embeds)
        logits = llama_forward(llama, model.kv, [seq], embeds[0], [S], embeds_lo=None if lo is None else lo[0])
        st = DecodeState(llama, model.kv, [seq], n)
        got_ids, rows = [], []
        for t in range(n):                                            # teacher-forced: every step sees the reference's previous id
            rows.append(logits[0].float().cpu())
            got_ids.append(int(ops.argmax(logits)[0]))
            if t + 1 < n:
                st.feed(torch.tensor([ref_ids[t]], dtype=torch.int32, device=dev))
                logits = st.forward()
        model.kv.release(seq.pages)
        rows = torch.stack(rows)
        margins, rms = g[f"s{b}_margin"], g[f"s{b}_rms"]
        bounds = [3.0 * (2.0 ** 0.5) * tol * float(r) for r in rms]
        t5i, t5v = g[f"s{b}_top5_ids"], g[f"s{b}_top5_vals"]
        asserted = exempt = 0
        for t in range(n):
            if margins[t] > bounds[t]:
                assert got_ids[t] == ref_ids[t], (name, b, t, got_ids[t], ref_ids[t], float(margins[t]), bounds[t])
                asserted += 1
            else:
                cand = [int(i) for i, v in zip(t5i[t], t5v[t]) if t5v[t][0] - v <= bounds[t]]
                assert got_ids[t] in cand, (name, b, t, got_ids[t], cand)
                exempt += 1
        d_first = FW.rel(rows[0], g[f"s{b}_first_logits"])
        d_top5 = FW.rel(torch.gather(rows, 1, torch.as_tensor(t5i).long()), t5v)
        d_proj = FW.rel(rows.double() @ cases.fw_directions(rows.shape[-1]), g[f"s{b}_proj"])
        # the drop-in surface, free-running: generate() on the same sample, compared up to the first step it may legitimately leave
        out = model.generate(ids.to(dev), images=[img.to(dev).to(odt)], regions=[box], do_sample=False, max_new_tokens=n, eos_token_id=-1)
        free = out[0, ids.shape[1]:].tolist()
        agree = next((t for t in range(n) if free[t] != ref_ids[t]), n)
        assert all(margins[t] <= bounds[t] for t in range(agree, min(agree + 1, n))), (name, b, agree, free, ref_ids)
        equal = sum(int(a == r) for a, r in zip(got_ids, ref_ids))
        per_sample.append({"S": S, "asserted": asserted, "exempt": exempt, "ids_equal": equal, "free_running_agree_steps": agree,
                           "embeds_vs_reference": e_rows, "first_logits_vs_reference": d_first, "top5_values_vs_reference": d_top5,
                           "proj_vs_reference": d_proj, "min_margin": float(margins.min()), "mean_bound": float(np.mean(bounds))})
        b_ = BOUNDS[op]
        assert e_rows <= b_["embeds"] and e_proj <= b_["embeds"], (b, e_rows, e_proj)
        assert d_first <= b_["last"] and d_top5 <= b_["rows"] and d_proj <= 2 * b_["proj"], (b, d_first, d_top5, d_proj)
        if precise == 3:
            assert d_first <= 1.0e-3, (b, d_first)
        tot_asserted, tot_exempt, tot_equal = tot_asserted + asserted, tot_exempt + exempt, tot_equal + equal
    total = n * len(per_sample)
    _note(f"{name}_{op}" + ("-precise3" if precise == 3 else ""), {"operand": op, "case": name, "prefill_precise_level": precise, "image_size": FD.CASES[name]["image"], "steps_per_sample": n,
                           "ids_compared": total, "ids_asserted_exact": tot_asserted, "ids_exempt_near_tie": tot_exempt,
                           "ids_equal": tot_equal, "samples": per_sample})
    # floor: at least 60 % (bf16) / 85 % (fp16) of the reference's ids are decidable at the build's noise bound, and asserted exact
    assert tot_asserted >= (0.5 if op == "bf16" else 0.8) * total, (tot_asserted, tot_exempt)
    assert tot_equal >= 0.85 * total, (tot_equal, total)
