"""Parity of the COMPOSITE operators at the widths the benchmark runs: vt_llama_forward at H = 4096 / I = 11008 /
32 heads, vt_vit_forward at ViT-L/14, 336 px, T = 8, vt_projector_forward 1024 -> 4096 and vt_region_forward (1024 -> 4096) on
the 24 x 24 grid -- weights and inputs drawn exactly as bench.py draws them (N(0, 0.02^2), zero biases, unit norm gains) -- against

  (1) the oracle evaluated live on the host in fp32 and in bf16-storage-emulation mode, on the FULL tensors;
  (2) outputs of the REFERENCE's own modules on the same seeds (tests/golden/fullwidth.npz, written by
      tests/golden/make_golden.gen_fullwidth in the build container): projections of every row + whole rows + top-5 ids.

The float contract these tests state and measure:
  * HIP is no farther from fp32 (oracle == reference to 2e-5, tests/test_oracle_fullwidth.py) than the oracle's emulation of the
    kernels' bf16 storage points is (x 1.25 + 2e-4) -- measured: equal to three digits everywhere;
  * HIP is within FW_TOL_EMU = 2e-3 of that emulation wherever no softmax weight is rounded (projector, region: 1e-4 .. 2e-4) and
    for the ViT layers (1.3e-3 after two layers), and within FW_TOL_EMU_DECODER = 8.6e-3 for decoder layers at H = 4096 (measured 4.1e-3 /
    5.7e-3). Every operator of a decoder layer run on the emulation's own inputs is within 1e-3 of it (tests/test_gpu_parity_ops.py:
    <= 5.2e-5, flash attention 6.7e-4 since round 3 packs the softmax weights to fp16 -- round 2, bf16 P against the running maximum:
    2.0e-3); what a whole layer adds on top (3.3e-3) is rounding FLIPS: a 1e-7 difference in an fp32 sum moves the next bf16 store by a
    whole ulp, and o_proj / RMSNorm / the MLP amplify it -- the same size as a fraction of the emulation's own distance from fp32
    (1.0..1.3e-2), in an independent direction. tests/test_oracle_fullwidth.py::test_storage_emulation_is_defined_only_up_to_rounding_flips
    measures that floor on the emulation alone: a ONE-ulp (1e-7) perturbation of the fp32 input rows moves the emulated two-layer logits
    by 3.2e-3 in bf16 and 7.3e-4 in fp16 (1.9e-6 without emulation).
north_star's 1e-3 against an fp32 reference is below what ONE bf16 store leaves (2^-9 relative per element, ~1.7e-3 rel-L2): it is
met per operator against the emulation and reported -- not asserted -- for the chains.
Integer outputs (region cell masks and counts) are bit-exact against the reference. Measured numbers are printed (pytest -s) and
written to $VT_PARITY_REPORT (-> profiles/r3_parity_fullwidth.json).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import vitron_oracle as O
from tests import fullwidth_util as FW
from tests.golden import cases

pytestmark = pytest.mark.gpu

FW_TOL_EMU = 2e-3            # HIP vs emulating oracle, whole tensors
FW_TOL_EMU_DECODER = 8.6e-3  # decoder layers: rounding flips of the bf16 chains (see above); measured 4.1e-3 (1 layer) / 5.7e-3 (2 layers), x 1.5
# The fp16-operand build (round 4; libvitron_hip_f16.so, the reference's own inference dtype): one store leaves 2^-12 per element, so
# the chains are compared with the REFERENCE / plain fp32 at north_star's scale. Bounds = round-4 measurements x 1.5
# (profiles/r4_parity_fullwidth.json); the bf16 bounds above are unchanged.
FP16_TOL_VS_FP32_DECODER = 2.4e-3   # logits / hidden of 1-2 decoder layers at H = 4096 vs fp32
FP16_TOL_VS_FP32_TOWER = 6e-4       # 1-2 ViT-L layers, projector, region extractor vs fp32
OPERANDS = ["bf16", "fp16"]
REPORT = {}


@pytest.fixture(scope="module")
def dev():
    from vitron_amd import _lib
    _lib.load()
    torch.set_num_threads(min(32, os.cpu_count() or 8))      # PyTorch's CPU GEMMs are slower on all 256 hardware threads of the GPU box than on 16-32
    return torch.device("cuda:0")


def f32(sd):
    return {k: v.float() for k, v in sd.items()}


def _note(name, **kw):
    REPORT[name] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in kw.items()}
    print(f"[parity-fullwidth] {name}: " + json.dumps(REPORT[name]), flush=True)
    out = os.environ.get("VT_PARITY_REPORT")
    if out:
        with open(out, "w") as f:
            json.dump(REPORT, f, indent=1)


@pytest.mark.parametrize("op", OPERANDS)
@pytest.mark.parametrize("name", list(FW.ALL_LLAMA))      # S = 1088 / 2048 (336 px) and S = 768 / 2560 (the reference-native 224 px)
def test_decoder_prefill_at_7b_width_vs_oracle_and_reference(dev, name, op):
    from vitron_amd.engine import PackedLlama, PagedKVCache, SequenceState, llama_forward
    odt, emu, _ = FW.operand(op)
    g = FW.golden_of(name)
    cfg, sd, x = FW.llama_case(name)
    S = x.shape[0]
    llama = PackedLlama(sd, cfg, dev, dtype=odt)
    kv = PagedKVCache(llama, (S + 63) // 64 + 1)
    seq = SequenceState()
    logits, hidden = llama_forward(llama, kv, [seq], x.to(dev).to(odt), [S], logit_rows=list(range(S)), return_hidden=True)
    logits, hidden = logits.float().cpu(), hidden.float().cpu()
    (l32, h32), (lem, hem) = FW.oracle_llama(name, False), FW.oracle_llama(name, emu)
    l32, h32, lem, hem = l32.unsqueeze(0), h32.unsqueeze(0), lem.unsqueeze(0), hem.unsqueeze(0)
    d_emu, d_f32, emu_f32 = FW.rel(logits, lem[0]), FW.rel(logits, l32[0]), FW.rel(lem[0], l32[0])
    h_emu, h_f32, hemu_f32 = FW.rel(hidden, hem[0]), FW.rel(hidden, h32[0]), FW.rel(hem[0], h32[0])
    ref_proj, ref_rows = FW.vs_pin(logits, g, f"llama_{name}_logits")
    top1, top5 = FW.topk_agreement(logits, g, f"llama_{name}_logits")
    top1_emu, top5_emu = FW.topk_agreement(lem[0], g, f"llama_{name}_logits")
    _note(f"llama_{name}_{op}", rows=S, layers=cfg["num_hidden_layers"], logits_vs_emulation=d_emu, logits_vs_fp32=d_f32,
          emulation_vs_fp32=emu_f32, hidden_vs_emulation=h_emu, hidden_vs_fp32=h_f32, hidden_emulation_vs_fp32=hemu_f32,
          logits_vs_reference_rows=ref_rows, logits_vs_reference_proj=ref_proj, top1_vs_reference=top1, top5_overlap_vs_reference=top5,
          top1_of_emulation=top1_emu, top5_overlap_of_emulation=top5_emu)
    if op == "fp16":     # against plain fp32 and the reference's own rows, at north_star's scale
        assert d_f32 <= FP16_TOL_VS_FP32_DECODER and h_f32 <= FP16_TOL_VS_FP32_DECODER and ref_rows <= FP16_TOL_VS_FP32_DECODER, (d_f32, h_f32, ref_rows)
    assert d_emu <= FW_TOL_EMU_DECODER and h_emu <= FW_TOL_EMU_DECODER, (d_emu, h_emu)
    assert d_f32 <= 1.25 * emu_f32 + 2e-4 and h_f32 <= 1.25 * hemu_f32 + 2e-4, (d_f32, emu_f32, h_f32, hemu_f32)
    assert ref_rows <= 1.25 * emu_f32 + 2e-4, (ref_rows, emu_f32)            # against the reference's own logits rows
    assert top1 >= top1_emu - 0.01 and top5 >= top5_emu - 0.01, (top1, top1_emu, top5, top5_emu)


PRECISE_TOL_VS_FP32 = 1.0e-3     # north_star's bound, asserted: 1-2 decoder layers at H = 4096 in precise_qk mode vs fp32 and vs the reference's rows
PRECISE_TOL_VS_EMU = 1.2e-3      # vs the oracle's emulation of the mode's storage points: with the q / k path gone what is left on both sides is
                                 # independent rounding (P against the running vs the global maximum, accumulation order): measured 7.4e-4 .. 9.5e-4,
                                 # the same size as either side's distance from fp32 -- a loose regression net, the fp32 bound above is the contract


PRECISE2_TOL_VS_FP32 = 6e-4      # level 2 (every GEMM A operand a pair): what is left is V^T / P in fp16 (CPU estimate 3.5e-4 .. 4.5e-4)


@pytest.mark.parametrize("level", [1, 2], ids=["precise_qk", "precise2"])
@pytest.mark.parametrize("name", list(FW.ALL_LLAMA))
def test_decoder_prefill_precise_qk_fp16(dev, name, level):
    """vt_llama_model.precise_qk (round 5): q / k and the norm output that feeds their projection travel as hi + lo operand pairs
    (A_hi.W^T + A_lo.W^T into fp32, rotary in fp32, scores = K_hi.(Q_hi + Q_lo)^T + K_lo.Q_hi^T). fp16 build; against plain fp32,
    the reference's own rows and the oracle's emulation of the mode. Also: the same prompt prefetched in two chunks (the second
    chunk's attention sees the first chunk's keys as K_hi only) stays inside the bound, and the mode leaves the K / V^T pages a
    standard pass can decode from."""
    from vitron_amd.engine import PackedLlama, PagedKVCache, SequenceState, llama_forward
    odt, emu, _ = FW.operand("fp16")
    g = FW.golden_of(name)
    cfg, sd, x = FW.llama_case(name)
    S = x.shape[0]
    llama = PackedLlama(sd, dict(cfg, precise=level), dev, dtype=odt)
    assert llama.model.precise_qk == level
    tol32 = PRECISE_TOL_VS_FP32 if level == 1 else PRECISE2_TOL_VS_FP32
    kv = PagedKVCache(llama, 2 * ((S + 63) // 64 + 2))
    seq = SequenceState()
    xd = x.to(dev).to(odt)
    logits, hidden = llama_forward(llama, kv, [seq], xd, [S], logit_rows=list(range(S)), return_hidden=True)
    logits, hidden = logits.float().cpu(), hidden.float().cpu()
    (l32, h32), (lem, hem) = FW.oracle_llama(name, False), FW.oracle_llama(name, emu, precise_qk=level)
    (lstd, _) = FW.oracle_llama(name, emu)
    d_f32, h_f32, d_emu, h_emu = FW.rel(logits, l32), FW.rel(hidden, h32), FW.rel(logits, lem), FW.rel(hidden, hem)
    emu_f32, std_f32 = FW.rel(lem, l32), FW.rel(lstd, l32)
    ref_proj, ref_rows = FW.vs_pin(logits, g, f"llama_{name}_logits")
    top1, top5 = FW.topk_agreement(logits, g, f"llama_{name}_logits")
    # two chunks: rows [0, S1) then [S1, S) on the same sequence
    S1 = (S // 2 // 64) * 64 + 24                      # the second chunk starts inside a page
    seq2 = SequenceState()
    llama_forward(llama, kv, [seq2], xd[:S1], [S1], logit_rows=[S1 - 1])
    lg2 = llama_forward(llama, kv, [seq2], xd[S1:], [S - S1], logit_rows=list(range(S - S1))).float().cpu()
    d_chunk = FW.rel(lg2, l32[S1:])
    # standard mode decoding on top of the pages the precise prefill wrote
    llama.set_precise_qk(False)
    step = llama_forward(llama, kv, [seq], xd[-1:], [1]).float().cpu()      # one more (repeated) row: runs the decode kernels on the pages
    assert torch.isfinite(step).all()
    _note(f"llama_{name}_fp16_precise{level}", rows=S, layers=cfg["num_hidden_layers"], logits_vs_fp32=d_f32, hidden_vs_fp32=h_f32,
          logits_vs_emulation=d_emu, hidden_vs_emulation=h_emu, emulation_vs_fp32=emu_f32, standard_emulation_vs_fp32=std_f32,
          logits_vs_reference_rows=ref_rows, logits_vs_reference_proj=ref_proj, top1_vs_reference=top1, top5_overlap_vs_reference=top5,
          two_chunk_logits_vs_fp32=d_chunk)
    assert d_f32 <= tol32 and h_f32 <= tol32 and ref_rows <= tol32, (d_f32, h_f32, ref_rows)
    assert d_emu <= PRECISE_TOL_VS_EMU and h_emu <= PRECISE_TOL_VS_EMU, (d_emu, h_emu)
    assert d_f32 <= 0.75 * std_f32, (d_f32, std_f32)              # the mode buys what the storage-point analysis says (x 0.55 measured on the CPU)
    assert d_chunk <= 1.15 * PRECISE_TOL_VS_FP32, d_chunk
    assert llama.model.precise_qk == 0


@pytest.mark.parametrize("op", OPERANDS)
@pytest.mark.parametrize("name", ["video336", "image336", "video224", "image224"])   # 224: N = 257, 2056 / 514 token rows
def test_towers_at_vit_l_336_vs_oracle_and_reference(dev, name, op):
    from vitron_amd.engine import PackedVit
    odt, emu, rnd = FW.operand(op)
    g = FW.golden_of(name)
    cfg, sd, x = FW.vit_case(name)
    for nl in (1, cases.FW_VIT_LAYERS):
        vit = PackedVit(sd, cfg, dev, select_layer=nl, dtype=odt)
        feats, hidden = vit.forward(x.to(dev).to(odt), return_hidden=True)
        hidden = hidden.float().cpu().reshape(-1, 1024)
        with torch.no_grad():
            h32 = O.vit_forward(f32(sd), cfg, x, num_layers=nl).reshape(-1, 1024)
            hem = O.vit_forward(f32(sd), cfg, x, num_layers=nl, emulate_bf16=emu).reshape(-1, 1024)
        d_emu, d_f32, emu_f32 = FW.rel(hidden, hem), FW.rel(hidden, h32), FW.rel(hem, h32)
        ref_proj, ref_rows = FW.vs_pin(hidden, g, f"vit_{name}_hidden_{nl}")
        # feature_select: patch tokens (CLS dropped) of this hidden state, bf16
        N = hidden.shape[0] // (x.shape[0] * (x.shape[2] if x.dim() == 5 else 1))
        patch = hem.reshape(-1, N, 1024)[:, 1:].reshape(-1, 1024)
        d_feat = FW.rel(feats.float().cpu().reshape(-1, 1024), rnd(patch))
        _note(f"vit_{name}_layers{nl}_{op}", rows=hidden.shape[0], vs_emulation=d_emu, vs_fp32=d_f32, emulation_vs_fp32=emu_f32,
              vs_reference_rows=ref_rows, vs_reference_proj=ref_proj, features_vs_emulation=d_feat)
        # one WHOLE tower layer stays inside north_star's 1e-3 of the emulation (measured 5.7e-4 video / 4.4e-4 image); two layers 1.24e-3
        if op == "fp16":
            assert d_f32 <= FP16_TOL_VS_FP32_TOWER and ref_rows <= FP16_TOL_VS_FP32_TOWER, (nl, d_f32, ref_rows)
        assert d_emu <= (8.6e-4 if nl == 1 else 1.9e-3) and d_feat <= (2.2e-3 if nl == 1 else 3.3e-3), (nl, d_emu, d_feat)
        assert d_f32 <= 1.25 * emu_f32 + 2e-4 and ref_rows <= 1.25 * emu_f32 + 3e-4, (nl, d_f32, ref_rows, emu_f32)


TOWER_PRECISE_TOL = 6e-5            # towers in precise level 2 vs fp32 / the reference, BOTH builds (measured 1.0e-5 .. 2.9e-5), x 2


@pytest.mark.parametrize("op", OPERANDS)
@pytest.mark.parametrize("name", ["video336", "image336", "video224", "image224"])
def test_towers_precise_level_2_vs_reference(dev, name, op):
    """vt_vit_model.precise = 2: every GEMM A operand of the tower an operand pair -- norm outputs, q and k through the scores (the decoder's
    precise attention kernels at head_dim 64), v from fp32 straight into the V^T tiles, attention outputs, the temporal attention in fp32,
    GELU outputs. What is left is V^T / P in fp16 and the pairs' 2^-17 (bf16) / 2^-23 (fp16): the hidden state is within 6e-5 of the fp32
    oracle and of the REFERENCE's stored outputs in both operand builds -- the tower no longer contributes to the end-to-end distance."""
    from vitron_amd.engine import PackedVit, pair_lo
    odt, emu, _ = FW.operand(op)
    g = FW.golden_of(name)
    cfg, sd, x = FW.vit_case(name)
    nl = cases.FW_VIT_LAYERS
    vit = PackedVit(sd, cfg, dev, select_layer=nl, dtype=odt)
    _, h_std = vit.forward(x.to(dev).to(odt), return_hidden=True)
    vit.set_precise(2)
    feats, hidden = vit.forward(x.to(dev).to(odt), return_hidden=True)
    lo = pair_lo(feats)
    assert lo is not None and lo.shape == feats.shape
    hidden, h_std = hidden.float().cpu().reshape(-1, 1024), h_std.float().cpu().reshape(-1, 1024)
    with torch.no_grad():
        h32 = O.vit_forward(f32(sd), cfg, x, num_layers=nl).reshape(-1, 1024)
        hem = O.vit_forward(f32(sd), cfg, x, num_layers=nl, emulate_bf16=emu, precise=2).reshape(-1, 1024)
    d_f32, std_f32, d_emu = FW.rel(hidden, h32), FW.rel(h_std, h32), FW.rel(hidden, hem)
    ref_proj, ref_rows = FW.vs_pin(hidden, g, f"vit_{name}_hidden_{nl}")
    assert d_emu <= 3e-5, d_emu                  # against the oracle's emulation of the mode's own storage points
    # the features leave as a pair whose sum is the fp32 hidden state of the patch tokens
    N = hidden.shape[0] // (x.shape[0] * (x.shape[2] if x.dim() == 5 else 1))
    patch = hidden.reshape(-1, N, 1024)[:, 1:].reshape(-1, 1024)
    d_pair = FW.rel((feats.float() + lo.float()).cpu().reshape(-1, 1024), patch)
    _note(f"vit_{name}_layers{nl}_{op}-precise2", rows=hidden.shape[0], vs_fp32=d_f32, standard_mode_vs_fp32=std_f32, vs_reference_rows=ref_rows,
          vs_reference_proj=ref_proj, feature_pair_vs_hidden=d_pair)
    assert d_f32 <= TOWER_PRECISE_TOL and ref_rows <= TOWER_PRECISE_TOL and ref_proj <= TOWER_PRECISE_TOL, (d_f32, ref_rows, ref_proj)
    assert d_f32 <= 0.2 * std_f32, (d_f32, std_f32)
    assert d_pair <= (3e-5 if op == "bf16" else 1e-6), d_pair
    vit.set_precise(0)
    _, h_again = vit.forward(x.to(dev).to(odt), return_hidden=True)
    assert torch.equal(h_again.float().cpu().reshape(-1, 1024), h_std)         # back to the standard kernels, bit for bit


@pytest.mark.parametrize("op", OPERANDS)
@pytest.mark.parametrize("name", ["video336", "image224"])
def test_towers_precise_level_1_on_the_mx_pipe(dev, name, op):
    """vt_vit_model.precise = 1 at ViT-L width (round 6: the towers' share of the model's precise level 3): the spatial MLP's two products run as
    ONE launch each that adds the MX-FP4 product of the A operand's rounding remainder (LayerNorm and the GELU epilogue write the remainder's image),
    the attention paths stay standard, the features leave as a pair. Against the oracle's emulation of exactly these storage points, against
    fp32 and against the REFERENCE's stored rows: most of the standard mode's distance is gone (what is left is the attention paths')."""
    from vitron_amd.engine import PackedVit, pair_lo
    odt, emu, _ = FW.operand(op)
    g = FW.golden_of(name)
    cfg, sd, x = FW.vit_case(name)
    nl = cases.FW_VIT_LAYERS
    vit = PackedVit(sd, cfg, dev, select_layer=nl, dtype=odt)
    _, h_std = vit.forward(x.to(dev).to(odt), return_hidden=True)
    vit.set_precise(3)                                   # the MODEL's level: 3 -> tower level 1
    assert vit.model.precise == 1 and vit.layers[0].w14 and vit.layers[0].w24
    feats, hidden = vit.forward(x.to(dev).to(odt), return_hidden=True)
    lo = pair_lo(feats)
    assert lo is not None and lo.shape == feats.shape
    hidden, h_std = hidden.float().cpu().reshape(-1, 1024), h_std.float().cpu().reshape(-1, 1024)
    with torch.no_grad():
        h32 = O.vit_forward(f32(sd), cfg, x, num_layers=nl).reshape(-1, 1024)
        hem = O.vit_forward(f32(sd), cfg, x, num_layers=nl, emulate_bf16=emu, precise=1).reshape(-1, 1024)
    d_f32, std_f32, d_emu, emu_f32 = FW.rel(hidden, h32), FW.rel(h_std, h32), FW.rel(hidden, hem), FW.rel(hem, h32)
    ref_proj, ref_rows = FW.vs_pin(hidden, g, f"vit_{name}_hidden_{nl}")
    _note(f"vit_{name}_layers{nl}_{op}-precise1", rows=hidden.shape[0], vs_fp32=d_f32, standard_mode_vs_fp32=std_f32, vs_emulation=d_emu,
          emulation_vs_fp32=emu_f32, vs_reference_rows=ref_rows, vs_reference_proj=ref_proj)
    assert d_f32 <= 0.6 * std_f32, (d_f32, std_f32)
    assert d_f32 <= 1.25 * emu_f32 + 1e-4 and ref_rows <= 1.25 * emu_f32 + 1e-4, (d_f32, ref_rows, emu_f32)
    vit.set_precise(0)
    _, h_again = vit.forward(x.to(dev).to(odt), return_hidden=True)
    assert torch.equal(h_again.float().cpu().reshape(-1, 1024), h_std)


@pytest.mark.parametrize("which", ["336", "224"])
@pytest.mark.parametrize("op", OPERANDS)
def test_projector_and_region_at_full_width_vs_oracle_and_reference(dev, op, which):
    """which = "336": the 24 x 24 grid of the BASELINE image size on a 224 and a 336 canvas; "224": RegionExtractor exactly as the
    reference ships it (224 canvas, layer.py:60) on the 16 x 16 grid of the 224 px tower, projector on 2 x 256 rows."""
    from vitron_amd.engine import PackedProjector, PackedRegion
    odt, emu, _ = FW.operand(op)
    g = FW.golden(which)
    sd, x = FW.projector_case(which)
    out = PackedProjector(sd, dev, dtype=odt).forward(x.to(dev).to(odt)).float().cpu()
    with torch.no_grad():
        o32, oem = O.projector_forward(f32(sd), x), O.projector_forward(f32(sd), x, emulate_bf16=emu)
    d_emu, d_f32, emu_f32 = FW.rel(out, oem), FW.rel(out, o32), FW.rel(oem, o32)
    ref_proj, ref_rows = FW.vs_pin(out, g, "projector")
    _note(f"projector_{which}_{op}", rows=x.shape[0], vs_emulation=d_emu, vs_fp32=d_f32, emulation_vs_fp32=emu_f32, vs_reference_rows=ref_rows)
    assert d_emu <= 1e-3 and d_f32 <= 1.25 * emu_f32 + 2e-4 and ref_rows <= 1.25 * emu_f32 + 3e-4, (d_emu, d_f32, ref_rows, emu_f32)
    if op == "fp16":
        assert d_f32 <= FP16_TOL_VS_FP32_TOWER and ref_rows <= FP16_TOL_VS_FP32_TOWER, (d_f32, ref_rows)
    for canvas, grid in (((224, 24), (336, 24)) if which == "336" else ((224, 16),)):
        sd, feats, boxes = FW.region_case(canvas, grid)
        reg = PackedRegion(sd, dev, image_size=canvas, dtype=odt)
        out, cells, count = reg.forward(feats.to(dev).to(odt), boxes, return_mask=True)
        assert np.array_equal(cells.cpu().numpy(), g[f"region_c{canvas}_cells"])        # bit exact vs the REFERENCE at G = 24 / 16
        assert count.cpu().tolist() == g[f"region_c{canvas}_cells"].sum(-1).tolist()
        with torch.no_grad():        # (box coordinates reach the LocationEncoder in fp32 since round 4: no rounding to emulate)
            oem, _, _ = O.region_forward(f32(sd), feats, boxes, canvas, emu)
            o32, _, _ = O.region_forward(f32(sd), feats, boxes, canvas)
        got = out[:, 0].float().cpu()
        d_emu, d_f32, emu_f32 = FW.rel(got, oem[:, 0]), FW.rel(got, o32[:, 0]), FW.rel(oem[:, 0], o32[:, 0])
        d_ref = FW.rel(got, g[f"region_c{canvas}_out"])
        _note(f"region_canvas{canvas}_grid{grid}_{op}", boxes=len(boxes), vs_emulation=d_emu, vs_fp32=d_f32, emulation_vs_fp32=emu_f32, vs_reference=d_ref)
        assert d_emu <= FW_TOL_EMU and d_f32 <= 1.25 * emu_f32 + 5e-4 and d_ref <= 1.25 * emu_f32 + 5e-4, (canvas, d_emu, d_f32, d_ref, emu_f32)
        if op == "fp16":
            assert d_f32 <= FP16_TOL_VS_FP32_TOWER and d_ref <= FP16_TOL_VS_FP32_TOWER, (canvas, d_f32, d_ref)
