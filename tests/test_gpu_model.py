"""End-to-end parity of the drop-in surface (vitron_amd.model.LlavaLlamaForCausalLM, towers, projector, region
extractor) on the MI355X against (a) the golden vectors produced by the REFERENCE's own modules (pure fp32) and
(b) the CPU oracle in bf16-storage emulation mode.

Tolerances (rel-L2), and why:
  TOL = 1e-3        north_star's bar. Holds per kernel (tests/test_gpu_kernels.py) and for SHALLOW chains against the
                    emulating oracle (projector, region extractor, one ViT layer, text-only decoder prefill).
  TOL_FP32 = 4.7e-3 shallow chains (towers up to three layers, projector, region extractor) against the REFERENCE's fp32 goldens: 1.5 x the
                    worst measured (3.1e-3; the emulation's own distance from fp32 is the same size).
  TOL_DEEP = 2.6e-2 deep chains (ViT -> projector -> splice -> decoder); = 1.5 x the worst measured (round 3, [parity-tiny] lines in
                    profiles/r3_parity_lines.txt: 1.73e-2 vs the reference, 1.58e-2 vs the emulation). bf16 storage of GEMM operands (eps 2^-8) puts
                    even the emulating oracle ~1.5e-2 away from the fp32 reference on these test weights, and tiny fp32
                    summation-order differences flip bf16 roundings, so HIP-vs-emulation drifts to the same noise floor.
                    What IS asserted for deep chains: the HIP result is no farther from the REFERENCE's fp32 output than
                    the emulating oracle is (x1.25 + 1e-3), i.e. the whole residual is the bf16 storage format, and the
                    greedy tokens agree.
Integer outputs (cell masks, spliced layout) are bit exact."""
import os

import numpy as np
import pytest
import torch

from oracle import vitron_oracle as O
from tests.golden import cases
from tests.util import f32, rel_l2
from vitron_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-3
TOL_FP32 = 4.7e-3
TOL_DEEP = 2.6e-2
TOL_SHALLOW = 3e-3


def no_worse_than_emulation(hip, emu, ref):
    return rel_l2(hip, ref) <= 1.25 * rel_l2(emu, ref) + 1e-3


@pytest.fixture(scope="module")
def dev():
    from vitron_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _states():
    return {
        "image_tower": synth.vit_state(cases.VIT_IMAGE, synth.make_generator(cases.SEED_VIT), **cases.VIT_INIT),
        "video_tower": synth.vit_state(cases.VIT_VIDEO, synth.make_generator(cases.SEED_VIT), **cases.VIT_INIT),
        "projector": synth.projector_state(cases.MM_HIDDEN, cases.LLM["hidden_size"], synth.make_generator(cases.SEED_PROJ), **cases.MLP_INIT),
        "region": synth.region_state(cases.MM_HIDDEN, cases.LLM["hidden_size"], synth.make_generator(cases.SEED_REGION), **cases.MLP_INIT),
        "llama": synth.llama_state(cases.LLM, synth.make_generator(cases.SEED_LLM), **cases.LLM_INIT),
    }


@pytest.fixture(scope="module")
def model(dev):
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM
    st = _states()
    cfg = LlavaConfig(**cases.LLM, mm_hidden_size=cases.MM_HIDDEN, mm_image_tower="golden/LanguageBind_Image",
                      mm_video_tower="golden/LanguageBind_Video_merge")
    m = LlavaLlamaForCausalLM(cfg)
    m.get_image_tower().load_state(cases.VIT_IMAGE, st["image_tower"])
    m.get_video_tower().load_state(cases.VIT_VIDEO, st["video_tower"])
    sd = dict(st["llama"])
    sd.update({"model.mm_projector." + k: v for k, v in st["projector"].items()})
    sd.update({"model.region_extractor." + k: v for k, v in st["region"].items()})
    m.load_state_dict(sd)
    return m.to(dev)


CFGS = {"image": cases.VIT_IMAGE, "video": cases.VIT_VIDEO, "llama": cases.LLM}


@pytest.mark.parametrize("name,cfg,shape", [("video", cases.VIT_VIDEO, (2, 3, 4, 56, 56)), ("image", cases.VIT_IMAGE, (3, 3, 56, 56))])
def test_vit_tower(dev, name, cfg, shape):
    from vitron_amd.engine import PackedVit
    g = np.load(os.path.join(G, "vit.npz"))
    sd = synth.vit_state(cfg, synth.make_generator(cases.SEED_VIT), **cases.VIT_INIT)
    x = cases.pixels(shape, cases.SEED_PIX)
    for sel in (-2, -1, 1):
        vit = PackedVit(sd, cfg, dev, select_layer=sel)
        feats, hidden = vit.forward(x.to(dev).bfloat16(), return_hidden=True)
        nl = vit.run_layers
        emu = O.vit_forward(f32(sd), cfg, x, nl, emulate_bf16=True)
        ref = torch.as_tensor(g[f"{name}_hidden_{nl}"])
        assert rel_l2(hidden, emu) <= (TOL if nl <= 1 else 3.8e-3), (sel, "vs emulating oracle")      # measured 2.5e-3 at depth 3
        assert rel_l2(hidden, ref) <= TOL_FP32 and no_worse_than_emulation(hidden, emu, ref), (sel, "vs reference fp32")
        assert rel_l2(feats.float().reshape(-1, 16, 128), O.bf16_round(emu[:, 1:])) <= (TOL if nl <= 1 else 5e-3)
        assert feats.shape == ((2, 4, 16, 128) if name == "video" else (3, 16, 128))
    assert rel_l2(vit.forward(x.to(dev)).float(), feats.float()) == 0.0   # fp32 pixels take the same path


@pytest.mark.parametrize("name", list(cases.VIT_B_CASES))
def test_vit_tower_second_shape(dev, name):
    """The second pinned tower shape (quick_gelu, 8 frames -> the T = 8 temporal kernel, 70 px -> 5 x 5 patches, 26 tokens):
    all layers on the device against the reference's hidden states and the bf16-emulating oracle."""
    from vitron_amd.engine import PackedVit
    g = np.load(os.path.join(G, "vit_b.npz"))
    cfg, shape = cases.VIT_B_CASES[name]
    sd = synth.vit_state(cfg, synth.make_generator(cases.SEED_VIT + 1), **cases.VIT_INIT)
    x = cases.pixels(shape, cases.SEED_PIX + 9)
    for sel in (-1, -2):
        vit = PackedVit(sd, cfg, dev, select_layer=sel)
        feats, hidden = vit.forward(x.to(dev).bfloat16(), return_hidden=True)
        nl = vit.run_layers
        emu = O.vit_forward(f32(sd), cfg, x, nl, emulate_bf16=True)
        ref = torch.as_tensor(g[f"{name}_hidden_{nl}"])
        assert rel_l2(hidden, emu) <= (TOL if nl <= 1 else 3.8e-3), (sel, "vs emulating oracle")      # measured 2.5e-3 at depth 3
        assert rel_l2(hidden, ref) <= TOL_FP32 and no_worse_than_emulation(hidden, emu, ref), (sel, "vs reference fp32")
        assert feats.shape == ((1, 8, 25, 128) if name == "video_b" else (2, 25, 128))


@pytest.mark.parametrize("name", list(cases.VIT_TMLP_CASES))
def test_vit_image_tower_with_time_attention_and_temporal_mlp(dev, name):
    """The IMAGE tower file's add_time_attn variant (reference image/modeling_image.py:74-84,105-134: temporal attention AND a temporal
    MLP in every layer) through the tower wrapper -- 4-frame clips, and num_frames = 1 (the embedding add is skipped, the block
    degenerates to out_proj(v_proj(LN(x))) + MLP): all layers against the reference's hidden states and the emulating oracle."""
    from types import SimpleNamespace
    from vitron_amd.engine import PackedVit
    from vitron_amd.model.multimodal_encoder.languagebind import LanguageBindImageTower
    g = np.load(os.path.join(G, "vit_tmlp.npz"))
    cfg, shape = cases.VIT_TMLP_CASES[name]
    sd = synth.vit_state(cfg, synth.make_generator(cases.SEED_VIT + 2), **cases.VIT_INIT)
    x = cases.pixels(shape, cases.SEED_PIX + 2)
    for sel in (-1, -2):
        vit = PackedVit(sd, cfg, dev, select_layer=sel)
        feats, hidden = vit.forward(x.to(dev).bfloat16(), return_hidden=True)
        nl = vit.run_layers
        emu = O.vit_forward(f32(sd), cfg, x, nl, emulate_bf16=True)
        ref = torch.as_tensor(g[f"{name}_hidden_{nl}"])
        assert rel_l2(hidden, emu) <= 5e-3, (sel, rel_l2(hidden, emu), "vs emulating oracle")
        assert rel_l2(hidden, ref) <= 1.4 * TOL_FP32 and no_worse_than_emulation(hidden, emu, ref), (sel, rel_l2(hidden, ref), "vs reference fp32")   # two more storage points per layer than the video tower
        # precise level 2 (operand pairs through the temporal attention, the temporal MLP, the spatial attention and the MLP): the bf16
        # build lands within 1e-4 of the REFERENCE's fp32 hidden state (standard mode: ~1e-2 at this init)
        vit.set_precise(2)
        _, hp = vit.forward(x.to(dev).bfloat16(), return_hidden=True)
        vit.set_precise(0)
        assert rel_l2(hp, ref) <= 1e-4 and rel_l2(hp, ref) <= 0.05 * rel_l2(hidden, ref), (sel, rel_l2(hp, ref), rel_l2(hidden, ref))
    tower = LanguageBindImageTower("tmlp/LanguageBind_Image", SimpleNamespace(mm_vision_select_layer=-2), delay_load=True)
    tower.load_state(cfg, sd, dev)                                  # (round 4 refused these weights with NotImplementedError)
    f = tower(x.to(dev).bfloat16())
    assert f.shape[-2:] == (16, 128) and rel_l2(f.reshape(-1, 128), feats.reshape(-1, 128)) == 0.0


def test_projector_and_region(dev, model):
    g = np.load(os.path.join(G, "region_projector.npz"))
    st = _states()
    x = cases.features((37, cases.MM_HIDDEN), cases.SEED_FEATS + 1)
    y = model.get_model().mm_projector(x.to(dev).bfloat16())
    assert rel_l2(y.float(), O.projector_forward(f32(st["projector"]), x, True)) <= TOL
    assert rel_l2(y.float(), torch.as_tensor(g["projector_out"])) <= TOL_FP32
    for tag, (cin, cout, grid) in cases.REGION_CASES.items():
        feats = cases.features((len(cases.BOXES), grid * grid, cin), cases.SEED_FEATS)
        out, cells, count = model.get_region_extractor().packed.forward(feats.to(dev).bfloat16(), cases.BOXES, return_mask=True)
        assert np.array_equal(cells.cpu().numpy(), g[f"region_{tag}_cells"])        # bit exact vs the REFERENCE
        assert count.cpu().tolist() == g[f"region_{tag}_cells"].sum(-1).tolist()
        emu, _, _ = O.region_forward(f32(st["region"]), feats, cases.BOXES, 224, True)      # box coordinates stay fp32 (round 4)
        assert rel_l2(out.float(), emu) <= TOL
        assert rel_l2(out.float(), torch.as_tensor(g[f"region_{tag}_out"])) <= TOL_FP32


def test_f1_branches_clip_tower_and_mlp3x_projector(dev, tmp_path):
    """build_image_tower picks CLIPVisionTower for `openai*` names (reference multimodal_encoder/builder.py:12)
    and build_vision_projector loops N Linear layers for mlpNx_gelu (multimodal_projector/builder.py:39-46). Both against outputs of
    the REFERENCE's own classes (tests/golden/f1.npz) and the emulating oracle; the tower is loaded from a checkpoint DIRECTORY in
    the transformers layout (vision_model.* names, safetensors), as load_model() reads it."""
    import json
    from types import SimpleNamespace
    from safetensors.torch import save_file
    from vitron_amd.model.multimodal_encoder.builder import build_image_tower
    from vitron_amd.model.multimodal_encoder.clip_encoder import CLIPVisionTower
    from vitron_amd.model.multimodal_projector.builder import build_vision_projector
    g = np.load(os.path.join(G, "f1.npz"))
    cfg = cases.CLIP_TOWER
    sd = synth.vit_state(cfg, synth.make_generator(cases.SEED_VIT + 2), **cases.VIT_INIT)
    ck = tmp_path / "openai" / "clip-vit-tiny-patch14"
    ck.mkdir(parents=True)
    (ck / "config.json").write_text(json.dumps({"model_type": "clip_vision_model", **{k: v for k, v in cfg.items() if k not in ("add_time_attn", "num_frames")}}))
    save_file({"vision_model." + k: v.contiguous() for k, v in sd.items()}, str(ck / "model.safetensors"))
    x = cases.pixels(cases.CLIP_TOWER_SHAPE, cases.SEED_PIX + 21)
    emu = O.vit_forward(f32(sd), cfg, x, cfg["num_hidden_layers"] - 1, emulate_bf16=True)
    for feat in ("patch", "cls_patch"):
        args = SimpleNamespace(mm_image_tower="openai/clip-vit-tiny-patch14", mm_vision_select_layer=-2, mm_vision_select_feature=feat)
        t = build_image_tower(args, delay_load=True, cache_dir=str(tmp_path))
        assert isinstance(t, CLIPVisionTower) and not t.is_loaded and t.config.hidden_size == 128 and t.num_patches == 16
        t.load_model()
        t.to(dev)
        out = t(x.to(dev).bfloat16())
        ref = torch.as_tensor(g[f"clip_{feat}"])
        assert tuple(out.shape) == ref.shape and out.dtype == torch.bfloat16
        want = emu if feat == "cls_patch" else emu[:, 1:]
        assert rel_l2(out.float(), O.bf16_round(want)) <= TOL, feat
        assert rel_l2(out.float(), ref) <= TOL_FP32 and no_worse_than_emulation(out.float().cpu(), O.bf16_round(want), ref), feat
        lst = t([x[0].to(dev).bfloat16(), x[2].to(dev).bfloat16()])             # list input: one [1, P, D] tensor per image (:41-47)
        assert rel_l2(lst[1].float(), torch.as_tensor(g[f"clip_{feat}_list1"])) <= TOL_FP32 and lst[0].shape[0] == 1
    with pytest.raises(FileNotFoundError):
        build_image_tower(SimpleNamespace(mm_image_tower="laion/not-there", mm_vision_select_layer=-2), delay_load=True).load_model()
    with pytest.raises(ValueError):
        build_image_tower(SimpleNamespace(mm_image_tower="somewhere/else", mm_vision_select_layer=-2))
    # mlp3x_gelu
    H = cases.LLM["hidden_size"]
    gp = synth.make_generator(cases.SEED_PROJ + 3)
    psd = synth.projector_state(cases.MM_HIDDEN, H, gp, **cases.MLP_INIT)
    extra = synth.projector_state(H, H, gp, **cases.MLP_INIT)
    psd["4.weight"], psd["4.bias"] = extra["2.weight"], extra["2.bias"]
    pm = build_vision_projector(SimpleNamespace(mm_projector_type="mlp3x_gelu", mm_hidden_size=cases.MM_HIDDEN, hidden_size=H))
    pm.load_state_dict(psd)
    pm.to(dev)
    xp = cases.features((cases.PROJ3_ROWS, cases.MM_HIDDEN), cases.SEED_FEATS + 5)
    y = pm(xp.to(dev).bfloat16())
    assert rel_l2(y.float(), O.projector_forward(f32(psd), xp, True)) <= TOL
    assert rel_l2(y.float(), torch.as_tensor(g["proj3_out"])) <= TOL_FP32
    with pytest.raises(KeyError):
        build_vision_projector(SimpleNamespace(mm_projector_type="mlp4x_gelu", mm_hidden_size=cases.MM_HIDDEN, hidden_size=H)).load_state_dict(psd)


def test_encode_images_videos_api(dev, model):
    img = torch.stack([cases.pixels((3, 56, 56), cases.SEED_PIX + i) for i in range(2)]).to(dev).bfloat16()
    f, r = model.encode_images(img, [cases.BOXES[1], cases.BOXES[2]])
    assert f.shape == (2, 16, 256) and r.shape == (2, 1, 256)
    f2, z = model.encode_images(img, None)
    assert torch.equal(f, f2) and z.shape == f.shape and not z.any()                # zeros_like dummy (llava_arch.py:179-181)
    v = model.encode_videos(cases.pixels((1, 3, 4, 56, 56), 5).to(dev).bfloat16())
    assert v.shape == (1, 4, 16, 256)
    assert model.get_video_tower().config.num_frames == 4 and model.get_image_tower().num_patches == 16


@pytest.mark.parametrize("name", list(cases.glue_cases()))
def test_multimodal_prefill_logits(dev, model, name):
    g = np.load(os.path.join(G, "glue_llm.npz"))
    case = cases.glue_cases()[name]
    model.config.tokenizer_model_max_length = case.get("max_length")
    model.config.tokenizer_padding_side = case.get("padding_side", "right")
    ids = case["input_ids"].to(dev)
    am = None if case["attention_mask"] is None else case["attention_mask"].to(dev)
    images = [im.to(dev).bfloat16() for im in case["images"]]
    (_, pos, mask, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids, None, am, None, None, images, case["regions"])
    ref_e, ref_l, ref_m = g[f"{name}_embeds"], g[f"{name}_logits"], g[f"{name}_mask"]
    assert tuple(embeds.shape) == ref_e.shape                                        # spliced layout: exact
    host_mask = np.array(model._last_splice[0], dtype=np.int32)
    assert np.array_equal(host_mask, ref_m)
    w = {k: f32(v) for k, v in _states().items()}
    e_logits, e_embeds, e_mask, _ = O.multimodal_forward(w, CFGS, case["input_ids"], case["attention_mask"], case["images"],
                                                         case["regions"], case.get("max_length"), case.get("padding_side", "right"), True)
    assert rel_l2(embeds.float(), e_embeds) <= 5e-3
    assert rel_l2(embeds.float(), torch.as_tensor(ref_e)) <= 5e-3          # visual tokens are stored in bf16
    out = model(input_ids=ids, attention_mask=am, images=images, regions=case["regions"], use_cache=False)
    valid = torch.as_tensor(ref_m).bool()
    lg, rl = out.logits.cpu()[valid], torch.as_tensor(ref_l)[valid]
    shallow = name == "text_only"                                          # decoder only: short chain
    d_emu, d_ref = rel_l2(lg, e_logits[valid]), rel_l2(lg, rl)
    print(f"[parity-tiny] prefill_{name}: vs_emulation {d_emu:.3e} vs_reference {d_ref:.3e} emulation_vs_reference {rel_l2(e_logits[valid], rl):.3e}", flush=True)
    # shallow chain (two decoder layers): two bf16-storage chains that agree to ~1e-4 per operator still differ by the roundings
    # that flip between them (one bf16 ulp = 4e-3 on the element): measured 1.8e-3
    assert d_emu <= (TOL_SHALLOW if shallow else TOL_DEEP), "vs emulating oracle"
    assert d_ref <= TOL_DEEP and no_worse_than_emulation(lg, e_logits[valid], rl), "vs reference fp32"
    assert float((lg.argmax(-1) == rl.argmax(-1)).float().mean()) >= 0.9
    model.config.tokenizer_model_max_length = None
    model.config.tokenizer_padding_side = "right"


def test_greedy_generate_token_ids(dev, model):
    """prefill + paged-KV decode: greedy ids must equal the oracle's greedy loop (bit exact), and decode logits must
    agree with re-running the prefill on the extended sequence."""
    w = {k: f32(v) for k, v in _states().items()}
    for name in ("image_region", "video", "batch_pad"):
        case = cases.glue_cases()[name]
        ids = case["input_ids"].to(dev)
        am = None if case["attention_mask"] is None else case["attention_mask"].to(dev)
        images = [im.to(dev).bfloat16() for im in case["images"]]
        n_new = 12
        out, step_logits = model.generate(ids, images=images, regions=case["regions"], attention_mask=am, do_sample=False,
                                          max_new_tokens=n_new, eos_token_id=-1, return_logits=True)
        new = out[:, ids.shape[1]:].cpu()
        embeds, mask, pos = O.multimodal_prepare(w, CFGS, case["input_ids"], case["attention_mask"], case["images"], case["regions"], emulate_bf16=True)
        # oracle greedy per sample on its valid rows (the product packs sequences; the reference pads them)
        for b in range(ids.shape[0]):
            L = int(mask[b].sum())
            e = embeds[b, :L].unsqueeze(0)
            ref = O.greedy_generate(w["llama"], cases.LLM, e, torch.ones(1, L, dtype=torch.long), torch.arange(L).unsqueeze(0), n_new, True)
            # near-ties between the top-2 logits may legitimately flip under different accumulation order: compare up
            # to the first step whose oracle margin is below the numerical noise floor
            got = new[b].tolist()
            agree = 0
            for t in range(n_new):
                if got[t] != int(ref[0, t]):
                    break
                agree += 1
            if agree < n_new:
                lg = step_logits[agree][b].float().cpu()
                top2 = lg.topk(2).values
                assert float(top2[0] - top2[1]) < 3e-2 * float(lg.abs().max()), (name, b, agree, got, ref.tolist())


def test_batched_generate_rows_are_the_batch1_rows_of_the_reference(dev, model, record_property):
    """B > 1 generate: this package packs a batch, so every sample gets the ids the REFERENCE returns for that
    sample ALONE (tests/golden/greedy_batch.npz: the reference's own prepare + forward per sample, make_golden.gen_greedy_batch).
    The reference's padded-batch generate() differs from that for every sample shorter than the longest: transformers 4.31 reads
    the next token from the last COLUMN (a pad row for a right-padded sample), and the decode-step fix-up (llava_arch.py:196-205)
    extends the ids-length mask with ones, which attends the pad rows. Asserted here: ours == the batch-1 rows wherever the
    reference's top-2 margin is above the bf16 noise; the longest sample also == the reference's padded-batch row. Reported: how
    many ids of the shorter sample differ from the reference's padded-batch row (all 12 in the fixture)."""
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "greedy_batch.npz"))
    case = cases.glue_cases()["batch_pad"]
    ids = case["input_ids"].to(dev)
    n_new = int(G["batch_pad_padded_ids"].shape[1])
    out, step_logits = model.generate(ids, images=[im.to(dev).bfloat16() for im in case["images"]], regions=case["regions"],
                                      attention_mask=case["attention_mask"].to(dev), do_sample=False, max_new_tokens=n_new,
                                      eos_token_id=-1, return_logits=True)
    new = out[:, ids.shape[1]:].cpu()
    lengths = G["batch_pad_spliced_lengths"]
    longest = int(lengths.argmax())
    report = {}
    for b in range(ids.shape[0]):
        want, margin, rms = G[f"batch_pad_alone{b}_ids"], G[f"batch_pad_alone{b}_margin"], G[f"batch_pad_alone{b}_rms"]
        got = new[b].tolist()
        agree = next((t for t in range(n_new) if got[t] != int(want[t])), n_new)
        if agree < n_new:           # a flip is allowed only where the reference's own margin is within the bf16 noise of the row
            assert float(margin[agree]) < 3e-2 * float(rms[agree]), (b, agree, got, want.tolist(), float(margin[agree]))
        report[f"sample{b}_ids_equal_to_reference_batch1"] = agree
        report[f"sample{b}_ids_differing_from_reference_padded_batch"] = int(sum(int(g != int(w)) for g, w in zip(got[:agree], G["batch_pad_padded_ids"][b][:agree])))
    assert G["batch_pad_padded_ids"][longest].tolist() == G[f"batch_pad_alone{longest}_ids"].tolist()   # the fixture's own statement
    shorter = 1 - longest
    assert report[f"sample{shorter}_ids_differing_from_reference_padded_batch"] > 0     # the quirk is real and we do not reproduce it
    record_property("batched_generate_vs_reference", report)
    print("batched generate vs the reference:", report)


@pytest.mark.parametrize("op", ["bf16", "fp16"])
def test_padded_batch_generate_reproduces_the_references_batch_ids(dev, model, record_property, op):
    """generate(padded_batch=True) (round 5): the ids the REFERENCE's own B > 1 generate() returns for the right-padded
    `batch_pad` case (tests/golden/greedy_batch.npz: transformers 4.31's loop restated over the reference's forward) -- first token of the
    shorter sample read from its last PAD row, pad rows attended by every later step, the rows that share an index with the ids-length
    mask's zeros masked. Compared id by id up to the first step a flip is legitimate (the fp32 oracle's top-2 margin at that step inside
    the bf16 noise of the row); the default packed path must keep returning the batch-1 rows."""
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "greedy_batch.npz"))
    case = cases.glue_cases()["batch_pad"]
    ids, am = case["input_ids"].to(dev), case["attention_mask"].to(dev)
    odt = torch.bfloat16
    if op == "fp16":                       # the reference's inference dtype: a fresh model of the same weights in the fp16-operand build
        from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM
        st = _states()
        model = LlavaLlamaForCausalLM(LlavaConfig(**cases.LLM, mm_hidden_size=cases.MM_HIDDEN, mm_image_tower="golden/LanguageBind_Image",
                                                  mm_video_tower="golden/LanguageBind_Video_merge"))
        model.get_image_tower().load_state(cases.VIT_IMAGE, st["image_tower"])
        model.get_video_tower().load_state(cases.VIT_VIDEO, st["video_tower"])
        sd = dict(st["llama"])
        sd.update({"model.mm_projector." + k: v for k, v in st["projector"].items()})
        sd.update({"model.region_extractor." + k: v for k, v in st["region"].items()})
        model.load_state_dict(sd)
        model.to(dev, dtype=torch.float16)
        odt = torch.float16
    images = [im.to(dev).to(odt) for im in case["images"]]
    want = G["batch_pad_padded_ids"]
    n_new = int(want.shape[1])
    out, step_logits = model.generate(ids, images=images, regions=case["regions"], attention_mask=am, do_sample=False, max_new_tokens=n_new,
                                      eos_token_id=-1, return_logits=True, padded_batch=True)
    new = out[:, ids.shape[1]:].cpu()
    stats = model.last_generate_stats["padded_batch"]
    lengths = G["batch_pad_spliced_lengths"].tolist()
    shorter, longest = int(np.argmin(lengths)), int(np.argmax(lengths))
    assert stats["pad_rows"] == {shorter: max(lengths) - min(lengths)} and stats["masked_rows"] == {shorter: ids.shape[1] - int(am[shorter].sum())}
    # the fp32 oracle's padded-batch loop: the same ids as the reference (tests/test_oracle_golden.py) and the margins behind them
    w = {k: f32(v) for k, v in _states().items()}
    e, mask, pos = O.multimodal_prepare(w, CFGS, case["input_ids"], case["attention_mask"], case["images"], case["regions"])
    o_ids, o_rows = O.greedy_generate(w["llama"], cases.LLM, e, mask.long(), pos, n_new, ids_mask=case["attention_mask"].long(), return_logits=True)
    assert o_ids.tolist() == want.tolist()
    report = {}
    for b in range(ids.shape[0]):
        got = new[b].tolist()
        agree = next((t for t in range(n_new) if got[t] != int(want[b][t])), n_new)
        worst = max(rel_l2(step_logits[t][b].float().cpu(), o_rows[b, t]) for t in range(agree + (agree < n_new)))
        if agree < n_new:           # a flip is allowed only where the reference's own margin is within the bf16 noise of the row
            top2 = o_rows[b, agree].topk(2).values
            assert float(top2[0] - top2[1]) < 3e-2 * float(o_rows[b, agree].pow(2).mean().sqrt()), (b, agree, got, want[b].tolist())
        assert worst <= 1.1e-1, (b, worst)                                   # TOL_TINY_LOGITS of the decode parity tests
        report[f"sample{b}_ids_equal_to_reference_padded_batch"] = agree
        report[f"sample{b}_worst_step_logits_vs_oracle"] = round(worst, 5)
    # the first token comes from the pad row, as in the reference; the fp16 build (the reference's inference dtype) returns the reference's
    # padded-batch ids exactly (measured: 12 of 12 on both samples; bf16: 7 / 6 until a step whose margin is inside the noise)
    assert report[f"sample{shorter}_ids_equal_to_reference_padded_batch"] >= (1 if op == "bf16" else n_new)
    assert op == "bf16" or report[f"sample{longest}_ids_equal_to_reference_padded_batch"] == n_new
    alone_short = G[f"batch_pad_alone{shorter}_ids"].tolist()
    assert new[shorter].tolist() != alone_short                                       # ... and differs from the packed / batch-1 result
    packed = model.generate(ids, images=images, regions=case["regions"], attention_mask=am, do_sample=False, max_new_tokens=n_new, eos_token_id=-1)
    assert packed[shorter, ids.shape[1]:].tolist()[:1] == alone_short[:1]
    model.reset_prefix_cache()
    assert len(model.kv.free) == model.kv.num_pages                                    # every page (pads, compaction copies) returned
    record_property(f"padded_batch_generate_vs_reference_{op}", report)
    print(f"padded-batch generate vs the reference [{op}]:", report)


def test_decode_matches_prefill(dev, model):
    from vitron_amd.engine import PagedKVCache, SequenceState, llama_forward
    llama = model.get_model().llama
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(3, cases.LLM["vocab_size"], (150,), generator=g)
    emb = model.get_model().embed_tokens(ids.to(dev))
    kv = PagedKVCache(llama, 8)
    s_full = SequenceState()
    full = llama_forward(llama, kv, [s_full], emb, [150], logit_rows=list(range(150)))
    kv.release(s_full.pages)
    s = SequenceState()
    a = llama_forward(llama, kv, [s], emb[:70], [70], logit_rows=list(range(70)))
    b = llama_forward(llama, kv, [s], emb[70:149], [79], logit_rows=list(range(79)))   # chunked prefill with past
    c = llama_forward(llama, kv, [s], emb[149:150], [1])                                 # single-token decode
    got = torch.cat([a, b, c], 0)
    assert rel_l2(got, full) <= 3.9e-3   # chunking changes where the running maximum stands when P is rounded inside the attention kernels; measured 2.6e-3
    assert torch.equal(got.argmax(-1), full.argmax(-1)) or rel_l2(got, full) <= 5e-4


@pytest.mark.gpu
def test_decode_step_folded_rmsnorm_vs_oracle(dev):
    """H = 1024 (a multiple of 1024) switches vt_llama_forward's decode steps to the 5-launch layer: weight-streaming GEMMs
    with RMSNorm folded in (residual GEMMs emit bf16(x .* w) + partial sums of squares, consumers scale rows by rstd) and
    the fused rotary/append/attention kernel. Three ragged sequences, prefill then 5 batched decode steps: every step's
    logits against the fp32 oracle (same bf16-rounded weights) and against a single full prefill on the device."""
    from vitron_amd.engine import PackedLlama, PagedKVCache, SequenceState, llama_forward
    cfg = dict(synth.VICUNA_7B, hidden_size=1024, intermediate_size=1408, num_hidden_layers=3, num_attention_heads=8, vocab_size=640)
    sd = synth.llama_state(cfg, synth.make_generator(11), w_std=0.05)
    llama = PackedLlama(sd, cfg, dev)
    sd32 = {k: v.float() for k, v in sd.items()}
    g = torch.Generator().manual_seed(12)
    lens0 = [70, 129, 5]
    steps = 5
    embs = [O.bf16_round(torch.randn((n + steps, 1024), generator=g) * 0.5) for n in lens0]
    refs = []
    for e in embs:   # fp32 oracle over the whole sequence: row i only sees rows <= i, so one causal pass gives every step
        lg, _ = O.llama_forward(sd32, cfg, e.unsqueeze(0))
        refs.append(lg[0])
    kv = PagedKVCache(llama, 16)
    seqs = [SequenceState() for _ in lens0]
    flat = torch.cat([e[:n] for e, n in zip(embs, lens0)]).to(dev).bfloat16()
    got = [llama_forward(llama, kv, seqs, flat, lens0)]
    for t in range(steps):
        x = torch.stack([e[n + t] for e, n in zip(embs, lens0)]).to(dev).bfloat16()
        got.append(llama_forward(llama, kv, seqs, x, [1] * len(lens0)))
    # the same rows through ONE prefill per sequence (tile GEMMs, separate RMSNorm, flash attention)
    full = []
    for e, n in zip(embs, lens0):
        s_ = SequenceState()
        full.append(llama_forward(llama, kv, [s_], e.to(dev).bfloat16(), [n + steps], logit_rows=list(range(n + steps))))
        kv.release(s_.pages)
    worst = max(rel_l2(got[t][b].cpu(), refs[b][n - 1 + t]) for t in range(steps + 1) for b, n in enumerate(lens0))
    print(f"[parity-tiny] decode_folded_rmsnorm_h1024: worst_step_vs_fp32 {worst:.3e}", flush=True)
    for t in range(steps + 1):
        for b, n in enumerate(lens0):
            row = n - 1 + t
            ref, dev_full, g_ = refs[b][row], full[b][row].cpu(), got[t][b].cpu()
            assert rel_l2(g_, ref) <= 3e-2, (t, b, rel_l2(g_, ref))            # single rows after 32 layers; worst measured 2.56e-2
            assert rel_l2(g_, ref) <= 1.5 * rel_l2(dev_full, ref) + 2e-3, (t, b)       # folding costs no accuracy
            assert rel_l2(g_, dev_full) <= 1.6e-2, (t, b, rel_l2(g_, dev_full))   # two independent bf16 paths, each <= ~1e-2 from fp32; measured 1.03e-2


@pytest.mark.gpu
def test_multi_turn_reuses_towers_and_kv_prefix(dev):
    """Second turn of a conversation (same image + box, prompt = first prompt + reply + new text): the image is not encoded
    again, only the rows behind the last whole common page are prefilled, and the logits match a model without reuse."""
    from vitron_amd import synth
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM
    cfg = dict(synth.VICUNA_7B, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, vocab_size=512)
    vit = dict(synth.VIT_L14, image_size=112, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2)
    models = []
    for reuse in (True, False):
        m = LlavaLlamaForCausalLM(LlavaConfig(**cfg, mm_hidden_size=128, mm_region_image_size=112, kv_prefix_reuse=reuse))
        m.init_synthetic(dev, seed=7, vit_image=vit, vit_video=None)
        models.append(m)
    g = torch.Generator().manual_seed(5)
    image = torch.randn((3, 112, 112), generator=g).bfloat16().to(dev)
    box = [10.0, 20.0, 90.0, 100.0]
    p1 = torch.tensor([[1, -200] + torch.randint(3, 500, (40,), generator=g).tolist() + [-300, 1] +
                       torch.randint(3, 500, (30,), generator=g).tolist()], device=dev)
    outs = []
    for m in models:
        o1 = m.generate(p1, images=[image], regions=[box], do_sample=False, max_new_tokens=12, eos_token_id=-1)
        reply = o1[0, p1.shape[1]:]
        p2 = torch.cat([p1[0], reply, torch.tensor([5, 6, 7, 8, 9], device=dev)]).unsqueeze(0)
        o2, lg = m.generate(p2, images=[image], regions=[box], do_sample=False, max_new_tokens=4, eos_token_id=-1, return_logits=True)
        outs.append((o1, o2, lg, dict(m.last_generate_stats)))
    (a1, a2, la, sa), (b1, b2, lb, sb) = outs
    assert torch.equal(a1, b1)                                  # first turn: nothing to reuse, identical work
    rows = 64 + 1 + 30 + 1 + 40 + 1 + 12 + 5                     # spliced prompt of turn 2: 64 visual rows + region row + text
    assert sa["prompt_rows"] == rows and sb["prompt_rows"] == rows
    assert sa["tower_items"] == 0 and sa["reused_tokens"] == 128 and sa["prefill_rows"] == rows - 128
    assert sb["reused_tokens"] == 0 and sb["prefill_rows"] == rows
    for x, y in zip(la, lb):                                    # chunked prefill behind a cached prefix == full prefill
        assert rel_l2(x.float(), y.float()) <= 5e-3
    # a different box invalidates the region row (row 65) -> only the first whole page (visual rows) is kept
    o3 = models[0].generate(p2, images=[image], regions=[[0.0, 0.0, 50.0, 50.0]], do_sample=False, max_new_tokens=2, eos_token_id=-1)
    assert models[0].last_generate_stats["reused_tokens"] == 64 and models[0].last_generate_stats["tower_items"] == 1
    models[0].reset_prefix_cache()
    assert len(models[0].kv.free) == models[0].kv.num_pages


@pytest.mark.gpu
def test_padded_batch_left_padding_and_refusals(dev, model):
    """generate(padded_batch=True) with tokenizer_padding_side = 'left' (reference llava_arch.py:379-386): a batch whose samples carry the same
    number of visual rows gets, per sample, the ids of the sample alone -- what the reference's loop returns there
    (tests/test_oracle_golden.py::test_left_padded_batch_with_equal_visual_rows_is_every_sample_alone); unequal padding and text-only batches
    are refused explicitly (never silently different ids)."""
    from tests.test_oracle_golden import _left_padded_batch
    case = _left_padded_batch()
    ids, am = case["input_ids"].to(dev), case["attention_mask"].to(dev)
    images = [im.to(dev).to(torch.bfloat16) for im in case["images"]]
    n = 8
    old = getattr(model.config, "tokenizer_padding_side", "right")
    model.config.tokenizer_padding_side = "left"
    try:
        out = model.generate(ids, images=images, regions=case["regions"], attention_mask=am, do_sample=False, max_new_tokens=n, eos_token_id=-1,
                             padded_batch=True)
        for b, solo in enumerate(case["solo"]):
            one = model.generate(torch.tensor([solo], device=dev), images=[images[b]], regions=[case["regions"][b]], do_sample=False,
                                 max_new_tokens=n, eos_token_id=-1)
            assert out[b, ids.shape[1]:].tolist() == one[0, len(solo):].tolist(), b
        # unequal numbers of visual rows: the reference attends pad rows / masks real rows there -- refused
        bad_ids = ids.clone()
        assert int(bad_ids[1, -4]) == -200
        bad_ids[1, -4], bad_ids[1, -3] = 5, 6     # the second sample loses its <image> / <objs> rows: padding in id space != in the spliced space
        with pytest.raises(NotImplementedError):
            model.generate(bad_ids, images=images, regions=case["regions"], attention_mask=am, do_sample=False, max_new_tokens=2, padded_batch=True)
        # a right-padded mask under the left setting
        with pytest.raises(NotImplementedError):
            model.generate(ids, images=images, regions=case["regions"], attention_mask=am.flip(1), do_sample=False, max_new_tokens=2, padded_batch=True)
    finally:
        model.config.tokenizer_padding_side = old
    # text-only batches take another path in the reference (llava_arch.py:196 returns early): refused, in either padding mode
    txt = torch.tensor([[1, 5, 6, 7], [1, 8, 0, 0]], device=dev)
    with pytest.raises(NotImplementedError):
        model.generate(txt, attention_mask=torch.tensor([[1, 1, 1, 1], [1, 1, 0, 0]], device=dev), do_sample=False, max_new_tokens=2, padded_batch=True)
    model.reset_prefix_cache()


def test_serving_engine_continuous_batching_matches_solo_runs(dev, model):
    """Five requests (text-only, image, image + region, video) joining a running decode batch at different steps: every
    request's greedy tokens equal the tokens `generate` produces for it alone; pages all return to the pool."""
    from vitron_amd.serving import ServingEngine
    g = torch.Generator().manual_seed(21)
    V = cases.LLM["vocab_size"]
    img = lambda: torch.randn((3, 56, 56), generator=g).bfloat16().to(dev)            # noqa: E731
    clip = torch.randn((3, 4, 56, 56), generator=g).bfloat16().to(dev)                  # the golden video tower: 4 frames
    rnd = lambda n: torch.randint(3, V, (n,), generator=g).tolist()                     # noqa: E731
    reqs = [
        dict(input_ids=torch.tensor([[1] + rnd(23)]), images=None, regions=None, max_new_tokens=9),
        dict(input_ids=torch.tensor([[1, -200] + rnd(11)]), images=[img()], regions=None, max_new_tokens=14),
        dict(input_ids=torch.tensor([[1, -200] + rnd(5) + [-300, 1] + rnd(7)]), images=[img()], regions=[[20.0, 30.0, 150.0, 200.0]], max_new_tokens=6),
        dict(input_ids=torch.tensor([[1] + [-200] * 4 + rnd(9)]), images=[clip], regions=None, max_new_tokens=11),
        dict(input_ids=torch.tensor([[1] + rnd(70)]), images=None, regions=None, max_new_tokens=5),
    ]
    model.config.kv_prefix_reuse = False
    solo = []
    for r in reqs:
        o = model.generate(r["input_ids"].to(dev), images=r["images"], regions=r["regions"], do_sample=False,
                           max_new_tokens=r["max_new_tokens"], eos_token_id=-1)
        solo.append(o[0, r["input_ids"].shape[1]:].cpu())
    eng = ServingEngine(model, max_batch=3, kv_pages=64)
    ids = [eng.submit(reqs[0]["input_ids"], reqs[0]["images"], reqs[0]["regions"], reqs[0]["max_new_tokens"], eos_token_id=-1),
           eng.submit(reqs[1]["input_ids"], reqs[1]["images"], reqs[1]["regions"], reqs[1]["max_new_tokens"], eos_token_id=-1)]
    seen = {i: [] for i in range(5)}
    steps = 0
    while eng.pending():
        if steps == 2:      # two more arrive while the first two decode; max_batch = 3 makes the fourth wait for a free slot
            ids.append(eng.submit(reqs[2]["input_ids"], reqs[2]["images"], reqs[2]["regions"], reqs[2]["max_new_tokens"], eos_token_id=-1))
            ids.append(eng.submit(reqs[3]["input_ids"], reqs[3]["images"], reqs[3]["regions"], reqs[3]["max_new_tokens"], eos_token_id=-1))
        if steps == 5:
            ids.append(eng.submit(reqs[4]["input_ids"], reqs[4]["images"], reqs[4]["regions"], reqs[4]["max_new_tokens"], eos_token_id=-1))
        for rid, t in eng.step():
            seen[rid].append(t)
        steps += 1
        assert len(eng.active) <= 3 and steps < 200
    assert ids == [0, 1, 2, 3, 4]
    for i in range(5):
        assert seen[i] == solo[i].tolist(), (i, seen[i], solo[i].tolist())
    assert len(model.kv.free) == model.kv.num_pages
    # batch_prefill: the admitted requests share one packed decoder prefill -> same tokens up to logit near-ties
    eng = ServingEngine(model, max_batch=5, kv_pages=64, batch_prefill=True)
    for r in reqs:
        eng.submit(r["input_ids"], r["images"], r["regions"], r["max_new_tokens"], eos_token_id=-1)
    outs = eng.run()
    agree = sum(int(outs[i].tolist() == solo[i].tolist()) for i in range(5))
    assert agree >= 4 and all(len(outs[i]) == len(solo[i]) for i in range(5)), (agree, [outs[i].tolist() for i in range(5)])
    assert len(model.kv.free) == model.kv.num_pages
    model.config.kv_prefix_reuse = True


def test_serving_engine_pool_sizing_and_waiting(dev, model):
    """(1) the default engine (kv_pages=None) sizes its pool for the whole admitted batch, also when a later request
    of the batch is larger than the first; (2) a request that does not fit waits (embedded once) until running requests retire;
    (3) a request larger than the pool is admitted once the engine is idle and the pool can grow; (4) a failing admission leaves
    the queue and the pool as they were. Tokens always equal the solo greedy run."""
    from vitron_amd.serving import ServingEngine
    g = torch.Generator().manual_seed(31)
    V = cases.LLM["vocab_size"]
    rnd = lambda n: torch.randint(3, V, (n,), generator=g).tolist()                     # noqa: E731
    reqs = [dict(input_ids=torch.tensor([[1] + rnd(20)]), max_new_tokens=6),            # 1 + 1 pages
            dict(input_ids=torch.tensor([[1] + rnd(200)]), max_new_tokens=40),          # 4 + 1 pages: larger than the first
            dict(input_ids=torch.tensor([[1] + rnd(90)]), max_new_tokens=10)]
    model.config.kv_prefix_reuse = False
    solo = [model.generate(r["input_ids"].to(dev), do_sample=False, max_new_tokens=r["max_new_tokens"], eos_token_id=-1)[0, r["input_ids"].shape[1]:].tolist()
            for r in reqs]
    model.reset_prefix_cache()
    model.kv, model.kv_pages = None, None
    eng = ServingEngine(model, max_batch=4)                                             # (1) kv_pages=None, mixed sizes
    for r in reqs:
        eng.submit(r["input_ids"], max_new_tokens=r["max_new_tokens"], eos_token_id=-1)
    eng.step()
    assert len(eng.active) == 3 and not eng.waiting                                     # all three decode together
    outs = eng.run()
    assert [outs[i].tolist() for i in range(3)] == solo
    assert len(model.kv.free) == model.kv.num_pages
    # (2) a pool that the first two fill exactly: the third arrives later, waits, is embedded ONCE, and joins when pages come back
    eng = ServingEngine(model, max_batch=4, kv_pages=7)
    embeds = []
    orig = eng._embed
    eng._embed = lambda r: (embeds.append(r.rid), orig(r))[1]
    for r in reqs[:2]:
        eng.submit(r["input_ids"], max_new_tokens=r["max_new_tokens"], eos_token_id=-1)
    eng.step()
    eng.submit(reqs[2]["input_ids"], max_new_tokens=reqs[2]["max_new_tokens"], eos_token_id=-1)
    eng.step()
    assert len(eng.active) == 2 and len(eng.waiting) == 1                               # 3 pages wanted, none free
    outs = eng.run()
    assert [outs[i].tolist() for i in range(3)] == solo and sorted(embeds) == [0, 1, 2]
    assert len(model.kv.free) == model.kv.num_pages == 7
    # (3) the constructor's kv_pages is an UPPER BOUND (round 4): a request that cannot fit a pool of that size fails on its
    # own -- with its exception, without being re-raised on every step -- and the requests around it are served; the pool is not rebuilt
    eng = ServingEngine(model, max_batch=4, kv_pages=3)
    eng.submit(reqs[0]["input_ids"], max_new_tokens=reqs[0]["max_new_tokens"], eos_token_id=-1)
    eng.step()
    big = eng.submit(reqs[1]["input_ids"], max_new_tokens=reqs[1]["max_new_tokens"], eos_token_id=-1)     # needs 5 pages, the pool is capped at 3
    eng.submit(reqs[0]["input_ids"], max_new_tokens=reqs[0]["max_new_tokens"], eos_token_id=-1)           # behind it: must not starve
    outs = eng.run()
    assert outs[0].tolist() == solo[0] and outs[2].tolist() == solo[0] and big not in outs and model.kv.num_pages == 3
    assert big in eng.errors() and "capped at 3" in str(eng.errors()[big]) and len(model.kv.free) == 3
    # (4) pages held by someone else: the request does not fit right now -> it WAITS (no exception, nothing leaked) and runs once they return
    eng = ServingEngine(model, max_batch=4, kv_pages=6)
    held = model.kv.alloc(2)
    eng.submit(reqs[1]["input_ids"], max_new_tokens=reqs[1]["max_new_tokens"], eos_token_id=-1)
    assert eng.step() == [] and len(eng.waiting) == 1 and not eng.active and len(model.kv.free) == 4
    with pytest.raises(RuntimeError, match="no progress"):        # run() must not spin while the caller holds the pages
        eng.run()
    assert len(eng.waiting) == 1 and len(model.kv.free) == 4      # the request is still queued, nothing leaked
    model.kv.release(held)
    assert eng.run()[0].tolist() == solo[1]
    model.config.kv_prefix_reuse = True


@pytest.mark.gpu
def test_decode_feed_kernel(dev):
    """vt_decode_feed against its definition: pad once finished, EOS flags, embedding rows, metadata advance."""
    from vitron_amd import ops
    g = torch.Generator().manual_seed(11)
    V, H, B = 97, 256, 5
    table = torch.randn((V, H), generator=g).bfloat16().to(dev)
    nxt = torch.tensor([3, 96, 50, 7, 7], dtype=torch.int32, device=dev)
    fin = torch.tensor([0, 0, 1, 0, 0], dtype=torch.int32, device=dev)
    eos = torch.tensor([7, 96], dtype=torch.int32, device=dev)
    desc = torch.tensor([[i, 1, 10 * i + 5, 3 * i] for i in range(B)], dtype=torch.int32, device=dev)
    pos = torch.tensor([10 * i + 4 for i in range(B)], dtype=torch.int32, device=dev)
    tok = torch.full((2, B), -1, dtype=torch.int32, device=dev)
    x = torch.zeros((B, H), dtype=torch.bfloat16, device=dev)
    d0, p0 = desc.clone(), pos.clone()
    ops.decode_feed(table, nxt, fin, eos, 2, tok, x, desc, pos)
    want = [3, 96, 2, 7, 7]                                     # sequence 2 was finished already: pad
    assert tok[0].tolist() == want and tok[1].tolist() == [0, 1, 1, 1, 1] and fin.tolist() == [0, 1, 1, 1, 1]
    assert torch.equal(x, table[torch.tensor(want, device=dev)])
    assert torch.equal(desc[:, 2], d0[:, 2] + 1) and torch.equal(desc[:, [0, 1, 3]], d0[:, [0, 1, 3]]) and torch.equal(pos, p0 + 1)
    ops.decode_feed(table, nxt, fin, None, 2, tok, x, desc, pos)   # no EOS set: flags only carry over
    assert tok[0].tolist() == [3, 2, 2, 2, 2] and fin.tolist() == [0, 1, 1, 1, 1] and torch.equal(pos, p0 + 2)
    with pytest.raises(Exception):
        ops.decode_feed(table, nxt[:3], fin, eos, 2, tok, x, desc, pos)


@pytest.mark.gpu
def test_decode_state_matches_host_driven_steps(dev, model):
    """The device-resident decode state runs the same launches as llama_forward with host-built metadata: logits are
    bit-identical step by step (ragged batch, page boundaries crossed), and rollback() leaves the cache consistent."""
    from vitron_amd import ops
    from vitron_amd.engine import DecodeState, PagedKVCache, SequenceState, llama_forward
    llama = model.get_model().llama
    g = torch.Generator().manual_seed(21)
    lens = [61, 130, 7]
    ids = torch.randint(3, cases.LLM["vocab_size"], (sum(lens),), generator=g).to(dev)
    emb = model.get_model().embed_tokens(ids)
    steps = 9
    runs = []
    for use_state in (False, True):
        kv = PagedKVCache(llama, 16)
        seqs = [SequenceState() for _ in lens]
        logits = llama_forward(llama, kv, seqs, emb, lens)
        trace = [logits.clone()]
        state = DecodeState(llama, kv, seqs, steps + 1) if use_state else None
        for _ in range(steps):
            tok = ops.argmax(logits)
            if use_state:
                state.feed(tok)
                logits = state.forward()
            else:
                logits = llama_forward(llama, kv, seqs, model.get_model().embed_tokens(tok.long()), [1] * len(lens))
            trace.append(logits.clone())
        runs.append((trace, [s.length for s in seqs], state, kv, seqs))
    (ta, la, _, _, _), (tb, lb, state, kv, seqs) = runs
    assert la == lb == [l + steps for l in lens]
    for a, b in zip(ta, tb):
        assert torch.equal(a, b)
    # speculative pass + rollback, then the same token again: same logits as the first time
    tok = ops.argmax(tb[-1])
    slot = state.feed(tok)
    l1 = state.forward().clone()
    toks, fin = state.read(slot)
    assert toks == tok.tolist() and fin == [False] * len(lens)
    state.rollback()
    assert [s.length for s in seqs] == lb
    l2 = llama_forward(llama, kv, seqs, model.get_model().embed_tokens(tok.long()), [1] * len(lens))
    assert torch.equal(l1, l2)
    with pytest.raises(Exception):
        DecodeState(llama, kv, [SequenceState()], 4)             # no context


@pytest.mark.gpu
def test_generate_stops_on_eos_with_speculative_step_rolled_back(dev):
    """generate() enqueues pass t+1 before it reads token t: when token t is EOS (or a stop keyword fires) the extra pass
    must not leak into the output, the pad/finished rule of a batch, or the KV prefix kept for the next turn."""
    from vitron_amd import synth
    from vitron_amd.mm_utils import KeywordsStoppingCriteria
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM
    cfg = dict(synth.VICUNA_7B, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, vocab_size=512)
    m = LlavaLlamaForCausalLM(LlavaConfig(**cfg, mm_hidden_size=128, mm_region_image_size=112))
    m.init_synthetic(dev, seed=9, vit_image=None, vit_video=None)
    g = torch.Generator().manual_seed(2)
    p = torch.tensor([[1] + torch.randint(3, 500, (90,), generator=g).tolist(),
                      [1] + torch.randint(3, 500, (90,), generator=g).tolist()], device=dev)
    m.config.kv_prefix_reuse = False
    # sampled (counter-based device RNG: same seed, same tokens) so that the tiny random model emits distinct tokens
    kw_args = dict(do_sample=True, temperature=1.0, top_p=1.0, seed=123)
    full = m.generate(p, max_new_tokens=10, eos_token_id=-1, **kw_args)
    free = len(m.kv.free)
    assert free == m.kv.num_pages
    new = full[:, p.shape[1]:].tolist()
    # stop row 0 at its 4th token, row 1 at its 7th: row 0 pads from then on, the run ends at step 7
    e0, e1 = new[0][3], new[1][6]
    if e0 in new[1][:6] or e1 in new[0][:3] or e0 in new[0][:3] or e1 in new[1][:6]:
        pytest.skip("synthetic tokens collide with the chosen EOS ids")
    out = m.generate(p, max_new_tokens=10, eos_token_id=[e0, e1], pad_token_id=0, **kw_args)
    got = out[:, p.shape[1]:].tolist()
    assert got[1] == new[1][:7] and got[0] == new[0][:4] + [0, 0, 0]
    assert len(m.kv.free) == free
    # keyword stop on a single row, with the multi-turn prefix kept: the follow-up turn must see a consistent cache
    class Tok:
        bos_token_id = 1
        def __call__(self, text):
            r = type("R", (), {})()
            r.input_ids = [1] + [int(t) for t in text.split()]
            return r
        def batch_decode(self, ids, skip_special_tokens=True):
            return [" ".join(str(int(t)) for t in row) for row in ids]
    p1 = p[:1]
    new1 = m.generate(p1, max_new_tokens=10, eos_token_id=-1, **kw_args)[0, p1.shape[1]:].tolist()
    m.config.kv_prefix_reuse = True
    kw = f"{new1[4]} {new1[5]}"
    crit = KeywordsStoppingCriteria([kw], Tok(), p1)
    o1 = m.generate(p1, max_new_tokens=10, eos_token_id=-1, stopping_criteria=[crit], **kw_args)
    assert o1[0, p1.shape[1]:].tolist() == new1[:6]
    p2 = torch.cat([o1[0], torch.tensor([11, 12, 13], device=dev)]).unsqueeze(0)
    o2, lg2 = m.generate(p2, do_sample=False, max_new_tokens=3, eos_token_id=-1, return_logits=True)
    assert m.last_generate_stats["reused_tokens"] == 64
    m.reset_prefix_cache()
    m.config.kv_prefix_reuse = False
    o3, lg3 = m.generate(p2, do_sample=False, max_new_tokens=3, eos_token_id=-1, return_logits=True)
    for a, b in zip(lg2, lg3):
        assert rel_l2(a.float(), b.float()) <= 3.9e-3          # measured 2.6e-3
    assert torch.equal(o2, o3) or rel_l2(lg2[-1].float(), lg3[-1].float()) <= 5e-4


@pytest.mark.gpu
def test_prefill_folded_rmsnorm_vs_oracle(dev):
    """rows > 64: RMSNorm folded into the MFMA tile GEMMs (residual GEMMs emit bf16(x .* w_next) + per-row partial sums of x^2,
    vt_rowscale_finalize turns them into rstd, the consumer GEMMs scale their accumulator rows). Every row's logits against the
    fp32 oracle, with the fold and with separate norm launches (vt_llama_model.prefill_norm_fold = 0): folding costs no accuracy. Shapes chosen
    to hit the 4-phase kernel (K % 128 == 0), the small tiles (K = 192 gate / down) and ragged row counts."""
    import os
    from vitron_amd.engine import PackedLlama, PagedKVCache, SequenceState, llama_forward
    for H, I, heads, rows in ((1024, 1408, 8, 333), (256, 192, 2, 97)):
        cfg = dict(synth.VICUNA_7B, hidden_size=H, intermediate_size=I, num_hidden_layers=3, num_attention_heads=heads, vocab_size=640)
        sd = synth.llama_state(cfg, synth.make_generator(31), w_std=0.05)
        llama = PackedLlama(sd, cfg, dev)
        sd32 = {k: v.float() for k, v in sd.items()}
        g = torch.Generator().manual_seed(32)
        e = O.bf16_round(torch.randn((rows, H), generator=g) * 0.5)
        ref, _ = O.llama_forward(sd32, cfg, e.unsqueeze(0))
        ref = ref[0]
        kv = PagedKVCache(llama, 16)
        got = {}
        for fold in ("1", "0"):
            llama.set_prefill_norm_fold(fold == "1")
            try:
                s_ = SequenceState()
                got[fold] = llama_forward(llama, kv, [s_], e.to(dev).bfloat16(), [rows], logit_rows=list(range(rows))).cpu()
                kv.release(s_.pages)
            finally:
                llama.set_prefill_norm_fold(False)
        ef, es = rel_l2(got["1"], ref), rel_l2(got["0"], ref)
        assert not torch.equal(got["1"], got["0"])                    # the switch really selects two different paths
        print(f"[parity-tiny] prefill_norm_fold_h{H}: folded_vs_fp32 {ef:.3e} separate_vs_fp32 {es:.3e}", flush=True)
        assert ef <= 3e-2 and es <= 3e-2, (H, ef, es)                 # measured 2.0e-2 at H = 1024 (32 layers), x 1.5
        assert ef <= 1.25 * es + 1e-3, (H, ef, es)                    # no farther from fp32 than the separate-norm path
        worst = max(rel_l2(got["1"][r], ref[r]) for r in range(rows))
        assert worst <= 4.9e-2, (H, worst)                            # no single row off (a wrong row factor would be O(1)); measured 3.2e-2


@pytest.mark.gpu
def test_generate_edge_cases_empty_and_ragged(dev):
    """Degenerate calls the chat loop can produce: nothing to generate, a one-token prompt, a very ragged text batch (1 and 200
    tokens, right-padded with a mask) -- same tokens as running each sequence alone, pages all returned."""
    from vitron_amd import synth
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM
    cfg = dict(synth.VICUNA_7B, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, vocab_size=512)
    m = LlavaLlamaForCausalLM(LlavaConfig(**cfg, mm_hidden_size=128, mm_region_image_size=112, kv_prefix_reuse=False))
    m.init_synthetic(dev, seed=19, vit_image=None, vit_video=None)
    g = torch.Generator().manual_seed(4)
    long = torch.tensor([[1] + torch.randint(3, 500, (199,), generator=g).tolist()], device=dev)
    one = torch.tensor([[7]], device=dev)
    out0 = m.generate(long, do_sample=False, max_new_tokens=0, eos_token_id=-1)
    assert torch.equal(out0, long)                                                  # nothing generated, nothing lost
    a = m.generate(one, do_sample=False, max_new_tokens=5, eos_token_id=-1)
    b = m.generate(long, do_sample=False, max_new_tokens=5, eos_token_id=-1)
    assert a.shape == (1, 6) and b.shape == (1, 205)
    ids = torch.zeros((2, 200), dtype=torch.long, device=dev)
    mask = torch.zeros((2, 200), dtype=torch.long, device=dev)
    ids[0, :1], mask[0, :1] = one[0], 1
    ids[1], mask[1] = long[0], 1
    both = m.generate(ids, attention_mask=mask, do_sample=False, max_new_tokens=5, eos_token_id=-1)
    assert both[0, 200:].tolist() == a[0, 1:].tolist()
    assert both[1, 200:].tolist() == b[0, 200:].tolist()
    assert len(m.kv.free) == m.kv.num_pages


@pytest.mark.gpu
def test_multimodal_glue_random_layouts_vs_reference(dev, model):
    """The 24 seeded random batches the REFERENCE's prepare_inputs_labels_for_multimodal was run on (tests/golden/
    glue_random.npz): through the product (towers, region extractor, projector, splice kernel) the padded shape and the mask are
    exact and every row of the spliced embeddings lands where the reference put it (stored: a fixed random projection of the
    reference's fp32 embeddings; visual rows carry bf16 storage noise, text rows are exact)."""
    g = np.load(os.path.join(G, "glue_random.npz"))
    proj = cases.glue_projection(cases.LLM["hidden_size"])
    try:
        for name, case in cases.random_glue_cases().items():
            model.config.tokenizer_model_max_length = case.get("max_length")
            model.config.tokenizer_padding_side = case.get("padding_side", "right")
            ids = case["input_ids"].to(dev)
            am = None if case["attention_mask"] is None else case["attention_mask"].to(dev)
            images = [im.to(dev).bfloat16() for im in case["images"]]
            (_, pos, mask, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids, None, am, None, None, images, case["regions"])
            ref_m, ref_p = g[f"{name}_mask"], torch.as_tensor(g[f"{name}_proj"])
            assert np.array_equal(np.array(model._last_splice[0], dtype=np.int32), ref_m), name
            got = embeds.double().cpu() @ proj
            assert got.shape == ref_p.shape, name
            assert rel_l2(got.float(), ref_p.float()) <= 7.4e-3, (name, rel_l2(got.float(), ref_p.float()))   # measured 4.9e-3
            # a misplaced row would show as an O(1) error in that row: bound the worst row against the scale of the case
            assert float((got - ref_p).abs().max()) <= 0.05 * float(ref_p.abs().max()), name
    finally:
        model.config.tokenizer_model_max_length = None
        model.config.tokenizer_padding_side = "right"


@pytest.mark.gpu
def test_output_hidden_states_and_attentions(dev, model):
    """forward(output_hidden_states=True) (reference llava_llama.py:69 -> transformers 4.31 LlamaModel.forward): num_layers + 1 tensors
    [B, S, H] -- the stream in front of every decoder layer (entry 0 = the spliced embeddings), then the final-normed output of the last
    layer -- on a right-padded batch (zeros at the padding rows), against the oracle layer by layer. output_attentions=True is refused
    explicitly (the flash kernels never materialise the probabilities)."""
    case = cases.glue_cases()["batch_pad"]
    ids, am = case["input_ids"].to(dev), case["attention_mask"].to(dev)
    images = [im.to(dev).bfloat16() for im in case["images"]]
    out = model(input_ids=ids, attention_mask=am, images=images, regions=case["regions"], use_cache=False, output_hidden_states=True)
    L = cases.LLM["num_hidden_layers"]
    hs = out.hidden_states
    assert isinstance(hs, tuple) and len(hs) == L + 1
    w = {k: f32(v) for k, v in _states().items()}
    e, mask, pos = O.multimodal_prepare(w, CFGS, case["input_ids"], case["attention_mask"], case["images"], case["regions"], emulate_bf16=True)
    valid = mask.bool()
    assert all(tuple(h.shape) == tuple(e.shape) for h in hs)
    assert all(float(h.float().cpu()[~valid].abs().max()) == 0.0 for h in hs)               # padding rows: zeros
    with torch.no_grad():
        for l in range(L):
            _, _, h = O.llama_forward(w["llama"], cases.LLM, e, pos, mask, None, True, num_layers=l, return_hidden=True)
            d = rel_l2(hs[l].float().cpu()[valid], h[valid])
            assert d <= (6e-3 if l == 0 else TOL_DEEP), (l, d)
        _, _, h = O.llama_forward(w["llama"], cases.LLM, e, pos, mask, None, True, return_hidden=True)
        final = O.bf16_round(O.rmsnorm(h, w["llama"]["model.norm.weight"], cases.LLM["rms_norm_eps"]))
        assert rel_l2(hs[L].float().cpu()[valid], final[valid]) <= TOL_DEEP
    assert hs[L].dtype == model.dtype and hs[0].dtype == torch.float32
    plain = model(input_ids=ids, attention_mask=am, images=images, regions=case["regions"], use_cache=False)
    assert plain.hidden_states is None and torch.equal(plain.logits, out.logits)            # the trace changes nothing
    with pytest.raises(NotImplementedError):
        model(input_ids=ids, attention_mask=am, images=images, regions=case["regions"], use_cache=False, output_attentions=True)
