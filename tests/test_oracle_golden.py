"""Pin the CPU oracle (oracle/vitron_oracle.py) against golden vectors produced by the REFERENCE's own modules
(tests/golden/make_golden.py, run in the build container against /root/reference). CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import vitron_oracle as O
from tests.golden import cases
from vitron_amd import synth

G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def f32(sd):
    return {k: v.float() for k, v in sd.items()}


class StubTok:
    bos_token_id = 1

    def __call__(self, text):
        class R:
            pass
        r = R()
        r.input_ids = [1] + [100 + ord(c) for c in text]
        return r


def test_mm_utils_known_answers():
    g = np.load(os.path.join(G, "mm_utils.npz"))
    tok = StubTok()
    for i, p in enumerate(cases.PROMPTS):
        assert O.tokenizer_image_token(p, tok) == g[f"image_token_{i}"].tolist()
        assert O.tokenizer_image_region_token(p, tok) == g[f"region_token_{i}"].tolist()
    for i, (r, isz, tsz) in enumerate(cases.REGION_RESCALE):
        assert O.preprocess_region(r, isz, tsz) == g[f"preprocess_region_{i}"].tolist()
    # the known answers quoted in SURVEY.md Appendix B
    assert O.tokenizer_image_token("ab<image>cd", tok) == [1, 197, 198, -200, 199, 200]
    assert O.tokenizer_image_region_token("a<image>\n<objs> b", tok) == [1, 197, -200, 110, -300, 1, 132, 198]


@pytest.mark.parametrize("name,cfg,shape", [("video", cases.VIT_VIDEO, (2, 3, 4, 56, 56)), ("image", cases.VIT_IMAGE, (3, 3, 56, 56))])
def test_vit_hidden_states(name, cfg, shape):
    g = np.load(os.path.join(G, "vit.npz"))
    sd = synth.vit_state(cfg, synth.make_generator(cases.SEED_VIT), **cases.VIT_INIT)
    assert synth.checksum(sd) == pytest.approx(float(g[f"{name}_checksum"]), rel=1e-12)
    x = cases.pixels(shape, cases.SEED_PIX)
    for nl in range(cfg["num_hidden_layers"] + 1):
        h = O.vit_forward(f32(sd), cfg, x, num_layers=nl)
        assert rel(h, g[f"{name}_hidden_{nl}"]) <= 1e-5, nl
    # feature_select: hidden_states[-2] without CLS
    f = O.tower_features(f32(sd), cfg, x, -2)
    ref = torch.as_tensor(g[f"{name}_hidden_{cfg['num_hidden_layers'] - 1}"])[:, 1:]
    assert rel(f.reshape(ref.shape), ref) <= 1e-5


@pytest.mark.parametrize("name", list(cases.VIT_B_CASES))
def test_vit_hidden_states_second_shape(name):
    """quick_gelu, 8 frames, a 5 x 5 patch grid: every hidden state of the reference's towers (tests/golden/vit_b.npz)."""
    g = np.load(os.path.join(G, "vit_b.npz"))
    cfg, shape = cases.VIT_B_CASES[name]
    sd = synth.vit_state(cfg, synth.make_generator(cases.SEED_VIT + 1), **cases.VIT_INIT)
    assert synth.checksum(sd) == pytest.approx(float(g[f"{name}_checksum"]), rel=1e-12)
    x = cases.pixels(shape, cases.SEED_PIX + 9)
    for nl in range(cfg["num_hidden_layers"] + 1):
        h = O.vit_forward(f32(sd), cfg, x, num_layers=nl)
        assert rel(h, g[f"{name}_hidden_{nl}"]) <= 1e-5, nl


@pytest.mark.parametrize("name", list(cases.VIT_TMLP_CASES))
def test_vit_hidden_states_image_tower_with_time_attention(name):
    """The IMAGE tower file's add_time_attn variant (temporal attention + temporal MLP per layer, image/modeling_image.py:74-84,
    105-134) on 4-frame clips and in its num_frames = 1 form: every hidden state of the reference's module (tests/golden/vit_tmlp.npz)."""
    g = np.load(os.path.join(G, "vit_tmlp.npz"))
    cfg, shape = cases.VIT_TMLP_CASES[name]
    sd = synth.vit_state(cfg, synth.make_generator(cases.SEED_VIT + 2), **cases.VIT_INIT)
    assert synth.checksum(sd) == pytest.approx(float(g[f"{name}_checksum"]), rel=1e-12)
    assert any("temporal_mlp.fc1" in k for k in sd)
    x = cases.pixels(shape, cases.SEED_PIX + 2)
    for nl in range(cfg["num_hidden_layers"] + 1):
        h = O.vit_forward(f32(sd), cfg, x, num_layers=nl)
        assert rel(h, g[f"{name}_hidden_{nl}"]) <= 1e-5, nl


@pytest.mark.parametrize("tag", list(cases.REGION_CASES))
def test_region_extractor(tag):
    g = np.load(os.path.join(G, "region_projector.npz"))
    cin, cout, grid = cases.REGION_CASES[tag]
    sd = synth.region_state(cin, cout, synth.make_generator(cases.SEED_REGION), **cases.MLP_INIT)
    assert synth.checksum(sd) == pytest.approx(float(g[f"region_{tag}_checksum"]), rel=1e-12)
    feats = cases.features((len(cases.BOXES), grid * grid, cin), cases.SEED_FEATS)
    out, cells, count = O.region_forward(f32(sd), feats, cases.BOXES)
    assert np.array_equal(cells.numpy(), g[f"region_{tag}_cells"])          # integer side: bit exact
    assert rel(out, g[f"region_{tag}_out"]) <= 1e-5
    if tag == "g16":  # SURVEY.md 8(c) known answer: box 1 -> rows 0-7 x cols 4-7 of the 16x16 grid
        m = cells[1].reshape(16, 16)
        assert int(count[1]) == 32 and bool(m[0:8, 4:8].all()) and int(count[3]) == 1 and int(count[4]) == 0


def test_projector():
    g = np.load(os.path.join(G, "region_projector.npz"))
    sd = synth.projector_state(cases.MM_HIDDEN, cases.LLM["hidden_size"], synth.make_generator(cases.SEED_PROJ), **cases.MLP_INIT)
    x = cases.features((37, cases.MM_HIDDEN), cases.SEED_FEATS + 1)
    assert rel(O.projector_forward(f32(sd), x), g["projector_out"]) <= 1e-5


def oracle_weights():
    w = {
        "image_tower": f32(synth.vit_state(cases.VIT_IMAGE, synth.make_generator(cases.SEED_VIT), **cases.VIT_INIT)),
        "video_tower": f32(synth.vit_state(cases.VIT_VIDEO, synth.make_generator(cases.SEED_VIT), **cases.VIT_INIT)),
        "projector": f32(synth.projector_state(cases.MM_HIDDEN, cases.LLM["hidden_size"], synth.make_generator(cases.SEED_PROJ), **cases.MLP_INIT)),
        "region": f32(synth.region_state(cases.MM_HIDDEN, cases.LLM["hidden_size"], synth.make_generator(cases.SEED_REGION), **cases.MLP_INIT)),
        "llama": f32(synth.llama_state(cases.LLM, synth.make_generator(cases.SEED_LLM), **cases.LLM_INIT)),
    }
    cfgs = {"image": cases.VIT_IMAGE, "video": cases.VIT_VIDEO, "llama": cases.LLM}
    return w, cfgs


@pytest.mark.parametrize("name", list(cases.glue_cases()))
def test_glue_and_prefill_logits(name):
    g = np.load(os.path.join(G, "glue_llm.npz"))
    case = cases.glue_cases()[name]
    w, cfgs = oracle_weights()
    logits, embeds, mask, pos = O.multimodal_forward(w, cfgs, case["input_ids"], case["attention_mask"], case["images"],
                                                      case["regions"], case.get("max_length"), case.get("padding_side", "right"))
    ref_e, ref_l, ref_m = g[f"{name}_embeds"], g[f"{name}_logits"], g[f"{name}_mask"]
    assert tuple(embeds.shape) == ref_e.shape                     # spliced layout: exact
    assert np.array_equal(mask.numpy().astype(np.int32), ref_m)
    assert rel(embeds, ref_e) <= 1e-5
    valid = torch.as_tensor(ref_m).bool()
    assert rel(logits[valid], torch.as_tensor(ref_l)[valid]) <= 1e-4
    assert torch.equal(logits[valid].argmax(-1), torch.as_tensor(ref_l)[valid].argmax(-1))


def test_glue_random_layouts_against_the_reference():
    """24 seeded random batches (cases.random_glue_cases) through the REFERENCE's prepare_inputs_labels_for_multimodal
    (tests/golden/glue_random.npz, make_golden.gen_glue_random): the oracle must produce the same spliced layout -- masks and
    position ids exact, every row of the embeddings (a fixed random projection of them is stored) to fp32 accuracy."""
    g = np.load(os.path.join(G, "glue_random.npz"))
    w, cfgs = oracle_weights()
    proj = cases.glue_projection(cases.LLM["hidden_size"])
    n = 0
    for name, case in cases.random_glue_cases().items():
        embeds, mask, pos = O.multimodal_prepare(w, cfgs, case["input_ids"], case["attention_mask"], case["images"], case["regions"],
                                                 case.get("max_length"), case.get("padding_side", "right"))
        ref_m, ref_p = g[f"{name}_mask"], g[f"{name}_proj"]
        assert np.array_equal(mask.numpy().astype(np.int32), ref_m), name
        got = (embeds.double() @ proj).numpy()
        assert got.shape == ref_p.shape, name
        assert np.abs(got - ref_p).max() <= 1e-4 * max(np.abs(ref_p).max(), 1e-6), (name, np.abs(got - ref_p).max())
        if f"{name}_pos" in g:
            valid = ref_m.astype(bool)
            assert np.array_equal(pos.numpy()[valid], g[f"{name}_pos"][valid]), name
        n += 1
    assert n == 24


GREEDY_TINY = ("image_region", "video", "text_only", "video_image_trunc")


@pytest.mark.parametrize("name", GREEDY_TINY)
def test_greedy_ids_against_the_reference(name):
    """tests/golden/greedy.npz (make_golden.gen_greedy): a hand-written greedy loop over the REFERENCE's own forward. The oracle's
    greedy_generate -- KV cache + the decode-step mask / position fix-up of llava_arch.py:196-205 -- must produce the same ids
    at every step (fp32 on both sides: margins are far above the summation-order noise) and the same logits rows."""
    g = np.load(os.path.join(G, "greedy.npz"))
    case = cases.glue_cases()[name]
    w, cfgs = oracle_weights()
    embeds, mask, pos = O.multimodal_prepare(w, cfgs, case["input_ids"], case["attention_mask"], case["images"], case["regions"],
                                             case.get("max_length"), case.get("padding_side", "right"))
    ref_ids, ref_rows, margin = g[f"{name}_ids"], g[f"{name}_logits"], g[f"{name}_margin"]
    n = len(ref_ids)
    assert float(margin.min()) > 1e-3                       # every stored step is decidable in fp32
    ids = O.greedy_generate(w["llama"], cfgs["llama"], embeds, mask.long(), pos, n)
    assert ids[0].tolist() == ref_ids.tolist()
    # teacher-forced logits of every step through the cached decode path
    emb = w["llama"]["model.embed_tokens.weight"]
    logits, past = O.llama_forward(w["llama"], cfgs["llama"], embeds, pos, mask.long())
    rows = [logits[0, -1]]
    m = mask.long()
    for t in range(n - 1):
        m = torch.cat([m, torch.ones((1, 1), dtype=m.dtype)], 1)
        p = m.sum(1, keepdim=True) - 1
        logits, past = O.llama_forward(w["llama"], cfgs["llama"], emb[int(ref_ids[t])].view(1, 1, -1), p, m, past)
        rows.append(logits[0, -1])
    assert rel(torch.stack(rows), ref_rows) <= 1e-4


def test_padded_batch_greedy_ids_against_the_reference():
    """tests/golden/greedy_batch.npz (make_golden.gen_greedy_batch): the reference's padded-batch generate() restated over its own
    forward, next to every sample run alone. The oracle reproduces BOTH: ids_mask=... follows the padded-batch behaviour to the
    letter (pad rows attended, first token read from the last column), ids_mask=None gives the batch-1 rows -- and the two differ
    for the shorter sample, which is the difference vitron_amd's packed generate() keeps."""
    g = np.load(os.path.join(G, "greedy_batch.npz"))
    case = cases.glue_cases()["batch_pad"]
    w, cfgs = oracle_weights()
    embeds, mask, pos = O.multimodal_prepare(w, cfgs, case["input_ids"], case["attention_mask"], case["images"], case["regions"])
    assert mask.long().sum(1).tolist() == g["batch_pad_spliced_lengths"].tolist()
    n = g["batch_pad_padded_ids"].shape[1]
    padded = O.greedy_generate(w["llama"], cfgs["llama"], embeds, mask.long(), pos, n, ids_mask=case["attention_mask"].long())
    assert padded.tolist() == g["batch_pad_padded_ids"].tolist()
    alone = O.greedy_generate(w["llama"], cfgs["llama"], embeds, mask.long(), pos, n)
    differing = 0
    for b in range(embeds.shape[0]):
        assert float(g[f"batch_pad_alone{b}_margin"].min()) > 1e-3
        L = int(mask[b].sum())
        one = O.greedy_generate(w["llama"], cfgs["llama"], embeds[b:b + 1, :L], torch.ones(1, L, dtype=torch.long),
                                torch.arange(L).unsqueeze(0), n)
        assert one[0].tolist() == g[f"batch_pad_alone{b}_ids"].tolist()
        differing += int(one[0].tolist() != g["batch_pad_padded_ids"][b].tolist())
    assert differing == 1                                    # the shorter sample; the longest is the same either way
    longest = int(mask.long().sum(1).argmax())
    assert alone[longest].tolist() == g[f"batch_pad_alone{longest}_ids"].tolist()


def _left_padded_batch():
    """Two (text + image + box) samples of different text lengths, LEFT-padded in id space as a tokenizer with padding_side = 'left' hands them
    over (zeros, then ones), each with ONE <image> and one <objs>: the padding in id space equals the padding in the spliced space."""
    from tests.golden.cases import SEED_IDS, SEED_PIX, BOXES, _ids, pixels
    a = [1] + _ids(1, SEED_IDS + 40) + [-200, -300] + _ids(6, SEED_IDS + 41)
    b = [1, -200, -300] + _ids(2, SEED_IDS + 42)
    L = max(len(a), len(b))
    ids = torch.tensor([[0] * (L - len(a)) + a, [0] * (L - len(b)) + b])
    am = torch.tensor([[0] * (L - len(a)) + [1] * len(a), [0] * (L - len(b)) + [1] * len(b)])
    return dict(input_ids=ids, attention_mask=am, images=[pixels((3, 56, 56), SEED_PIX + 31), pixels((3, 56, 56), SEED_PIX + 32)],
                regions=[BOXES[2], BOXES[0]], solo=[a, b])


def test_left_padded_batch_with_equal_visual_rows_is_every_sample_alone():
    """What vitron_amd.generate(padded_batch=True) relies on for tokenizer_padding_side = 'left' (reference llava_arch.py:379-386 pads the
    spliced rows on the left; :196-205 extends the ids-length mask with ones at every decode step): when every sample carries the same
    number of visual rows, the zeros of the ids-length mask cover exactly the sample's pad rows and sum(mask) - 1 is its own length + step, so
    the reference's padded-batch loop returns, for every sample, the ids it returns for that sample ALONE. Shown on the oracle's restatement
    of that loop (pinned on the reference's ids for the right-padded case above)."""
    case = _left_padded_batch()
    w, cfgs = oracle_weights()
    n = 8
    embeds, mask, pos = O.multimodal_prepare(w, cfgs, case["input_ids"], case["attention_mask"], case["images"], case["regions"], padding_side="left")
    S = embeds.shape[1]
    assert (S - mask.long().sum(1)).tolist() == (case["attention_mask"].shape[1] - case["attention_mask"].sum(1)).tolist()
    padded = O.greedy_generate(w["llama"], cfgs["llama"], embeds, mask.long(), pos, n, ids_mask=case["attention_mask"].long())
    for b, ids in enumerate(case["solo"]):
        e1, m1, p1 = O.multimodal_prepare(w, cfgs, torch.tensor([ids]), None, [case["images"][b]], [case["regions"][b]])
        one = O.greedy_generate(w["llama"], cfgs["llama"], e1, m1.long(), p1, n)
        assert padded[b].tolist() == one[0].tolist(), b


def _proj3_state():
    H = cases.LLM["hidden_size"]
    g = synth.make_generator(cases.SEED_PROJ + 3)
    psd = synth.projector_state(cases.MM_HIDDEN, H, g, **cases.MLP_INIT)
    extra = synth.projector_state(H, H, g, **cases.MLP_INIT)
    psd["4.weight"], psd["4.bias"] = extra["2.weight"], extra["2.bias"]
    return psd


def test_f1_branches_clip_tower_and_mlp3x_projector():
    """tests/golden/f1.npz (make_golden.gen_f1): the reference's CLIPVisionTower (clip_encoder.py:7-78) around a transformers
    CLIPVisionModel, select_feature 'patch' / 'cls_patch', and build_vision_projector('mlp3x_gelu') (builder.py:39-46)."""
    g = np.load(os.path.join(G, "f1.npz"))
    cfg = cases.CLIP_TOWER
    sd = synth.vit_state(cfg, synth.make_generator(cases.SEED_VIT + 2), **cases.VIT_INIT)
    assert synth.checksum(sd) == pytest.approx(float(g["clip_checksum"]), rel=1e-12)
    x = cases.pixels(cases.CLIP_TOWER_SHAPE, cases.SEED_PIX + 21)
    h = O.vit_forward(f32(sd), cfg, x, num_layers=cfg["num_hidden_layers"] - 1)      # hidden_states[-2]
    assert rel(h, g["clip_cls_patch"]) <= 1e-5 and rel(h[:, 1:], g["clip_patch"]) <= 1e-5
    assert rel(O.tower_features(f32(sd), cfg, x, -2), g["clip_patch"]) <= 1e-5
    assert rel(h[0:1], g["clip_cls_patch_list0"]) <= 1e-5 and rel(h[2:3, 1:], g["clip_patch_list1"]) <= 1e-5
    psd = _proj3_state()
    assert synth.checksum(psd) == pytest.approx(float(g["proj3_checksum"]), rel=1e-12)
    xp = cases.features((cases.PROJ3_ROWS, cases.MM_HIDDEN), cases.SEED_FEATS + 5)
    assert rel(O.projector_forward(f32(psd), xp), g["proj3_out"]) <= 1e-5


def test_mx4_quantiser_and_level3_emulation():
    """The oracle's restatement of precise level 3 (mx4_quant / _lin_mx / llama_forward(precise_qk = 3)): the quantiser's known answers -- exponent
    rule, ties to the even mantissa, nothing clipped, the code map -- and that the emulated mode lands closer to fp32 than the standard
    emulation of the same storage format (the 4-bit image carries most of what the 16-bit store drops)."""
    x = torch.tensor([[0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, -5.0, 6.0, 0.24, 0.26, -0.0] + [0.0] * 20])
    deq, q, e = O.mx4_quant(x, 32)
    assert int(e) == 0 and q[0, :12].tolist() == [0.0, 1.0, 1.0, 2.0, 2.0, 4.0, 4.0, -4.0, 6.0, 0.0, 0.5, -0.0]
    assert O.mx4_codes(q)[0, :12].tolist() == [0, 2, 2, 4, 4, 6, 6, 14, 7, 0, 1, 8]
    for amax, want in ((6.0, 0), (6.0001, 1), (3.0001, 0), (3.0, -1), (1.5, -2), (0.02, -8), (0.0, -126), (1e38, 124)):
        assert int(O.mx4_exponent(torch.tensor(amax))) == want, (amax, want)
    g = torch.Generator().manual_seed(3)
    v = torch.randn((64, 256), generator=g) * torch.rand((64, 1), generator=g) * 4
    for blk in (32, None):
        d, qq, ee = O.mx4_quant(v, blk)
        assert float((qq.abs().amax())) <= 6.0 and float((d - v).norm() / v.norm()) < 0.16          # e2m1 against a power-of-two scale: ~11-13 %
        sc = torch.ldexp(torch.ones_like(ee, dtype=torch.float32), ee)
        assert bool(((v.reshape(64, -1, 32 if blk else 256).abs().amax(-1) / sc) <= 6.0).all())     # nothing clips
    # a Linear of level 3 against the exact product: the remainder's image removes most of the 16-bit store's error
    w = (torch.randn((96, 256), generator=g) * 0.05).to(torch.bfloat16).float()
    exact = v.double() @ w.double().t()
    for emu in ("fp16", True):
        plain = (O._r(v, emu).double() @ w.double().t())
        mx = O._lin_mx(v, w, emu).double()
        assert float((mx - exact).norm()) < 0.3 * float((plain - exact).norm())
    # two decoder layers at a width the mode accepts (hidden % 512 == 0): level 3's emulation is closer to fp32 than the standard emulation
    cfg = dict(cases.LLM, hidden_size=512, intermediate_size=1408, num_attention_heads=4, num_hidden_layers=2, vocab_size=320)
    sd = synth.llama_state(cfg, synth.make_generator(77), w_std=0.02)
    sd = {k: t.float() for k, t in sd.items()}
    emb = torch.randn((1, 96, 512), generator=g) * 0.5
    ref = O.llama_forward(sd, cfg, emb)[0]
    std = O.llama_forward(sd, cfg, emb, emulate_bf16="fp16")[0]
    l3 = O.llama_forward(sd, cfg, emb, emulate_bf16="fp16", precise_qk=3)[0]
    d_std, d_l3 = float((std - ref).norm() / ref.norm()), float((l3 - ref).norm() / ref.norm())
    assert d_l3 < 0.8 * d_std, (d_l3, d_std)
