"""CPU tests of the host-side logic of the product (no GPU, no kernels launched): prompt plumbing, the integer
splice plan, region slice resolution, weight packing layouts, and that the C-ABI library loads and exports every
symbol include/vitron_hip.h declares."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import vitron_oracle as O
from tests.golden import cases
from tests.test_oracle_golden import StubTok

G = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mm_utils_image_helpers_match_reference_goldens():
    """load_image_from_base64 / expand2square / process_images (reference mm_utils.py:48-77) against the reference's own functions' outputs on
    the pictures of tests/golden/mm_images_cases.py (tests/golden/make_golden_mm_images.py)."""
    import types
    from tests.golden import mm_images_cases as MC
    from vitron_amd import mm_utils
    g = np.load(os.path.join(G, "mm_utils_images.npz"))
    pics = [MC.picture(w, h) for w, h in MC.SIZES]
    proc = MC.StubProcessor()
    for i, im in enumerate(pics):
        sq = mm_utils.expand2square(im, MC.FILL)
        assert sq.size == (max(im.size), max(im.size)) and np.array_equal(np.asarray(sq), g[f"square_{i}"])
        assert np.array_equal(np.asarray(mm_utils.load_image_from_base64(MC.png_base64(im))), g[f"b64_{i}"])
    assert mm_utils.expand2square(pics[2], MC.FILL) is pics[2]                       # a square picture comes back as it is
    pad = mm_utils.process_images(pics, proc, types.SimpleNamespace(image_aspect_ratio="pad"))
    assert isinstance(pad, torch.Tensor) and np.array_equal(pad.numpy(), g["process_pad"])
    assert np.array_equal(mm_utils.process_images(pics, proc, types.SimpleNamespace(image_aspect_ratio=None)).numpy(), g["process_plain"])
    assert np.array_equal(mm_utils.process_images(pics[:2], proc, types.SimpleNamespace()).numpy(), g["process_missing_attr"])

    class Ragged(MC.StubProcessor):                                                  # shapes that differ stay a list (mm_utils.py:75-76)
        def preprocess(self, image, return_tensors=None):
            t = self._one(image)
            return {"pixel_values": [t[:, : min(image.size[0], 16)]]}
    rag = mm_utils.process_images([pics[0], pics[3]], Ragged(), types.SimpleNamespace(image_aspect_ratio="pad"))
    assert isinstance(rag, list) and len(rag) == 2


def test_mm_utils_match_reference_goldens():
    from vitron_amd import mm_utils
    g = np.load(os.path.join(G, "mm_utils.npz"))
    tok = StubTok()
    for i, p in enumerate(cases.PROMPTS):
        assert mm_utils.tokenizer_image_token(p, tok) == g[f"image_token_{i}"].tolist()
        assert mm_utils.tokenizer_image_region_token(p, tok) == g[f"region_token_{i}"].tolist()
        assert mm_utils.tokenizer_image_region_token(p, tok, return_tensors="pt").tolist() == g[f"region_token_{i}"].tolist()
    for i, (r, isz, tsz) in enumerate(cases.REGION_RESCALE):
        assert mm_utils.preprocess_region(r, isz, tsz) == g[f"preprocess_region_{i}"].tolist()
    with pytest.raises(ValueError):
        mm_utils.tokenizer_image_token("a", tok, return_tensors="np")
    assert mm_utils.get_model_name_from_path("/a/b/checkpoint-12/") == "b_checkpoint-12"


def test_prompt_tokenisation_randomised_against_the_oracle():
    """Random prompts built from text pieces, '<image>' and '<objs>' (adjacent, leading, trailing, none): the product's
    tokenizer_image_token / tokenizer_image_region_token return the ids the oracle's restatement of mm_utils.py:80-117 returns."""
    import random
    from vitron_amd import mm_utils
    rnd = random.Random(99)
    tok = StubTok()
    pieces = ["<image>", "<objs>", "a", "bc", " ", "\n", "x y", ""]
    for _ in range(300):
        prompt = "".join(rnd.choice(pieces) for _ in range(rnd.randint(0, 8)))
        assert mm_utils.tokenizer_image_token(prompt.replace("<objs>", "o"), tok) == O.tokenizer_image_token(prompt.replace("<objs>", "o"), tok)
        assert mm_utils.tokenizer_image_region_token(prompt, tok) == O.tokenizer_image_region_token(prompt, tok), prompt
        for first in (True, False):
            assert mm_utils.tokenizer_image_token(prompt.replace("<objs>", ""), tok, is_first=first) == \
                O.tokenizer_image_token(prompt.replace("<objs>", ""), tok, is_first=first), (prompt, first)


def test_keywords_stopping_criteria_semantics():
    """reference mm_utils.py:146-177: stop on an id-tail match or on the keyword appearing in the decoded tail."""
    from vitron_amd.mm_utils import KeywordsStoppingCriteria

    class Tok(StubTok):
        def batch_decode(self, ids, skip_special_tokens=True):
            return ["".join(chr(int(t) - 100) for t in row if int(t) != 1) for row in ids]

    tok = Tok()
    prompt = torch.tensor([tok("hello").input_ids])
    crit = KeywordsStoppingCriteria(["</s>", "##"], tok, prompt)
    assert [k.tolist() for k in crit.keyword_ids] == [tok("</s>").input_ids[1:], tok("##").input_ids[1:]]   # BOS stripped
    assert crit.max_keyword_len == 4 and crit.start_len == prompt.shape[1]
    grow = lambda text: torch.cat([prompt, torch.tensor([tok(text).input_ids[1:]])], 1)
    assert not crit(prompt, None)                      # nothing generated yet
    assert not crit(grow("abc"), None)
    assert crit(grow("abc##"), None)                   # id-tail match
    assert crit(grow("ab##c"), None)                   # only in the decoded window (4 new tokens)
    assert not crit(grow("##abcd"), None)              # keyword scrolled out of the window
    assert crit(grow("x</s>"), None)
    both = torch.cat([grow("ab##"), grow("abcd")], 0)  # batch: every row must be finished
    assert not crit(both, None)
    assert crit(torch.cat([grow("ab##"), grow("a##b")], 0), None)
    assert crit.call_for_batch(both, None) is True     # first row only


def test_output_parser_matches_reference_goldens():
    """vitron_amd.output_parser.parse_model_output against what the reference's own functions (app.py:345-395) returned for the
    same strings (tests/golden/output_parser.json, made by make_golden.gen_output_parser), plus the tuple protocol callers use."""
    import json
    from vitron_amd.output_parser import parse_model_output
    gold = json.load(open(os.path.join(G, "output_parser.json")))
    assert [g["text"] for g in gold] == cases.MODEL_OUTPUTS
    for g in gold:
        got = parse_model_output(g["text"])
        assert list(got) == g["parsed"], (g["text"], list(got), g["parsed"])
    output, module, instruction, region = parse_model_output(cases.MODEL_OUTPUTS[1])      # app.py:572 unpacks four values
    assert (module, instruction, region) == ("B", ["left one"], "[10, 20, 110, 220]") and output == "I segmented it.  done"


def test_bench_algorithmic_work_matches_survey_accounting():
    """bench.py's roofline numerator follows SURVEY.md 8(d): C3 (8 x 336 px frames + 512 tokens, S = 5120) is 77.20 TFLOP
    (ViT 3.82, projector 0.19, decoder linears 66.31, attention 6.87); C3 at 224 px 36.60."""
    import bench
    fl = bench.algorithmic_flops(5120, 4608, 8, 577, 576)
    assert fl["total"] / 1e12 == pytest.approx(77.20, abs=0.01)
    assert fl["vit"] / 1e12 == pytest.approx(3.82, abs=0.01) and fl["projector"] / 1e12 == pytest.approx(0.19, abs=0.005)
    assert fl["llm_linear"] / 1e12 == pytest.approx(66.31, abs=0.01) and fl["llm_attention"] / 1e12 == pytest.approx(6.87, abs=0.01)
    assert bench.algorithmic_flops(2560, 2048, 8, 257, 256)["total"] / 1e12 == pytest.approx(36.60, abs=0.01)


def test_constants_match_reference_values():
    from vitron_amd import constants as c
    assert (c.IGNORE_INDEX, c.IMAGE_TOKEN_INDEX, c.OBJS_TOKEN_INDEX) == (-100, -200, -300)
    assert (O.IGNORE_INDEX, O.IMAGE_TOKEN_INDEX, O.OBJS_TOKEN_INDEX) == (-100, -200, -300)


def _materialise(plan, tok, vis, reg):
    rows = []
    for kind, idx in plan:
        rows.append({0: lambda: tok[idx], 1: lambda: vis[idx], 2: lambda: reg[idx], 3: lambda: torch.zeros(tok.shape[1])}[kind]())
    return torch.stack(rows)


@pytest.mark.parametrize("name", list(cases.glue_cases()))
def test_splice_plan_matches_reference_layout(name):
    """The integer plan reproduces the reference's spliced layout: same shape/mask as the golden produced by the
    reference's prepare_inputs_labels_for_multimodal, and the same rows as the oracle's list surgery."""
    from vitron_amd.model.llava_arch import build_splice_plan
    g = np.load(os.path.join(G, "glue_llm.npz"))
    case = cases.glue_cases()[name]
    H, P, T = 8, 16, cases.VIT_VIDEO["num_frames"]
    tok = torch.arange(cases.LLM["vocab_size"] * H, dtype=torch.float32).reshape(-1, H)
    blocks, regs, vis_rows, flat_f, flat_r = [], [], 0, [], []
    use_regions = case["regions"] is not None and len(case["regions"]) > 0
    image_idx = [i for i, im in enumerate(case["images"]) if im.dim() == 3]
    nreg = 0
    per = {}
    for j, i in enumerate(image_idx):
        per[i] = ([(vis_rows + j * P, P)], [j])
    vis_rows += len(image_idx) * P
    for j, i in enumerate([i for i, im in enumerate(case["images"]) if im.dim() == 4]):
        per[i] = ([(vis_rows + (j * T + t) * P, P) for t in range(T)], [-1] * T)
    nvid = sum(1 for im in case["images"] if im.dim() == 4)
    vis = -1.0 - torch.arange((vis_rows + nvid * T * P) * H, dtype=torch.float32).reshape(-1, H)
    reg = 1e6 + torch.arange(max(len(image_idx), 1) * H, dtype=torch.float32).reshape(-1, H)
    for i in range(len(case["images"])):
        blocks += per[i][0]
        regs += per[i][1]
    for (first, n), r in zip(blocks, regs):
        flat_f.append(vis[first:first + n])
        flat_r.append(reg[r:r + 1] if r >= 0 else None)
    am = None if case["attention_mask"] is None else case["attention_mask"].tolist()
    plan, mask, pos, lengths = build_splice_plan(case["input_ids"].tolist(), am, blocks, regs if use_regions else None,
                                                 case.get("max_length"), case.get("padding_side", "right"))
    ref_m = g[f"{name}_mask"]
    assert np.array_equal(np.array(mask, dtype=np.int32), ref_m)                          # vs the REFERENCE
    e, m2, p2 = O.splice_embeddings(case["input_ids"], case["attention_mask"], tok, flat_f, flat_r if use_regions else None,
                                    case.get("max_length"), case.get("padding_side", "right"))
    got = torch.stack([_materialise(p, tok, vis, reg) for p in plan])
    assert torch.equal(got, e) and torch.equal(torch.tensor(mask).bool(), m2) and torch.equal(torch.tensor(pos), p2)
    assert lengths == [int(r.sum()) for r in ref_m]


def test_splice_plan_randomised_against_the_oracle():
    """200 random batches (sentinels at random places, images / video frames of mixed block sizes, optional regions, optional
    masks with holes, truncation, left / right padding): the integer plan must move exactly the rows the oracle's restatement of
    the reference's list surgery moves -- embeddings, masks and position ids bit for bit."""
    import random
    from vitron_amd.model.llava_arch import build_splice_plan
    rnd = random.Random(1234)
    H, V = 4, 64
    tok = torch.arange(V * H, dtype=torch.float32).reshape(V, H)
    for trial in range(200):
        B = rnd.randint(1, 4)
        L = rnd.randint(1, 24)
        use_regions = rnd.random() < 0.5
        ids, masks = [], []
        blocks, regs, flat_f, flat_r = [], [], [], []
        vis_rows, reg_rows = 0, 0
        for b in range(B):
            n_img = rnd.choice([0, 0, 1, 1, 2, 3])
            row = [rnd.randint(0, V - 1) for _ in range(L)]
            slots = rnd.sample(range(L), min(L, n_img + (rnd.randint(0, 2) if (use_regions and n_img) else 0)))
            slots.sort()
            img_slots = slots[:n_img]
            for s_ in img_slots:
                row[s_] = -200
            for s_ in slots[n_img:]:
                if s_ > min(img_slots):          # an <objs> before any <image> would index feature -1 in the reference
                    row[s_] = -300
            m = [1] * L
            if rnd.random() < 0.4:               # a mask with holes, but never over a sentinel (keeps the feature count in sync)
                for j in range(L):
                    if row[j] >= 0 and rnd.random() < 0.3:
                        m[j] = 0
            ids.append(row)
            masks.append(m)
            kept_imgs = sum(1 for j in range(L) if row[j] == -200 and m[j])
            for _ in range(max(kept_imgs, 1)):   # a sample without <image> still consumes one feature slot
                n = rnd.choice([1, 3, 5])
                blocks.append((vis_rows, n))
                has_reg = rnd.random() < 0.7
                regs.append(reg_rows if has_reg else -1)
                vis_rows += n
                reg_rows += 1 if has_reg else 0
        vis = -1.0 - torch.arange(max(vis_rows, 1) * H, dtype=torch.float32).reshape(-1, H)
        reg = 1e6 + torch.arange(max(reg_rows, 1) * H, dtype=torch.float32).reshape(-1, H)
        for (first, n), r in zip(blocks, regs):
            flat_f.append(vis[first:first + n])
            flat_r.append(reg[r:r + 1] if r >= 0 else torch.zeros((1, H)))
        if use_regions and any(r < 0 for r in regs):   # a region-less feature: give it a real row so both sides index the same thing
            regs2, reg2 = [], [reg]
            extra = reg.shape[0]
            for i, r in enumerate(regs):
                if r < 0:
                    regs2.append(extra)
                    extra += 1
                else:
                    regs2.append(r)
            reg = torch.cat([reg, torch.zeros((extra - reg.shape[0], H))])
            regs = regs2
        max_length = rnd.choice([None, None, rnd.randint(1, 30)])
        side = rnd.choice(["right", "left"])
        am = None if all(all(m) for m in masks) and rnd.random() < 0.5 else masks
        t_ids = torch.tensor(ids)
        t_am = None if am is None else torch.tensor(am)
        plan, mask, pos, lengths = build_splice_plan(ids, am, blocks, regs if use_regions else None, max_length, side)
        e, m2, p2 = O.splice_embeddings(t_ids, t_am, tok, flat_f, flat_r if use_regions else None, max_length, side)
        got = torch.stack([_materialise(p_, tok, vis, reg) for p_ in plan])
        assert torch.equal(got, e), (trial, ids, am, blocks, regs)
        assert torch.equal(torch.tensor(mask).bool(), m2) and torch.equal(torch.tensor(pos), p2), trial
        assert lengths == [int(r.sum()) for r in m2], trial


def test_splice_plan_random_layouts_match_the_reference_masks():
    """The product's integer plan on the 24 random batches the REFERENCE was run on (tests/golden/glue_random.npz): same padded
    shape, same attention mask, same position ids as the reference's prepare_inputs_labels_for_multimodal returned."""
    from vitron_amd.model.llava_arch import build_splice_plan
    g = np.load(os.path.join(G, "glue_random.npz"))
    P, T = 16, cases.VIT_VIDEO["num_frames"]
    for name, case in cases.random_glue_cases().items():
        use_regions = case["regions"] is not None and len(case["regions"]) > 0
        blocks, regs, rows, nimg = [], [], 0, 0
        for im in case["images"]:                       # flat feature list in `images` order: one block per image, T per clip
            if im.dim() == 3:
                blocks.append((rows, P))
                regs.append(nimg)
                nimg += 1
                rows += P
            else:
                for _ in range(T):
                    blocks.append((rows, P))
                    regs.append(-1)
                    rows += P
        am = None if case["attention_mask"] is None else case["attention_mask"].tolist()
        plan, mask, pos, lengths = build_splice_plan(case["input_ids"].tolist(), am, blocks, regs if use_regions else None,
                                                     case.get("max_length"), case.get("padding_side", "right"))
        ref_m = g[f"{name}_mask"]
        assert np.array_equal(np.array(mask, dtype=np.int32), ref_m), name
        assert lengths == [int(r.sum()) for r in ref_m], name
        if f"{name}_pos" in g:
            valid = ref_m.astype(bool)
            assert np.array_equal(np.array(pos)[valid], g[f"{name}_pos"][valid]), name


def test_splice_plan_errors_and_quirks():
    from vitron_amd.model.llava_arch import build_splice_plan
    # a sample without <image> still consumes a feature slot (llava_arch.py:317-324)
    plan, mask, pos, lens = build_splice_plan([[1, 5, 6], [1, -200, 7]], None, [(0, 2), (2, 2)], None)
    assert plan[1][1:3] == [(1, 2), (1, 3)] and lens == [3, 4] and mask[0] == [1, 1, 1, 0]
    with pytest.raises(ValueError):
        build_splice_plan([[1, -200, -200]], None, [(0, 2)], None)          # more sentinels than features
    with pytest.raises(ValueError):
        build_splice_plan([[1, -200, -300]], None, [(0, 2)], None)          # <objs> without regions
    with pytest.raises(ValueError):
        build_splice_plan([[1, -200, -300]], None, [(0, 2)], [-1])          # <objs> bound to a video frame
    # empty prompt edge: zero-length sample
    plan, mask, pos, lens = build_splice_plan([[1]], [[0]], [(0, 2)], None)
    assert lens == [0]


def test_region_slice_resolution_is_python_exact():
    from vitron_amd.engine import resolve_region_slices
    boxes = cases.BOXES + [[-10, 5.9, 300, 7.2], [50, 60, 40, 70], [223.99999, 0, 224.5, 1e9]]
    got = resolve_region_slices(boxes, 224)
    for (x1, y1, x2, y2), (r0, r1, c0, c1) in zip(boxes, got):
        m = torch.zeros(224, 224)
        m[int(x1):int(x2), int(y1):int(y2)] = 1
        rows = m.any(1).nonzero().flatten()
        cols = m.any(0).nonzero().flatten()
        exp = torch.zeros(224, 224)
        exp[r0:r1, c0:c1] = 1
        assert torch.equal(m, exp), (x1, y1, x2, y2)
    # 500 random boxes (negative, inverted, fractional, far outside, both image sizes) against the oracle's mask
    import random
    rnd = random.Random(7)
    for size in (224, 336):
        boxes = [[rnd.uniform(-1.5, 1.5) * size * rnd.choice([0.01, 0.5, 1, 3]) for _ in range(4)] for _ in range(250)]
        ref = O.region_mask(boxes, size)
        for b, (r0, r1, c0, c1), m in zip(boxes, resolve_region_slices(boxes, size), ref):
            exp = torch.zeros(size, size)
            exp[r0:r1, c0:c1] = 1
            assert torch.equal(m, exp), (size, b)


def test_weight_packing_layouts():
    from vitron_amd.engine import interleave_gate_up, merge_lora
    g = torch.arange(64 * 3, dtype=torch.float32).reshape(64, 3)
    u = -g
    w = interleave_gate_up(g, u)
    assert w.shape == (128, 3)
    assert torch.equal(w[0:16], g[0:16]) and torch.equal(w[16:32], u[0:16]) and torch.equal(w[32:48], g[16:32])
    base = torch.randn(8, 6)
    a, b = torch.randn(2, 6), torch.randn(8, 2)
    sd = {"encoder.layers.0.self_attn.q_proj.base_layer.weight": base, "encoder.layers.0.self_attn.q_proj.base_layer.bias": torch.zeros(8),
          "encoder.layers.0.self_attn.q_proj.lora_A.default.weight": a, "encoder.layers.0.self_attn.q_proj.lora_B.default.weight": b,
          "pre_layrnorm.weight": torch.ones(6)}
    m = merge_lora(sd, lora_alpha=16.0)
    assert set(m) == {"encoder.layers.0.self_attn.q_proj.weight", "encoder.layers.0.self_attn.q_proj.bias", "pre_layrnorm.weight"}
    assert torch.allclose(m["encoder.layers.0.self_attn.q_proj.weight"], base + 8.0 * (b @ a))


@pytest.mark.parametrize("operand", ["bf16", "fp16"])
def test_cabi_library_loads_and_exports_header_symbols(operand):
    """Every function declared in include/vitron_hip.h is exported by libvitron_hip.so AND by libvitron_hip_f16.so (the same ABI in
    the two operand formats) and bound in _lib.SIGNATURES."""
    from vitron_amd import _lib
    lib = _lib.load(operand=operand)
    hdr = open(os.path.join(ROOT, "include", "vitron_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t)\s+(vt_\w+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.vt_version() == _lib.ABI_VERSION
    assert lib.vt_operand_format() == {"bf16": _lib.OPERAND_BF16, "fp16": _lib.OPERAND_FP16}[operand]
    # error convention: bad arguments come back as a negative status + message, nothing throws, no GPU needed
    st = lib.vt_gemm_bf16(None, 8, None, 8, None, 8, None, 4, 4, 8, 0, 0, None, None)
    assert st == -1 and "null" in _lib.last_error(lib)
    assert lib.vt_projector_workspace_bytes(100, 4096) >= 100 * 4096 * 2


def test_operand_format_selection_is_by_dtype():
    """A tensor's dtype picks the library build (bf16 -> libvitron_hip.so, fp16 -> libvitron_hip_f16.so); anything else is refused;
    the two handles are distinct objects with their own error state."""
    from vitron_amd import _lib
    assert _lib.operand_of(torch.bfloat16) == "bf16" and _lib.operand_of(torch.float16) == "fp16" and _lib.operand_of("fp16") == "fp16"
    assert _lib.operand_of(torch.zeros(1, dtype=torch.float16)) == "fp16"
    for bad in (torch.float32, torch.int8, "fp8"):
        with pytest.raises(_lib.VitronHipError):
            _lib.operand_of(bad)
    a, b = _lib.lib_for(torch.bfloat16), _lib.lib_for(torch.float16)
    assert a is not b and a is _lib.load(operand="bf16") and b is _lib.load(operand="fp16")
    assert _lib.torch_dtype("fp16") == torch.float16 and _lib.torch_dtype("bf16") == torch.bfloat16
    assert a.vt_projector_workspace_bytes(0, 0) == b.vt_projector_workspace_bytes(0, 0)
    a.vt_gemm_bf16(None, 8, None, 8, None, 8, None, 4, 4, 8, 0, 0, None, None)          # sets a's message only
    assert "null" in _lib.last_error(a)
    from vitron_amd.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM, _resolve_dtype
    assert _resolve_dtype("float16") == torch.float16 and _resolve_dtype("torch.bfloat16") == torch.bfloat16 and _resolve_dtype(None) is None
    m = LlavaLlamaForCausalLM(LlavaConfig(torch_dtype="float16"))                        # a HF config.json carries the dtype as a string
    assert m.dtype == torch.float16 and m.half().dtype == torch.float16 and m.to(torch.bfloat16).dtype == torch.bfloat16


def test_cabi_argument_validation_without_a_gpu():
    """The error convention of include/vitron_hip.h on every family of entry points: invalid arguments come back as a negative
    status with a message in vt_last_error BEFORE anything is launched (so this runs on a box without a GPU), and the size
    queries are pure functions."""
    import ctypes as C
    from vitron_amd import _lib
    lib = _lib.load()
    P = 0x10000                      # an aligned non-null address; never dereferenced because validation fails first

    def bad(status, *needles):
        msg = _lib.last_error()
        assert status < 0, (status, msg)
        assert all(n in msg for n in needles), msg

    bad(lib.vt_gemm_bf16(P, 8, P, 8, P, 8, None, 0, 4, 8, 0, 0, None, None), "empty")
    bad(lib.vt_gemm_bf16(P, 8, P, 8, P, 8, None, 4, 6, 8, 0, 0, None, None), "multiple of 4")
    bad(lib.vt_gemm_bf16(P, 12, P, 8, P, 8, None, 4, 8, 8, 0, 0, None, None), "lda")
    bad(lib.vt_gemm_bf16(P + 2, 8, P, 8, P, 8, None, 4, 8, 8, 0, 0, None, None), "aligned")
    bad(lib.vt_gemm_bf16(P, 64, P, 64, P, 64, None, 128, 64, 40, 0, 2, None, None), "K=40")            # tile kernels: K % 64
    bad(lib.vt_gemm_bf16(P, 64, P, 64, P, 48, None, 128, 48, 64, 0, 0, P, None), "row scale")          # row factor: N % 32
    bad(lib.vt_gemm_bf16(P, 64, P, 64, P, 64, None, 128, 64, 64, 99, 2, None, None), "epilogue")
    bad(lib.vt_gemm_bf16_resid_splitk(P, 256, P, 256, P, 256, None, 128, 256, 256, 4, None, 0, None), "workspace")
    bad(lib.vt_rmsnorm(None, None, P, P, 4, 64, 1e-5, None))
    bad(lib.vt_layernorm(P, None, 0, 0, None, None, P, 4, 64, 1e-5, None))
    bad(lib.vt_argmax(P, 0, 10, 10, P, None), "argmax")
    bad(lib.vt_embed_splice(P, 100, None, 0, None, 0, P, 4, 12, P, None), "multiple of 8")
    bad(lib.vt_embed_splice(P, 100, None, 7, None, 0, P, 4, 16, P, None), "table sizes")                # rows without a table
    bad(lib.vt_decode_feed(P, 12, 100, P, P, None, 0, 0, P, P, P, P, 2, None), "multiple of 8")
    bad(lib.vt_decode_feed(P, 64, 100, P, P, None, 2, 0, P, P, P, P, 2, None), "eos")
    bad(lib.vt_decode_feed(P, 64, 100, None, P, None, 0, 0, P, P, P, P, 2, None), "null")
    bad(lib.vt_flash_attn(P, 384, P, P, P, P, 1, 16, P, 128, 2, 48, 0, 1.0, None))                     # head_dim 48
    bad(lib.vt_attn_temporal(P, P, 1, 8, 16, 0, None))
    bad(lib.vt_sample_top_p(P, 2, 100, 100, 0.0, 0, 0.9, 0, 0, P, None, None))                         # temperature 0
    bad(lib.vt_sample_top_p(P, 2, 100, 100, 1.0, -3, 0.9, 0, 0, P, None, None), "top_k")
    rw = _lib.VtRegionWeights()
    rw.in_dim, rw.out_dim = 1024, 4096
    bad(lib.vt_region_forward(C.byref(rw), P, P, P, 17, 24, 224, P, None, None, P, 1 << 24, None), "B=17")
    bad(lib.vt_region_forward(C.byref(rw), P, P, P, 4, 24, 224, P, None, None, P, 1 << 24, None), "weight pointer")
    m = _lib.VtLlamaModel()
    m.hidden, m.heads, m.head_dim, m.intermediate, m.num_layers, m.vocab = 256, 4, 48, 512, 1, 100
    kv = _lib.VtKvCache()
    bad(lib.vt_llama_forward(C.byref(m), C.byref(kv), P, 4, P, P, 1, 4, 1, 4, P, None, 0, None, None, P, 1 << 20, None), "head_dim")
    m.head_dim = 64
    m.rope_cos = m.rope_sin = P
    ws = lib.vt_llama_workspace_bytes(C.byref(m), 100, 1, 1, 100)
    assert ws > 100 * 256 * 4 and lib.vt_llama_workspace_bytes(C.byref(m), 200, 1, 1, 200) > ws
    bad(lib.vt_llama_forward(C.byref(m), C.byref(kv), P, 100, P, P, 1, 100, 2, 100, P, None, 0, None, None, P, 16, None), "workspace")
    bad(lib.vt_llama_forward(C.byref(m), C.byref(kv), P, 0, P, P, 1, 1, 1, 1, P, None, 0, None, None, P, 1 << 20, None), "empty")
    assert lib.vt_llama_workspace_bytes(None, 1, 1, 1, 1) == 0
    assert lib.vt_attn_decode_scratch_bytes(4, 32, 128, 2048) > lib.vt_attn_decode_scratch_bytes(4, 32, 128, 64) > 0
    assert lib.vt_region_workspace_bytes(4, 1024, 4096) > 0
    buf = C.create_string_buffer(8)                     # message longer than the buffer: truncated, NUL-terminated
    lib.vt_gemm_bf16(None, 8, None, 8, None, 8, None, 4, 4, 8, 0, 0, None, None)
    lib.vt_last_error(buf, 8)
    assert len(buf.value) <= 7


def test_ops_refuse_cpu_tensors():
    from vitron_amd import _lib, ops
    with pytest.raises(_lib.VitronHipError):
        ops.gemm(torch.zeros(4, 8, dtype=torch.bfloat16), torch.zeros(4, 8, dtype=torch.bfloat16))
    from vitron_amd.model import load_pretrained_model
    with pytest.raises(RuntimeError):
        load_pretrained_model("synthetic", None, "x", device="cpu")
    with pytest.raises(NotImplementedError):
        load_pretrained_model("synthetic", None, "x", load_8bit=True)


def test_prefix_cache_signatures_and_page_rounding():
    """vitron_amd.prefix_cache: row signatures separate tokens / visual rows / region rows and depend on the image key;
    reuse is the common prefix capped at len-1 and rounded down to whole 64-token pages; the feature cache is LRU."""
    import numpy as np
    import torch

    from vitron_amd.prefix_cache import (VisualFeatureCache, common_prefix, key64, reusable_tokens, row_signature,
                                         tensor_key)
    plan = np.array([[0, 1], [1, 0], [1, 1], [1, 2], [0, 7], [2, 0], [0, 1], [3, 0]])
    a = row_signature(plan, [(0, 3, key64(("img", 1)))], [key64(("box", 1))])
    b = row_signature(plan, [(0, 3, key64(("img", 2)))], [key64(("box", 1))])
    c = row_signature(plan, [(0, 3, key64(("img", 1)))], [key64(("box", 2))])
    assert a[0] == 1 and a[4] == 7 and a[7] == -1
    assert len(set(a[1:4].tolist())) == 3 and (a[1:4] > 32000).all()
    assert common_prefix(a, b) == 1            # a different image changes every visual row
    assert common_prefix(a, c) == 5            # a different box only changes the region row
    assert common_prefix(a, a) == len(a)
    assert reusable_tokens(np.arange(300), np.arange(300)) == 256      # last row must run -> 299 -> 4 pages
    assert reusable_tokens(np.arange(300), np.arange(130)) == 128
    assert reusable_tokens(np.arange(300), np.arange(64)) == 0         # 63 reusable -> not a whole page
    x = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4).to(torch.bfloat16)
    y = x.clone()
    y[1, 2, 3] += 1
    assert tensor_key(x) == tensor_key(x.clone()) and tensor_key(x) != tensor_key(y)
    assert tensor_key(x) != tensor_key(x.reshape(3, 2, 4))
    cache = VisualFeatureCache(2)
    cache.put("a", torch.zeros(1), None)
    cache.put("b", torch.zeros(1), None)
    assert cache.get("a") is not None
    cache.put("c", torch.zeros(1), None)       # evicts "b" (least recently used)
    assert cache.get("b") is None and cache.get("a") is not None and cache.get("c") is not None


def test_gemm_planner_fills_whole_rounds_of_the_chip():
    """Host logic of the tile-GEMM dispatcher (vt_gemm_plan_query, no launch): one big-tile workgroup per CU, so the planner must
    pick the tile height / row split that fills WHOLE rounds of 256 workgroups on the benchmark's shapes, keep the ping-pong
    kernel where the epilogue is VALU-heavy, and fall back to small tiles when a big-tile grid would cover a fraction of the chip."""
    from vitron_amd import _lib, ops
    W4, W4_320, P4 = _lib.CFG_256x256_W4, _lib.CFG_320x256_W4, _lib.CFG_256x256_P4
    small = (_lib.CFG_64x128, _lib.CFG_128x128)
    # decoder at S = 5120 (BASELINE configs[2]): qkv 3.75 rounds of 256-row tiles = 3 of 320-row tiles; o_proj / down_proj 1.25 -> 1
    assert ops.gemm_plan(5120, 12288, 4096, ops.EPI_BF16) == (W4_320, 0)
    assert ops.gemm_plan(5120, 4096, 4096, ops.EPI_F32_RESID) == (W4_320, 0)
    assert ops.gemm_plan(5120, 4096, 11008, ops.EPI_F32_RESID) == (W4_320, 0)
    assert ops.gemm_plan(5120, 22016, 4096, ops.EPI_SWIGLU_BF16) == (W4, 0)            # 6.72 rounds of 256 vs 6 x 1.25 of 320
    assert ops.gemm_plan(4096, 4096, 4096, ops.EPI_F32_RESID) == (W4, 0)               # exactly one round
    # projector at 8 x 576 visual tokens: 288 tiles of 256 rows = 2 rounds, 240 of 320 rows = 1
    assert ops.gemm_plan(4608, 4096, 4096, ops.EPI_BF16) == (W4_320, 0)
    assert ops.gemm_plan(4608, 4096, 1024, ops.EPI_BF16_GELU) == (W4_320, 0)
    # a VALU-heavy epilogue on a whole-round grid of 256-row tiles stays on the ping-pong kernel
    assert ops.gemm_plan(4096, 4096, 1024, ops.EPI_BF16_QGELU) == (P4, 0)
    # ViT qkv (4616 x 3072): one round either way -- 252 tiles of 224 rows (4704 padded rows) beat 228 of 256 (4864) and 180 taller ones
    W4_224 = _lib.CFG_224x256_W4
    assert ops.gemm_plan(4616, 3072, 1024, ops.EPI_BF16) == (W4_224, 0)
    # a single-image prompt (1088 rows) covers a quarter of the chip with big tiles, and the ViT's N = 1024 projections need 584
    # tiles of 64x128 for 512 slots: 160x128 tiles on the four-deep ring (224 / 232 workgroups, one round)
    W4R = _lib.CFG_160x128_W4
    assert ops.gemm_plan(1088, 4096, 4096, ops.EPI_F32_RESID) == (W4R, 0)
    assert ops.gemm_plan(4616, 1024, 4096, ops.EPI_F32_RESID) == (W4R, 0)
    assert ops.gemm_plan(1088, 12288, 4096, ops.EPI_BF16) == (W4_224, 0)                # 240 big tiles, one round: 5 x 224 rows pad 1088 to 1120, not 1280
    assert ops.gemm_plan(1088, 22016, 4096, ops.EPI_SWIGLU_BF16) == (W4_224, 0)         # 430 tiles, two rounds, each 7/8 as long
    # ... also on multiples of 256 since round 5 measured them (profiles/r5_gemm_cfg_224px.jsonl: the reference-native 224 px shapes): where
    # the ROUND COUNT ties, the shorter tile wins (2048 x 12288: 384 tiles of 256 rows = 2 rounds, 480 of 224 rows = 2 rounds x 7/8); where
    # 256-row tiles fill exactly one round they stay (2048 x 8192)
    assert ops.gemm_plan(2048, 8192, 4096, ops.EPI_BF16) == (W4, 0) and ops.gemm_plan(2048, 12288, 4096, ops.EPI_BF16) == (W4_224, 0)
    # the 224 px shapes (S = 768 / 2560, 2056 tower rows): decoder residual GEMMs at 2560 rows on 192 tiles of 224 rows instead of two rounds
    # of the ring kernel (207 vs 233 us at K = 11008); ONE round of 128x128 tiles where the grid fits it (video tower qkv, projector);
    # 768 x 22016 (258 tiles of 256 rows: two rounds for 1.008 rounds of work) as a COLUMN split -- 85 column tiles on whole rounds of big
    # tiles, the 256-column tail planned again; the single-round ring choices of rounds 2-4 unchanged
    assert ops.gemm_plan(2560, 4096, 11008, ops.EPI_F32_RESID) == (W4_224, 0) and ops.gemm_plan(2560, 4096, 4096, ops.EPI_F32_RESID) == (W4_224, 0)
    assert ops.gemm_plan(768, 12288, 4096, ops.EPI_BF16) == (W4_224, 0)
    assert ops.gemm_plan(2056, 3072, 1024, ops.EPI_BF16) == (_lib.CFG_128x128, 0) and ops.gemm_plan(2048, 4096, 4096, ops.EPI_BF16) == (_lib.CFG_128x128, 0)
    assert ops.gemm_plan_cols(768, 22016, 4096, ops.EPI_SWIGLU_BF16) == (W4, 0, 85 * 256)
    assert ops.gemm_plan_cols(256, 22016 - 85 * 256, 4096, ops.EPI_SWIGLU_BF16)[2] == 0 and ops.gemm_plan_cols(5120, 22016, 4096, ops.EPI_SWIGLU_BF16) == (W4, 0, 0)
    assert ops.gemm_plan(768, 4096, 4096, ops.EPI_F32_RESID) == (W4R, 0) and ops.gemm_plan(768, 4096, 11008, ops.EPI_F32_RESID) == (W4R, 0)
    assert ops.gemm_plan(1088, 4096, 4096 + 128, ops.EPI_F32_RESID)[0] in small         # the ring walks K in steps of 256
    # K not a multiple of 128 (no big-tile kernel) and the weight-streaming range
    assert ops.gemm_plan(4096, 4096, 4096 + 64, ops.EPI_BF16)[0] in small
    assert ops.gemm_plan(16, 4096, 4096, ops.EPI_BF16) == (_lib.CFG_SKINNY, 0)
    # row split: whole rounds on big tiles first, the remaining rows planned again
    cfg, first = ops.gemm_plan(8192 + 512, 4096, 4096, ops.EPI_F32_RESID)    # 2 rounds + a 512-row remainder < 2 rounds of 320-row tiles
    assert cfg == W4 and first == 8192 and ops.gemm_plan(512, 4096, 4096, ops.EPI_F32_RESID) == (W4R, 0)
    with pytest.raises(RuntimeError):
        ops.gemm_plan(0, 4096, 4096)


def test_tower_and_projector_factories_pick_the_reference_branches(tmp_path):
    """reference multimodal_encoder/builder.py:7-24 and multimodal_projector/builder.py:33-51, host side only: `openai*` / `laion*`
    names -> CLIPVisionTower (config read from the local checkpoint directory, nothing loaded under delay_load), `...LanguageBind_Image`
    / `...LanguageBind_Video_merge` -> the LanguageBind towers, anything else raises; 'linear' / 'mlpNx_gelu' / 'identity'."""
    import json
    from types import SimpleNamespace

    import pytest
    from vitron_amd.model.multimodal_encoder.builder import build_image_tower, build_video_tower
    from vitron_amd.model.multimodal_encoder.clip_encoder import CLIPVisionTower
    from vitron_amd.model.multimodal_encoder.languagebind import LanguageBindImageTower, LanguageBindVideoTower
    from vitron_amd.model.multimodal_projector.builder import IdentityMap, VisionProjector, build_vision_projector
    ck = tmp_path / "openai" / "clip-vit-large-patch14-336"
    ck.mkdir(parents=True)
    (ck / "config.json").write_text(json.dumps({"vision_config": {"hidden_size": 1024, "image_size": 336, "patch_size": 14, "num_hidden_layers": 24,
                                                                  "num_attention_heads": 16, "intermediate_size": 4096}}))
    t = build_image_tower(SimpleNamespace(mm_image_tower="openai/clip-vit-large-patch14-336", mm_vision_select_layer=-2,
                                          mm_vision_select_feature="cls_patch"), delay_load=True, cache_dir=str(tmp_path))
    assert isinstance(t, CLIPVisionTower) and not t.is_loaded
    assert t.config.hidden_size == 1024 and t.num_patches == 576 and t.hidden_size == 1024 and t.config.hidden_act == "quick_gelu"
    assert isinstance(build_image_tower(SimpleNamespace(image_tower="x/LanguageBind_Image"), delay_load=True), LanguageBindImageTower)
    assert isinstance(build_video_tower(SimpleNamespace(mm_video_tower="x/LanguageBind_Video_merge"), delay_load=True), LanguageBindVideoTower)
    for bad in ("x/other", None):
        with pytest.raises(ValueError):
            build_image_tower(SimpleNamespace(mm_image_tower=bad))
        with pytest.raises(ValueError):
            build_video_tower(SimpleNamespace(mm_video_tower=bad))
    with pytest.raises(ValueError):      # LanguageBind towers only know 'patch' (languagebind/__init__.py:96-104)
        build_image_tower(SimpleNamespace(mm_image_tower="x/LanguageBind_Image", mm_vision_select_feature="cls_patch"), delay_load=True)
    # the image tower's add_time_attn branch (modeling_image.py:74-84: temporal attention + temporal_mlp) is accepted since round 5
    # (packed by PackedVit, run by vt_vit_forward: tests/test_gpu_model.py::test_vit_image_tower_with_time_attention_and_temporal_mlp)
    it = build_image_tower(SimpleNamespace(image_tower="x/LanguageBind_Image"), delay_load=True)
    it.load_state({"hidden_size": 128, "num_attention_heads": 2, "patch_size": 14, "intermediate_size": 256, "image_size": 56,
                   "num_hidden_layers": 2, "add_time_attn": True},
                  {"encoder.layers.0.temporal_mlp.fc1.weight": torch.zeros(256, 128)})
    assert it.is_loaded and it.config.add_time_attn
    c = SimpleNamespace(mm_hidden_size=1024, hidden_size=4096)
    for name, depth in (("linear", 1), ("mlp2x_gelu", 2), ("mlp3x_gelu", 3), ("mlp5x_gelu", 5)):
        c.mm_projector_type = name
        p = build_vision_projector(c)
        assert isinstance(p, VisionProjector) and p.depth == depth
    c.mm_projector_type = "identity"
    assert isinstance(build_vision_projector(c), IdentityMap)
    c.mm_projector_type = "mlp_gelu"
    with pytest.raises(ValueError):
        build_vision_projector(c)


def test_serving_admission_host_logic_with_a_stub_model(monkeypatch):
    """vitron_amd.serving.ServingEngine._admit / step, host side only (stub model + stub page pool, the decoder pass patched out): an idle
    engine sizes the pool for the WHOLE candidate batch (never beyond the constructor's cap); a request that does not fit while others are
    live waits at the head of the queue with its spliced rows kept (embedded once); a request that fails on its own (encode, prefill, can
    never fit the pool) leaves the engine with its exception and the ones behind it are served; a device-level failure gives every page
    back and leaves every candidate queued."""
    import types

    import torch

    from vitron_amd import serving

    class Pool:
        def __init__(self, n):
            self.num_pages, self.free = n, list(range(n))

        def alloc(self, k):
            assert k <= len(self.free)
            out, self.free = self.free[:k], self.free[k:]
            return out

        def release(self, pages):
            self.free += list(pages)

    class Model:
        device = torch.device("cpu")
        config = types.SimpleNamespace(eos_token_id=2, vis_cache_entries=4)

        def __init__(self):
            self.kv, self.embeds, self.grow_fails = None, 0, False

        def get_model(self):
            return types.SimpleNamespace(llama=types.SimpleNamespace(embed=torch.zeros((4, 8))),
                                         embed_tokens=lambda ids: torch.zeros((1, ids.shape[1], 8)))

        bad_embed_len = None

        def prepare_inputs_labels_for_multimodal(self, ids, *a, **k):
            self.embeds += 1
            if ids.shape[1] == self.bad_embed_len:
                raise ValueError("images were passed but the model has no image tower")
            return (None, None, None, None, None, None)

        def _ensure_kv(self, n):
            if self.grow_fails:
                raise RuntimeError("pages in use")
            if self.kv is None or self.kv.num_pages < n:
                self.kv = Pool(n)

        def reset_prefix_cache(self):
            pass

    calls = {"n": 0, "fail": False, "fail_rows": None}

    def fake_forward(llama, kv, seqs, flat, lens, *a, **k):
        calls["n"] += 1
        if calls["fail"]:
            raise RuntimeError("vt_llama_forward failed (status -2): hipErrorLaunchFailure")
        if calls["fail_rows"] is not None and calls["fail_rows"] in lens:
            raise RuntimeError("llama_forward: position 9000 beyond rope table (8192)")
        for s, n in zip(seqs, lens):
            s.length += n
        return torch.zeros((len(seqs), 16))

    monkeypatch.setattr(serving, "llama_forward", fake_forward)
    monkeypatch.setattr(serving.ops, "argmax", lambda lg: torch.zeros((lg.shape[0],), dtype=torch.int32))
    monkeypatch.setattr(serving.ops, "embed_splice", lambda emb, vis, reg, plan: torch.zeros((plan.shape[0], 8)))
    m = Model()
    eng = serving.ServingEngine(m, max_batch=4)
    ids = lambda n: torch.ones((1, n), dtype=torch.long)             # noqa: E731
    need = lambda n, new: (n + new + 63) // 64 + 1                    # noqa: E731
    a, b = eng.submit(ids(100), max_new_tokens=28), eng.submit(ids(300), max_new_tokens=84)
    rest = eng._admit([eng.waiting.popleft(), eng.waiting.popleft()])
    assert rest == [] and len(eng.active) == 2 and m.kv.num_pages == need(100, 28) + need(300, 84) and not m.kv.free   # sized for the batch
    assert m.embeds == 2
    # a third request while the two are live: no page is free -> it waits (rows kept), nothing raised, nothing re-embedded on the retry
    c = eng.submit(ids(50), max_new_tokens=14)
    r = eng.waiting.popleft()
    assert eng._admit([r]) == [r] and r.flat is not None and r.need == need(50, 14) and m.embeds == 3
    assert eng._admit([r]) == [r] and m.embeds == 3
    # the first request retires: its pages come back, the waiting one fits now
    eng._retire(eng.active[0])
    eng.active = eng.active[1:]
    assert eng._admit([r]) == [] and r.flat is None and len(r.seq.pages) == need(50, 14) and r in eng.active
    # idle engine, the pool would have to grow but cannot (allocator says no) and every page of it is free: the request can never be
    # scheduled -> it FAILS on its own (it used to be re-queued and re-raised on every step, wedging the engine), no page leaked
    for q in list(eng.active):
        eng._retire(q)
    eng.active = []
    free_before = len(m.kv.free)
    m.grow_fails = True
    big = eng.submit(ids(4000), max_new_tokens=96)
    small = eng.submit(ids(40), max_new_tokens=8)
    cand = [eng.waiting.popleft(), eng.waiting.popleft()]
    assert eng._admit(cand) == [] and big in eng.failed and "KV pages" in str(eng.failed[big].error)
    assert [r.rid for r in eng.active] == [small] and len(m.kv.free) == free_before - need(40, 8)      # the one behind it is served
    eng._retire(eng.active[0])
    eng.active = []
    m.grow_fails = False
    # a prefill that fails for ONE request (its own fault: e.g. a prompt beyond the rotary table): that request fails, the other runs
    calls["fail_rows"] = 77
    bad, good = eng.submit(ids(77), max_new_tokens=8), eng.submit(ids(60), max_new_tokens=8)
    assert eng._admit([eng.waiting.popleft(), eng.waiting.popleft()]) == []
    assert bad in eng.failed and [r.rid for r in eng.active] == [good] and eng.failed[bad].seq.pages == []
    assert len(m.kv.free) == m.kv.num_pages - need(60, 8)
    eng._retire(eng.active[0])
    eng.active = []
    calls["fail_rows"] = None
    # a DEVICE error in the prefill (HIP status -2 of the C ABI) is nobody's fault: it propagates, pages back, candidates queued again
    calls["fail"] = True
    dev_a, dev_b = eng.submit(ids(50), max_new_tokens=8), eng.submit(ids(51), max_new_tokens=8)
    cand = [eng.waiting.popleft(), eng.waiting.popleft()]
    with pytest.raises(RuntimeError, match="status -2"):
        eng._admit(cand)
    assert list(eng.waiting) == cand and all(r.seq.pages == [] for r in cand) and len(m.kv.free) == m.kv.num_pages and not eng.active
    calls["fail"] = False
    assert eng._admit([eng.waiting.popleft(), eng.waiting.popleft()]) == [] and len(eng.active) == 2
    # an encode that fails for one request (bad image shape): isolated the same way, step() keeps serving
    for q in list(eng.active):
        eng._retire(q)
    eng.active = []
    m.bad_embed_len = 33
    e_bad, e_ok = eng.submit(ids(33), max_new_tokens=4), eng.submit(ids(20), max_new_tokens=4)
    out = eng.step()
    assert e_bad in eng.failed and isinstance(eng.errors()[e_bad], ValueError) and [rid for rid, _ in out] == [e_ok]
    assert eng.cancel(e_ok) is False and eng.pending() == 1
    outs = eng.run()
    assert e_ok in outs and e_bad not in outs and eng.pending() == 0 and len(m.kv.free) == m.kv.num_pages
    # a pool CAPPED by the constructor is never rebuilt larger: the prefix that fits is admitted, the rest waits
    monkeypatch.setattr(serving, "PagedKVCache", lambda llama, n: Pool(n))
    m2 = Model()
    eng2 = serving.ServingEngine(m2, max_batch=4, kv_pages=10)
    r1, r2, r3 = eng2.submit(ids(200), max_new_tokens=56), eng2.submit(ids(200), max_new_tokens=56), eng2.submit(ids(100), max_new_tokens=28)
    rest = eng2._admit([eng2.waiting.popleft() for _ in range(3)])
    assert m2.kv.num_pages == 10 and [r.rid for r in eng2.active] == [r1, r2] and [r.rid for r in rest] == [r3]       # 5 + 5 pages fit, 3 more do not
    huge = eng2.submit(ids(1000), max_new_tokens=8)                                                                    # 17 pages > the cap
    for q in list(eng2.active):
        eng2._retire(q)
    eng2.active = []
    assert eng2._admit(rest + [eng2.waiting.popleft()]) == [] and huge in eng2.failed and [r.rid for r in eng2.active] == [r3]
    assert eng2.cancel(12345) is False
    w = eng2.submit(ids(10), max_new_tokens=2)
    assert eng2.cancel(w) is True and not eng2.waiting
    assert (a, b, c, big) == (0, 1, 2, 3)


def test_hand_placed_attention_kernel_compiles_without_scratch():
    """flash_attn_w4_kernel counts its LDS-DMA pieces with s_waitcnt vmcnt(N): a register spill would put scratch accesses -- which
    count in vmcnt too -- between them and turn the counted wait into a race (seen once on the GPU, in the reference form of the
    kernel). Both operand builds of every variant must therefore compile with ScratchSize 0 and stay inside the register file of one
    wave per SIMD; hipcc cross-compiles for gfx950 without a GPU (3 s per build)."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "vitron_amd", "csrc", "vt_attn_w4.hip")
    for extra in ([], ["-DVT_OPERAND_F16=1"]):
        with tempfile.TemporaryDirectory() as tmp:
            r = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage",
                                *extra, "-c", src, "-o", os.path.join(tmp, "w4.o")], capture_output=True, text=True, cwd=tmp)
        assert r.returncode == 0, r.stderr[-2000:]
        names = re.findall(r"Function Name: (\S+)", r.stderr)
        scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
        vgprs = [int(x) for x in re.findall(r" VGPRs: (\d+)", r.stderr)]
        agprs = [int(x) for x in re.findall(r"AGPRs: (\d+)", r.stderr)]
        assert len(names) == 7 and all("flash_attn_w4_kernel" in n for n in names), names      # causal / not x placed / reference / persistent, + level 3's operand out
        assert scratch == [0] * 7, dict(zip(names, scratch))
        assert all(v <= 256 for v in vgprs) and all(a <= 256 for a in agprs), (vgprs, agprs)


def test_counted_dma_waits_have_no_scratch_access_in_their_window_and_mfma_tails_are_fenced():
    """Every kernel that waits for its LDS-DMA pieces by count (s_waitcnt vmcnt(N)) must have no scratch access between its first DMA
    and its last MFMA: a spill counts in vmcnt and may retire out of order with the loads, so the counted wait could let a piece through
    early (round 6 found one in the prologue of gemm_w4x_kernel<16-bit store> of the bf16 build). tools/check_scratch_window.py over the
    device assembly of every source file that holds LDS-DMA kernels (four-wave / ping-pong / ring GEMMs, the weight-streaming GEMM, both
    attention files), both operand builds (cross-compiled, no GPU)."""
    import importlib.util
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    spec = importlib.util.spec_from_file_location("check_scratch_window", os.path.join(ROOT, "tools", "check_scratch_window.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    with tempfile.TemporaryDirectory() as tmp:
        jobs = []
        for name, flags in (("vt_gemm8", []), ("vt_attn_w4", ["-fno-slp-vectorize"]), ("vt_gemm", []), ("vt_attn", [])):
            for tag, extra in (("bf16", []), ("f16", ["-DVT_OPERAND_F16=1"])):
                out = os.path.join(tmp, f"{name}_{tag}.s")
                cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", *flags, *extra, "-S", "--cuda-device-only",
                       os.path.join(ROOT, "vitron_amd", "csrc", name + ".hip"), "-o", out]
                jobs.append((out, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp)))
        for out, proc in jobs:
            _, err = proc.communicate()
            assert proc.returncode == 0, err[-2000:]
            n, bad = chk.check(out)
            assert n >= 7 and not bad, (os.path.basename(out), n, [b[0] for b in bad])
            # ... and no read of the last MFMAs' destinations between the last (asm) MFMA and the wait states that cover its latency: the copies
            # the compiler places on a loop's exit edge once took the very last destination four instructions behind its MFMA (round 6)
            n2, bad2 = chk.check_mfma_tail(out)
            assert not bad2, (os.path.basename(out), n2, bad2[:4])
            if "vt_gemm8" in out or "vt_attn_w4" in out:
                assert n2 >= 7, (os.path.basename(out), n2)
            # ... and the zeroing of an accumulator stays away from the first asm MFMA that reads it as C (the compiler sinks the writes behind
            # the source's s_nop: what keeps them apart is their order, which this pins)
            n3, bad3 = chk.check_mfma_head(out)
            assert not bad3, (os.path.basename(out), n3, bad3[:4])


def test_mfma_tail_checker_on_hand_written_assembly(tmp_path):
    """tools/check_scratch_window.py: check_mfma_tail must flag a read of the last asm MFMA's destination that comes before the wait states --
    also when the path to it follows an s_branch around an unrelated block -- and accept the same code with the wait states first (the form
    the loop's last trip now produces); check() must flag a scratch access between the first LDS-DMA and the last MFMA."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_scratch_window", os.path.join(ROOT, "tools", "check_scratch_window.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    head = """_Z4kernv:
\tbuffer_load_dwordx4 v[0:3], v4, s[0:3], 0 offen lds
.LBB0_1:
\t;;#ASMSTART
\tv_mfma_f32_16x16x32_bf16 a[0:3], v[6:9], v[10:13], a[0:3]
\t;;#ASMEND
\t;;#ASMSTART
\tv_mfma_f32_16x16x32_bf16 a[252:255], v[6:9], v[10:13], a[252:255]
\t;;#ASMEND
\ts_cmp_lt_i32 s1, s2
\ts_cbranch_scc1 .LBB0_1
"""
    tail = "\tv_accvgpr_read_b32 v20, a0\n\ts_endpgm\n.Lfunc_end0:\n"
    bad = head + "\ts_branch .LBB0_3\n.LBB0_2:\n\tv_accvgpr_write_b32 a252, 0\n.LBB0_3:\n\tv_accvgpr_read_b32 v8, a252\n\t;;#ASMSTART\n\ts_nop 15\n\ts_nop 7\n\t;;#ASMEND\n" + tail
    good = head + "\t;;#ASMSTART\n\ts_nop 15\n\ts_nop 7\n\t;;#ASMEND\n\ts_branch .LBB0_3\n.LBB0_2:\n\tv_accvgpr_write_b32 a252, 0\n.LBB0_3:\n\tv_accvgpr_read_b32 v8, a252\n" + tail
    spill = head.replace(".LBB0_1:\n", ".LBB0_1:\n\tscratch_store_dword off, v1, off offset:8\n") + "\t;;#ASMSTART\n\ts_nop 15\n\t;;#ASMEND\n" + tail
    for name, text, n_tail, n_scratch in (("bad", bad, 1, 0), ("good", good, 0, 0), ("spill", spill, 0, 1)):
        f = tmp_path / f"{name}.s"
        f.write_text(text)
        n2, bad2 = chk.check_mfma_tail(str(f))
        n1, bad1 = chk.check(str(f))
        assert n1 == 1 and n2 == 1, (name, n1, n2)
        assert len(bad2) == n_tail and len(bad1) == n_scratch, (name, bad1, bad2)
    assert "a252" in bad_line(chk, tmp_path / "bad.s")


def bad_line(chk, path):
    return chk.check_mfma_tail(str(path))[1][0][2]


def test_hand_placed_attention_schedule_is_what_the_generator_emits():
    """vt_attn_w4_si0.inc / si1.inc are generated (tools/gen_attn_w4.py): the committed files must be the generator's output, every
    sub-iteration must carry its 32 MFMAs (16 score + 16 P.V), 8 + 8 fragment reads, and SI0 the 8 LDS-DMA pieces of the tile."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_attn_w4", os.path.join(ROOT, "tools", "gen_attn_w4.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    for sub in (0, 1):
        text = gen.generate(sub)
        with open(os.path.join(ROOT, "vitron_amd", "csrc", f"vt_attn_w4_si{sub}.inc")) as f:
            assert f.read() == text, f"vt_attn_w4_si{sub}.inc is stale: run python tools/gen_attn_w4.py"
        assert text.count("W4A_QK(") + text.count("W4A_QK0(") == 16 and text.count("W4A_PV(") == 16
        assert text.count("kfr[") - text.count("kfr[", 0, 0) >= 8 and len(re.findall(r"kfr\[\d\] = ", text)) == 8 and len(re.findall(r"vfr\[\d\] = ", text)) == 8
        assert text.count("W4A_DMA_K(") == (4 if sub == 0 else 0) and text.count("W4A_DMA_V(") == (4 if sub == 0 else 0)
        assert text.count("v_exp_f32") == 32 and text.count("v_cvt_pk_f16_f32") == 16 and text.count("v_fma_f32") == 32
        # a transcendental result is never consumed by the instruction right behind it inside a block (the hardware asks for one wait state)
        for blk in re.findall(r'asm volatile\("([^"]*)"', text):
            ins = blk.split("\\n\\t")
            for a, b in zip(ins, ins[1:]):
                if a.startswith("v_exp_f32"):
                    assert a.split()[1].rstrip(",") not in [x.rstrip(",") for x in b.split()[2:]], (a, b)


def test_causal_attention_dispatch_order_is_a_better_schedule():
    """vt_flash_attn_block_order (host logic of the one-wave-per-SIMD attention launcher): for one 5120-token sequence x 32 heads on 256
    CUs the planned order must be a permutation of the grid, keep heads = XCD mod 8 (entry i runs on XCD i % 8), and -- replayed through
    the dispatcher's rule "next workgroup to the first CU that frees up" with the same cost model -- end well below the natural
    heaviest-first order (LPT: 11.8 % above the mean load; plan: ~3 %). Shapes where the natural order is already balanced keep it."""
    import ctypes
    import heapq
    from vitron_amd import _lib
    lib = _lib.load(build_if_needed=True)

    def order_of(heads, nqb, nseq, ncu=256):
        buf = (ctypes.c_int * (heads * nqb * nseq))()
        n = lib.vt_flash_attn_block_order(heads, nqb, nseq, ncu, buf, len(buf))
        assert n in (0, len(buf)), _lib.last_error(lib)
        return list(buf[:n])

    def makespan(order, heads, nqb, ncu=256):
        # per XCD: workgroup i goes to XCD i % 8, inside it to the first free CU (32 per XCD)
        ends = []
        for x in range(8):
            cus = [0.0] * (ncu // 8)
            heapq.heapify(cus)
            for i in range(x, len(order), 8):
                y = (order[i] // heads) % nqb
                heapq.heappush(cus, heapq.heappop(cus) + 21.8e3 + 3183.0 * 4 * (nqb - y))
            ends.append(max(cus))
        return max(ends)

    heads, nqb = 32, 20
    order = order_of(heads, nqb, 1)
    assert sorted(order) == list(range(heads * nqb))                                   # every block exactly once
    assert all((b % heads) % 8 == i % 8 for i, b in enumerate(order))                  # a head's pages stay behind one L2
    natural = [h + heads * y for y in range(nqb) for h in range(heads)]
    mean = sum(21.8e3 + 3183.0 * 4 * (nqb - y) for y in range(nqb)) * heads / 256
    assert makespan(natural, heads, nqb) / mean > 1.10
    assert makespan(order, heads, nqb) / mean < 1.04
    assert order_of(32, 20, 8) == []            # eight clips: 20 blocks per CU, list scheduling balances by itself
    assert order_of(32, 4, 1) == []             # one round
    assert order_of(32, 16, 1) == []            # 512 blocks = exactly two per CU in complementary pairs: nothing to gain
    o = order_of(24, 20, 1, 240)                # heads not a multiple of 8 on a 240-CU part: planned globally, still a permutation
    assert o == [] or sorted(o) == list(range(24 * 20))
