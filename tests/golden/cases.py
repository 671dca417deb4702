"""Shared definitions of the golden cases: tiny configs, seeds and seeded inputs.

Imported by tests/golden/make_golden.py (which runs the REFERENCE on them) and by the tests (which run the
oracle and the HIP path on the very same inputs). Head dims are the real ones (64 for the ViT, 128 for the LLM)
because the kernels are specialised for them; widths, depths and sequence lengths are shrunk.
"""
from __future__ import annotations

import torch

SEED_VIT, SEED_PIX, SEED_REGION, SEED_FEATS, SEED_PROJ, SEED_LLM, SEED_IDS = 1234, 4321, 77, 78, 79, 80, 81

VIT_VIDEO = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, patch_size=14,
                 image_size=56, hidden_act="gelu", layer_norm_eps=1e-5, add_time_attn=True, num_frames=4)
VIT_IMAGE = dict(VIT_VIDEO, add_time_attn=False, num_frames=1)
# second pinned tower shape: the class-default activation, the 8-frame temporal path (the T = 8 kernel), an odd patch grid (5 x 5)
VIT_VIDEO_B = dict(hidden_size=128, intermediate_size=192, num_hidden_layers=2, num_attention_heads=2, patch_size=14,
                   image_size=70, hidden_act="quick_gelu", layer_norm_eps=1e-5, add_time_attn=True, num_frames=8)
VIT_IMAGE_B = dict(VIT_VIDEO_B, add_time_attn=False, num_frames=1)
VIT_B_CASES = {"video_b": (VIT_VIDEO_B, (1, 3, 8, 70, 70)), "image_b": (VIT_IMAGE_B, (2, 3, 70, 70))}
# third pinned tower shape (round 5): the IMAGE tower file's add_time_attn variant -- temporal attention AND a temporal MLP per layer
# (reference image/modeling_image.py:74-84,105-134) -- on 4-frame clips and in its degenerate num_frames = 1 form (no embedding add)
VIT_IMAGE_TMLP = dict(VIT_VIDEO, temporal_mlp=True)
VIT_IMAGE_TMLP1 = dict(VIT_VIDEO, temporal_mlp=True, num_frames=1)
VIT_TMLP_CASES = {"tmlp_t4": (VIT_IMAGE_TMLP, (2, 3, 4, 56, 56)), "tmlp_t1": (VIT_IMAGE_TMLP1, (3, 3, 56, 56))}
MM_HIDDEN = 128
# F1 branches: the plain HF-CLIP image tower (reference clip_encoder.py) -- CLIPVisionConfig defaults
# (quick_gelu), a 4 x 4 patch grid -- and an mlp3x_gelu projector (multimodal_projector/builder.py:39-46)
CLIP_TOWER = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, patch_size=14,
                  image_size=56, hidden_act="quick_gelu", layer_norm_eps=1e-5, add_time_attn=False, num_frames=1)
CLIP_TOWER_SHAPE = (3, 3, 56, 56)
PROJ3_ROWS = 41
LLM = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, vocab_size=512,
           rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=512)

# init scales: larger than the 0.02 of the benchmark so attention / biases / norm affine actually matter
VIT_INIT = dict(w_std=0.05, b_std=0.05, ln_jitter=0.1, attn_std=0.15)
MLP_INIT = dict(w_std=0.05, b_std=0.05)
LLM_INIT = dict(w_std=0.05, ln_jitter=0.1, attn_std=0.12)

# boxes in the 224-pixel space of RegionExtractor; the last one selects no cell (empty mask)
BOXES = [[0, 0, 224, 224], [0, 58.94736842105263, 117.89473684210526, 117.89473684210526], [100, 20, 180, 200],
         [7, 7, 8, 8], [0, 0, 6, 6]]
REGION_CASES = {"g16": (128, 256, 16), "g4": (128, 256, 4)}

PROMPTS = ["ab<image>cd", "<image><image>\nq", "a<image>\n<objs> b", "x<objs>y<objs>", "plain text", ""]
MODEL_OUTPUTS = [
    "Sure, here it is. <module>A</module><instruction>prompt: a red fox in the snow</instruction>",
    "I segmented it.<module>B</module> <instruction>track: the dog: left one </instruction><region>[10, 20, 110, 220]</region> done",
    "<module>C</module><instruction>a</instruction><instruction>edit: make it night</instruction><instruction>x:y: z</instruction>",
    "no tags at all", "", "<module></module><instruction></instruction><region></region>",
    "dangling <module>D and <instruction>never closed", "<SP>hidden</SP>visible<module>E</module> tail <b>bold</b>!",
    "line one <module>F\n</module> spans a newline <region>r1</region><region>r2</region>",
    "a < b and c > d <module>G</module>", "<instruction>only: colon:</instruction>", "x<module>H</module>y<module>I</module>z",
    "<instruction> spaced : arg with spaces  </instruction>\n<instruction>second line: two</instruction>",
]
REGION_RESCALE = [([0, 100, 300, 200], [570, 380], [224, 224]), ([12.5, 3, 99, 640], [640, 480], [336, 336])]


def bf16r(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def pixels(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return bf16r(torch.randn(shape, generator=g))


def features(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return bf16r(torch.randn(shape, generator=g))


def _ids(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(3, LLM["vocab_size"], (n,), generator=g).tolist()


def glue_cases():
    """input_ids (with -200/-300 sentinels), images list, regions list -> what the reference's
    prepare_inputs_labels_for_multimodal + forward are run on."""
    img = lambda s: pixels((3, 56, 56), SEED_PIX + s)  # noqa: E731
    vid = lambda s: pixels((3, VIT_VIDEO["num_frames"], 56, 56), SEED_PIX + 100 + s)  # noqa: E731
    T = VIT_VIDEO["num_frames"]
    c = {}
    # image + region prompt (app.py:525-534 layout: <image>\n<objs> ...); stray BOS after -300 (Appendix B quirk)
    ids = [1] + _ids(2, SEED_IDS) + [-200] + _ids(3, SEED_IDS + 1) + [-300, 1] + _ids(4, SEED_IDS + 2)
    c["image_region"] = dict(input_ids=torch.tensor([ids]), attention_mask=None, images=[img(0)], regions=[BOXES[1]])
    # one clip: T consecutive -200 (app.py:520), no regions
    ids = [1] + [-200] * T + _ids(5, SEED_IDS + 3)
    c["video"] = dict(input_ids=torch.tensor([ids]), attention_mask=None, images=[vid(0)], regions=None)
    # batch of two with different lengths -> right padding; sample 1 has an image but no <objs>
    a = [1] + _ids(1, SEED_IDS + 4) + [-200, -300] + _ids(6, SEED_IDS + 5)
    b = [1, -200] + _ids(2, SEED_IDS + 6)
    L = max(len(a), len(b))
    ids = torch.tensor([a + [0] * (L - len(a)), b + [0] * (L - len(b))])
    am = torch.tensor([[1] * len(a) + [0] * (L - len(a)), [1] * len(b) + [0] * (L - len(b))])
    c["batch_pad"] = dict(input_ids=ids, attention_mask=am, images=[img(1), img(2)], regions=[BOXES[2], BOXES[0]])
    # text-only turn that still passes a (zeros) image and the default region (app.py:492,554-557)
    ids = [1] + _ids(9, SEED_IDS + 7)
    c["text_only"] = dict(input_ids=torch.tensor([ids]), attention_mask=None, images=[torch.zeros(3, 56, 56)],
                          regions=[[0, 0, 224, 224]])
    # video + image in one sample (videos come first in `images`, app.py:559), truncated by tokenizer_model_max_length
    ids = [1] + [-200] * T + _ids(2, SEED_IDS + 8) + [-200, -300] + _ids(3, SEED_IDS + 9)
    c["video_image_trunc"] = dict(input_ids=torch.tensor([ids]), attention_mask=None, images=[vid(1), img(3)],
                                  regions=[BOXES[0], BOXES[3]], max_length=70)
    # same, left padding side, in a batch with a short text+image sample
    a = [1] + _ids(2, SEED_IDS + 10) + [-200] + _ids(2, SEED_IDS + 11)
    b = [1] + _ids(1, SEED_IDS + 12) + [-200, -300]
    L = max(len(a), len(b))
    ids = torch.tensor([a + [0] * (L - len(a)), b + [0] * (L - len(b))])
    am = torch.tensor([[1] * len(a) + [0] * (L - len(a)), [1] * len(b) + [0] * (L - len(b))])
    c["batch_left"] = dict(input_ids=ids, attention_mask=am, images=[img(4), img(5)], regions=[BOXES[1], BOXES[2]],
                           padding_side="left")
    return c


def random_glue_cases(n=24, seed=20260924):
    """Seeded random layouts for the glue (more of what glue_cases() does by hand): batches of 1-3 samples, each with 0-2 images
    and / or one clip, optional <objs> after an image, optional regions (one box per entry of `images`, as app.py passes them),
    ragged lengths with right-padded masks, optional truncation, either padding side. The reference's
    prepare_inputs_labels_for_multimodal is run on these by make_golden.gen_glue_random; stored: the masks and a fixed random
    projection of the spliced embeddings (every row of the layout is pinned, the fixture stays small)."""
    import random
    rnd = random.Random(seed)
    T = VIT_VIDEO["num_frames"]
    out = {}
    for i in range(n):
        B = rnd.randint(1, 3)
        use_regions = rnd.random() < 0.6
        rows, images, regions = [], [], []
        for b in range(B):
            ids = [1] + _ids(rnd.randint(0, 3), SEED_IDS + 100 * i + b)
            entries = []
            if rnd.random() < 0.35:
                entries.append("vid")                                  # videos come first in `images` (app.py:559)
            entries += ["img"] * rnd.choice([0, 1, 1, 2])
            if not entries and rnd.random() < 0.5:
                entries = ["img"]
            for k, e in enumerate(entries):
                if e == "vid":
                    ids += [-200] * T
                    images.append(pixels((3, T, 56, 56), SEED_PIX + 1000 + 10 * i + 3 * b + k))
                else:
                    ids += [-200]
                    images.append(pixels((3, 56, 56), SEED_PIX + 2000 + 10 * i + 3 * b + k))
                    if use_regions and rnd.random() < 0.7:
                        ids += _ids(rnd.randint(0, 2), SEED_IDS + 7 * i + k) + [-300]
                regions.append([rnd.uniform(0, 100), rnd.uniform(0, 100), rnd.uniform(100, 224), rnd.uniform(100, 224)])
                ids += _ids(rnd.randint(0, 4), SEED_IDS + 13 * i + 5 * b + k)
            if not entries:                                            # text-only sample still consumes one (zeros) image
                images.append(torch.zeros(3, 56, 56))
                regions.append([0, 0, 224, 224])
                ids += _ids(rnd.randint(1, 5), SEED_IDS + 17 * i + b)
            rows.append(ids)
        L = max(len(r) for r in rows)
        ragged = any(len(r) != L for r in rows)
        ids_t = torch.tensor([r + [0] * (L - len(r)) for r in rows])
        am = torch.tensor([[1] * len(r) + [0] * (L - len(r)) for r in rows]) if (ragged or rnd.random() < 0.3) else None
        case = dict(input_ids=ids_t, attention_mask=am, images=images, regions=regions if use_regions else None)
        if rnd.random() < 0.3:
            case["max_length"] = rnd.randint(8, 90)
        if rnd.random() < 0.4:
            case["padding_side"] = "left"
        out[f"rand{i:02d}"] = case
    return out


def glue_projection(hidden):
    """The fixed direction the random-case goldens are projected on (float64)."""
    g = torch.Generator().manual_seed(424242)
    return torch.randn((hidden,), generator=g, dtype=torch.float64)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE-width cases (SURVEY.md 8(d) widths, bench.py's init: N(0, 0.02^2) weights, zero biases, unit norm gains). The
# reference's own modules are run on them by make_golden.gen_fullwidth; stored compactly in fullwidth.npz: for every big
# tensor its projection on FW_NPROJ fixed random directions (float64, every row pinned) + FW_ROWS whole rows + top-5 ids.
# ---------------------------------------------------------------------------------------------------------------------
FW_SEED = 20260925
FW_NPROJ, FW_ROWS = 4, 12
FW_INIT = dict(w_std=0.02)                         # == bench.py / init_synthetic defaults
FW_LLAMA = {"s1088_l2": (1088, 2), "s2048_l1": (2048, 1)}      # name -> (sequence length, decoder layers) at H=4096 / I=11008 / 32 heads
FW_VIT_LAYERS = 2                                  # encoder layers of the 336 px towers that are pinned
FW_VIDEO_SHAPE = (1, 3, 8, 336, 336)               # one 8-frame clip: 4616 token rows (C3)
FW_IMAGE_SHAPE = (2, 3, 336, 336)                  # two images: 2 x 577 rows (C2 / C5)
FW_PROJ_ROWS = 1152
# boxes for the region extractor at G = 24 (336 px tower) on the reference's 224 canvas (scale 9.33) and on a 336 canvas (scale 14)
FW_BOXES_224 = [[0, 0, 224, 224], [10.5, 20.25, 120.75, 200.0], [100, 20, 180, 200], [7, 7, 8, 8], [0, 0, 6, 6], [215, 0, 224, 224],
                [37.3, 112.0, 37.9, 113.0], [112, 112, 224, 224]]
FW_BOXES_336 = [[0, 0, 336, 336], [10.5, 20.25, 320.75, 200.0], [100, 20, 180, 300], [7, 7, 8, 8], [0, 0, 6, 6], [335, 0, 336, 336],
                [55.9, 168.0, 56.1, 169.0], [168, 168, 336, 336]]


# ---- the REFERENCE-NATIVE 224 px shapes (processing_image.py:20-21, processing_video.py:50-51, layer.py:60) ----
# N = 257 tokens per frame, G = 16; C2-224: S = 256 + 512 = 768, C3-224: S = 8 * 256 + 512 = 2560 -- the only shapes a real Vitron
# checkpoint runs (round 5; tests/golden/fullwidth_224.npz)
FW224_LLAMA = {"s768_l2": (768, 2), "s2560_l1": (2560, 1)}
FW224_VIDEO_SHAPE = (1, 3, 8, 224, 224)            # one 8-frame clip: 2056 token rows (C3-224)
FW224_IMAGE_SHAPE = (2, 3, 224, 224)               # two images: 2 x 257 rows (C2-224 / C5-224)
FW224_PROJ_ROWS = 512
# boxes for RegionExtractor(1024, 4096) on its default 224 canvas and the 16 x 16 grid of the 224 px tower (scale 14: the case the
# reference itself runs, app.py:533 / inference_image.py:34 rescale every box to [224, 224])
FW224_BOXES = [[0, 0, 224, 224], [0, 58.94736842105263, 117.89473684210526, 117.89473684210526], [100, 20, 180, 200], [7, 7, 8, 8],
               [0, 0, 6, 6], [223, 0, 224, 224], [37.3, 112.0, 37.9, 113.0], [112, 112, 224, 224], [10.5, 20.25, 120.75, 200.0]]


def fw_directions(dim, n=FW_NPROJ, seed=FW_SEED):
    g = torch.Generator().manual_seed(seed + dim)
    return torch.randn((dim, n), generator=g, dtype=torch.float64)


def fw_rows(total, n=FW_ROWS):
    """Evenly spread row indices incl. the first and the last row."""
    return sorted({int(round(i * (total - 1) / (n - 1))) for i in range(n)})


def fw_llama_embeds(S, seed):
    """Decoder input rows of the size spliced embeddings have under the bench init (token embeddings ~ N(0, 0.02^2))."""
    g = torch.Generator().manual_seed(seed)
    return bf16r(torch.randn((S, 4096), generator=g) * 0.02)


# user turns of the drop-in acceptance test, composed exactly as the reference composes them:
#   image:        inference_image.py:38      DEFAULT_IMAGE_TOKEN + '\n' + inp
#   image_region: app.py:525-534             ' ' + <image> + '\n' + <objs> + ' ' + user_input
#   video:        inference_image.py:88      ' '.join([<image>] * num_frames) + '\n' + inp      (num_frames = 4 on the tiny tower)
ACCEPTANCE_USER_TURNS = {
    "image": "<image>\nCould you help me transform the image into a video?",
    "image_region": " <image>\n<objs> What is in this region?",
    "video": " ".join(["<image>"] * 4) + "\nWhy is this video funny?",
}


# ---- pre-processing pins (round 5): synthetic decoded pictures / clips for the reference's own processors (preproc_ref.npz) -------------
# name -> (frames, H, W): portrait, landscape, already 224 x 224 (Resize returns its input), an up-scale, a long side that lands on
# int(224 * 500 / 333) = 336; clips: landscape with more frames than are sampled, portrait with exactly 8, FEWER frames than are sampled
# (np.linspace(..., dtype=int) repeats indices) on a square up-scale
PREPROC_IMAGES = {"portrait": (1, 300, 200), "landscape": (1, 240, 320), "identity": (1, 224, 224), "upscale": (1, 97, 131),
                  "long336": (1, 333, 500)}
PREPROC_CLIPS = {"clip_landscape": (20, 120, 200), "clip_portrait": (8, 150, 100), "clip_short": (5, 64, 64)}
PREPROC_SEED = 9100
PREPROC_IMG_STRIDE, PREPROC_CLIP_STRIDE = 3, 6      # the fixture keeps every n-th pixel of an output (+ float64 sums over all of them)


def preproc_frames(shape, seed):
    """Smooth + noisy uint8 frames [F,H,W,3] (structure at several scales, so that resampling errors show)."""
    g = torch.Generator().manual_seed(seed)
    F_, H, W = shape
    yy, xx = torch.meshgrid(torch.linspace(0, 3.0, H), torch.linspace(0, 5.0, W), indexing="ij")
    base = 127 + 90 * torch.sin(yy[None] * 2.1 + torch.arange(F_)[:, None, None]) * torch.cos(xx[None] * 1.3)
    img = base[..., None] + torch.randn((F_, H, W, 3), generator=g) * 25 + torch.tensor([0.0, 12.0, -9.0])
    return img.clamp(0, 255).to(torch.uint8)


def preproc_inputs():
    """name -> uint8 [F,H,W,3] for every pre-processing case, in a fixed order."""
    out = {}
    for i, (name, shp) in enumerate({**PREPROC_IMAGES, **PREPROC_CLIPS}.items()):
        out[name] = preproc_frames(shp, PREPROC_SEED + i)
    return out
