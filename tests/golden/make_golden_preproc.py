"""Fixture for the pre-processing restatement on REAL images (build container only: needs /root/reference/examples/*.jpg).

    python tests/golden/make_golden_preproc.py

What it pins and what it cannot: the reference's processors (languagebind/image/processing_image.py:15-31, video/processing_video.py:
26-53) are compositions of torchvision / pytorchvideo transforms, neither of which is installed offline, so the reference's own code
cannot produce this fixture. The oracle restates them (oracle/vitron_oracle.py: preprocess_image / preprocess_video): ToTensor, then
torch.nn.functional.interpolate on the TENSOR (bicubic, align_corners=False, no antialias -- what torchvision 0.15.2's Resize does
for tensors) to the size torchvision computes (short side -> S, long side int(S * long / short)), centre crop at
int(round((h - S) / 2)), Normalize. This script runs that restatement on three of the reference's COCO examples (decoded with PIL)
and stores, per image and output size, the shape arithmetic (resized size, crop offsets), per-channel mean / std, 64 probe
pixels and a checksum -- so that the interpolation call, the size / crop arithmetic and the normalisation are frozen against
drift (torch upgrades, edits of the oracle). The size / crop rules themselves remain a restatement of the two libraries' documented
behaviour ("parity unpinned" for that part, as DESIGN.md 4 says). Inputs are not copied into the repo: the test that reads this
fixture runs where /root/reference exists and is skipped elsewhere.

Round 5 adds a second fixture, preproc_ref.npz (gen_ref below): the reference's OWN processor code run over restated third-party
primitives on synthetic inputs -- self-contained, so the tests that read it run everywhere (`--ref-only` writes only that one)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import vitron_oracle as O  # noqa: E402

EXAMPLES = ["000000015269.jpg", "000000036260.jpg", "000000116439.jpg"]
REF_EXAMPLES = "/root/reference/examples"


def probes(n=64, seed=7):
    g = torch.Generator().manual_seed(seed)
    return torch.rand((n, 2), generator=g)


def describe(path, size):
    from PIL import Image
    im = Image.open(path).convert("RGB")
    x = torch.from_numpy(np.array(im))
    h, w = x.shape[:2]
    y = O.preprocess_image(x, size)
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nh, nw = size, int(size * w / h)
    p = probes()
    iy = (p[:, 0] * (size - 1)).round().long()
    ix = (p[:, 1] * (size - 1)).round().long()
    # the video path on the same picture repeated as a 2-frame clip (ShortSideScale + CenterCropVideo)
    v = O.preprocess_video(torch.stack([x, x]), size)
    return {"file": os.path.basename(path), "size": size, "input_hw": [h, w], "resized_hw": [nh, nw],
            "crop_top_left": [int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))],
            "mean": y.mean((1, 2)).tolist(), "std": y.std((1, 2)).tolist(),
            "probe_values": y[:, iy, ix].t().reshape(-1).tolist(), "checksum": float(y.double().abs().sum()),
            "video_checksum": float(v.double().abs().sum()), "video_probe_values": v[:, 1, iy, ix].t().reshape(-1).tolist()}


def main():
    out = [describe(os.path.join(REF_EXAMPLES, f), s) for f in EXAMPLES for s in (224, 336)]
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "preproc_real.json"), "w") as f:
        json.dump({"torch": torch.__version__, "cases": out}, f, indent=1)
    print(json.dumps([{k: v for k, v in c.items() if "probe" not in k} for c in out], indent=1))


def summarize(y, stride):
    """What the fixture keeps of one output: every `stride`-th pixel (fp32) and float64 [sum, sum |.|, sum of squares] per channel."""
    y = y.float()
    yd = y.double().reshape(3, -1)
    return y[..., ::stride, ::stride].contiguous().numpy(), torch.stack([yd.sum(1), yd.abs().sum(1), (yd * yd).sum(1)], 1).numpy()


def gen_ref():
    """tests/golden/preproc_ref.npz: the REFERENCE's own processors (processing_image.py:15-66, processing_video.py:26-153), imported
    unmodified and run over restated third-party primitives (oracle/ref_preproc.py says which, at which pinned version), on the
    synthetic pictures / clips of tests/golden/cases.py. Images go through LanguageBindImageProcessor.preprocess as PIL images (what
    app.py / inference_image.py hand over); clips through LanguageBindVideoProcessor under the 'decord' back-end (the config default,
    configuration_video.py:205) and through load_and_transform_video under 'opencv' (the function's default), the inference-time random
    flip pinned in both outcomes by seeding `random`."""
    import random
    import types

    from PIL import Image

    from oracle import ref_preproc as RP
    from tests.golden import cases
    pi, pv = RP.install()
    inputs = cases.preproc_inputs()
    out = {"torch_version": np.array(torch.__version__)}
    cfg = types.SimpleNamespace(vision_config=types.SimpleNamespace(video_decode_backend="decord", num_frames=8))
    # ---- images: one call on the list of PIL images, as mm_utils / app.py do ------------------------------------------------------
    proc = RP.make_processor(pi.LanguageBindImageProcessor, cfg)
    assert proc.image_mean == O.OPENAI_DATASET_MEAN and proc.crop_size == {"height": 224, "width": 224}
    names = list(cases.PREPROC_IMAGES)
    pvs = proc.preprocess([Image.fromarray(inputs[n][0].numpy()) for n in names], return_tensors="pt")["pixel_values"]
    assert pvs.shape == (len(names), 3, 224, 224) and pvs.dtype == torch.float32
    for n, y in zip(names, pvs):
        out[n + "_sub"], out[n + "_sums"] = summarize(y, cases.PREPROC_IMG_STRIDE)
        out[n + "_in_checksum"] = np.float64(inputs[n].double().sum())
    # ---- clips ----------------------------------------------------------------------------------------------------------------------
    # a seed whose first random.random() is < 0.5 (RandomHorizontalFlipVideo flips) and one whose first draw is >= 0.5 (it does not)
    seed_flip = next(k for k in range(100) if random.Random(k).random() < 0.5)
    seed_keep = next(k for k in range(100) if random.Random(k).random() >= 0.5)
    vproc = RP.make_processor(pv.LanguageBindVideoProcessor, cfg)
    for n in cases.PREPROC_CLIPS:
        RP.CLIPS[n + ".mp4"] = inputs[n]
        out[n + "_in_checksum"] = np.float64(inputs[n].double().sum())
        for tag, seed in (("keep", seed_keep), ("flip", seed_flip)):
            if tag == "flip" and n != "clip_landscape":
                continue
            random.seed(seed)
            RP.READ_LOG.clear()
            y = vproc(images=[n + ".mp4"], return_tensors="pt")["pixel_values"]           # 'decord' back-end
            assert y.shape == (1, 3, 8, 224, 224) and y.dtype == torch.float32
            out[f"{n}_{tag}_idx"] = np.array(RP.READ_LOG, dtype=np.int64)
            out[f"{n}_{tag}_sub"], out[f"{n}_{tag}_sums"] = summarize(y[0], cases.PREPROC_CLIP_STRIDE)
            # the 'opencv' back-end (frame-by-frame reads, BGR -> RGB) must give the same clip
            random.seed(seed)
            RP.READ_LOG.clear()
            cfg_cv = types.SimpleNamespace(vision_config=types.SimpleNamespace(video_decode_backend="opencv", num_frames=8))
            y2 = pv.load_and_transform_video(n + ".mp4", pv.get_video_transform(cfg_cv), video_decode_backend="opencv", num_frames=8)
            assert RP.READ_LOG == out[f"{n}_{tag}_idx"].tolist() and torch.equal(y2, y[0]), "decord and opencv back-ends disagree"
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "preproc_ref.npz"), **out)
    print("preproc_ref.npz", {k: getattr(v, "shape", v) for k, v in out.items() if k.endswith(("_sub", "_idx"))})


if __name__ == "__main__":
    if "--ref-only" not in sys.argv:
        main()
    gen_ref()
