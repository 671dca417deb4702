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
fixture runs where /root/reference exists and is skipped elsewhere."""
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


if __name__ == "__main__":
    main()
