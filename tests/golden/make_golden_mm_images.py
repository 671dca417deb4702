"""tests/golden/mm_utils_images.npz: outputs of the REFERENCE's own vitron/mm_utils.py helpers (load_image_from_base64, expand2square,
process_images: mm_utils.py:48-77) on the pictures of tests/golden/mm_images_cases.py. Run in the build container:

    python tests/golden/make_golden_mm_images.py

The reference module is imported through oracle/ref_shim.py (nothing under /root/reference is modified or copied)."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from tests.golden import mm_images_cases as MC  # noqa: E402


def main():
    ns = ref_shim.install()
    mu = ns.mm_utils
    out = {}
    proc = MC.StubProcessor()
    pics = [MC.picture(w, h) for w, h in MC.SIZES]
    for i, im in enumerate(pics):
        out[f"square_{i}"] = np.asarray(mu.expand2square(im, MC.FILL))
        out[f"b64_{i}"] = np.asarray(mu.load_image_from_base64(MC.png_base64(im)))
    pad = mu.process_images(pics, proc, types.SimpleNamespace(image_aspect_ratio="pad"))
    out["process_pad"] = pad.numpy()
    out["process_plain"] = mu.process_images(pics, proc, types.SimpleNamespace(image_aspect_ratio=None)).numpy()
    out["process_missing_attr"] = mu.process_images(pics[:2], proc, types.SimpleNamespace()).numpy()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mm_utils_images.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
