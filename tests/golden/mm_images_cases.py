"""Inputs of the mm_utils image-helper goldens (tests/golden/mm_utils_images.npz): small deterministic pictures and a stub image processor,
shared by the generating script (which runs the REFERENCE's functions on them) and the test (which runs vitron_amd.mm_utils)."""
import base64
import io

import numpy as np
import torch

SIZES = [(30, 20), (20, 31), (25, 25), (1, 7)]          # (width, height)
FILL = (10, 20, 30)


def picture(w, h):
    from PIL import Image
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([(7 * x + 3 * y) % 256, (5 * x * y + 11) % 256, (x + 13 * y) % 256], -1).astype(np.uint8)
    return Image.fromarray(a, "RGB")


def png_base64(img) -> str:
    buf = io.BytesIO()
    img.save(buf, format="PNG")
    return base64.b64encode(buf.getvalue()).decode("ascii")


class StubProcessor:
    """An image processor in the shape mm_utils uses one (CLIPImageProcessor's surface): `image_mean`, `preprocess(image, return_tensors)`
    and `__call__(images, return_tensors)`, each returning {'pixel_values': [...]}: the picture's top-left 16 x 16 as CHW floats."""
    image_mean = (0.48145466, 0.4578275, 0.40821073)

    def _one(self, img):
        a = np.zeros((16, 16, 3), dtype=np.float32)
        src = np.asarray(img.convert("RGB"), dtype=np.float32)[:16, :16]
        a[: src.shape[0], : src.shape[1]] = src
        return torch.from_numpy(a / 255.0).permute(2, 0, 1).contiguous()

    def preprocess(self, image, return_tensors=None):
        return {"pixel_values": [self._one(image)]}

    def __call__(self, images, return_tensors=None):
        return {"pixel_values": torch.stack([self._one(im) for im in images])}
