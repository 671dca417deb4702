"""Stage the reference's own Python modules for the hot path into oracle/_ref/ (git-ignored, travels to the GPU box).

TEST / MEASUREMENT INFRASTRUCTURE ONLY. The reference (/root/reference) exists in the build container but not on the GPU box,
and its modules are Python: there is nothing to compile. What `bench.py`'s `cpu_baseline` leg needs on the GPU box -- "the
reference's CPU path timed beside it on the same box's host cores" (BASELINE.json north_star) -- is the reference's
own code. This recipe packs EXACTLY the files oracle/ref_shim.py imports (the list below = SURVEY.md 8(c)'s list) into ONE archive

    oracle/_ref/vitron_ref.zip           (listed in .gitignore: never enters the history; NOT in .gpurunignore: it travels)

by reading them where they lie under /root/reference; nothing is copied into the tracked tree. oracle/ref_shim.py unpacks the
archive into a temporary directory when /root/reference is absent. `__graft_entry__.build()` calls stage() whenever
/root/reference is present. Only bench.py's cpu_baseline leg (kind "reference") and tests/golden/*.py go through ref_shim; the
product (vitron_amd/) never does.
"""
from __future__ import annotations

import os
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_DIR = os.path.join(ROOT, "oracle", "_ref")
ARCHIVE = os.path.join(OUT_DIR, "vitron_ref.zip")

# what ref_shim.install() ends up importing (sys.modules after install(), probed in the build container)
FILES = [
    "vitron/constants.py",
    "vitron/mm_utils.py",
    "vitron/model/llava_arch.py",
    "vitron/model/language_model/llava_llama.py",
    "vitron/model/multimodal_encoder/builder.py",
    "vitron/model/multimodal_encoder/clip_encoder.py",
    "vitron/model/multimodal_encoder/languagebind/__init__.py",
    "vitron/model/multimodal_encoder/languagebind/image/configuration_image.py",
    "vitron/model/multimodal_encoder/languagebind/image/modeling_image.py",
    "vitron/model/multimodal_encoder/languagebind/image/processing_image.py",
    "vitron/model/multimodal_encoder/languagebind/image/tokenization_image.py",
    "vitron/model/multimodal_encoder/languagebind/video/configuration_video.py",
    "vitron/model/multimodal_encoder/languagebind/video/modeling_video.py",
    "vitron/model/multimodal_encoder/languagebind/video/processing_video.py",
    "vitron/model/multimodal_encoder/languagebind/video/tokenization_video.py",
    "vitron/model/multimodal_projector/builder.py",
    "vitron/model/region_extractor/builder.py",
    "vitron/model/region_extractor/layer.py",
]


def stage(reference_root: str = "/root/reference") -> str | None:
    """Write oracle/_ref/vitron_ref.zip from `reference_root`; returns its path, or None when the reference tree is absent."""
    if not os.path.isdir(os.path.join(reference_root, "vitron", "model")):
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = ARCHIVE + f".tmp{os.getpid()}"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for rel in FILES:
            src = os.path.join(reference_root, rel)
            if not os.path.exists(src):
                raise FileNotFoundError(f"{src}: the reference tree does not hold a file ref_shim needs")
            # fixed timestamps: the archive is a pure function of the file contents
            info = zipfile.ZipInfo(rel, date_time=(2024, 10, 20, 0, 0, 0))
            info.compress_type = zipfile.ZIP_DEFLATED
            with open(src, "rb") as f:
                z.writestr(info, f.read())
    os.replace(tmp, ARCHIVE)
    return ARCHIVE


if __name__ == "__main__":
    print(stage())
