"""Tower factory: the two entry points the reference exposes in vitron/model/multimodal_encoder/builder.py:7-24.

The reference picks the tower class from the tail of the checkpoint path; only the LanguageBind towers are on the
hot path, so the OpenAI/LAION CLIP branch of the reference is rejected here instead of silently
taking a different code path.
"""
from . import languagebind as _lb

# modality -> (config attribute carrying the tower path, legacy attribute, path suffix -> tower class)
_TOWERS = {
    "image": ("mm_image_tower", "image_tower", {"LanguageBind_Image": _lb.LanguageBindImageTower}),
    "video": ("mm_video_tower", "video_tower", {"LanguageBind_Video_merge": _lb.LanguageBindVideoTower}),
}


def _build(modality, cfg, kwargs):
    attr, legacy, by_suffix = _TOWERS[modality]
    path = getattr(cfg, attr, None)
    if path is None:
        path = getattr(cfg, legacy, None)
    for suffix, cls in by_suffix.items():
        if path is not None and str(path).endswith(suffix):
            return cls(path, args=cfg, cache_dir="./cache_dir", **kwargs)
    raise ValueError(f"Unknown {modality} tower: {path}")


def build_image_tower(image_tower_cfg, **kwargs):
    return _build("image", image_tower_cfg, kwargs)


def build_video_tower(video_tower_cfg, **kwargs):
    return _build("video", video_tower_cfg, kwargs)
