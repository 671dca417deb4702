"""Tower factory: the two entry points the reference exposes in vitron/model/multimodal_encoder/builder.py:7-24.

The reference picks the tower class from the checkpoint path: `openai*` / `laion*` names -> the plain HF-CLIP tower
(CLIPVisionTower, builder.py:12), `...LanguageBind_Image` -> LanguageBindImageTower, `...LanguageBind_Video_merge` ->
LanguageBindVideoTower. All three run on the same kernels (vt_vit_forward); they differ in the loader mapping.
"""
from . import languagebind as _lb
from .clip_encoder import CLIPVisionTower


def _path(cfg, attr, legacy):
    path = getattr(cfg, attr, None)
    return getattr(cfg, legacy, None) if path is None else path


def build_image_tower(image_tower_cfg, **kwargs):
    image_tower = _path(image_tower_cfg, "mm_image_tower", "image_tower")
    if image_tower is not None and (str(image_tower).startswith("openai") or str(image_tower).startswith("laion")):
        return CLIPVisionTower(image_tower, args=image_tower_cfg, **kwargs)
    if image_tower is not None and str(image_tower).endswith("LanguageBind_Image"):
        return _lb.LanguageBindImageTower(image_tower, args=image_tower_cfg, cache_dir="./cache_dir", **kwargs)
    raise ValueError(f"Unknown image tower: {image_tower}")


def build_video_tower(video_tower_cfg, **kwargs):
    video_tower = _path(video_tower_cfg, "mm_video_tower", "video_tower")
    if video_tower is not None and str(video_tower).endswith("LanguageBind_Video_merge"):
        return _lb.LanguageBindVideoTower(video_tower, args=video_tower_cfg, cache_dir="./cache_dir", **kwargs)
    raise ValueError(f"Unknown video tower: {video_tower}")
