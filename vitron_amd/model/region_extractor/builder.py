"""Factory for the region extractor; same entry point the reference exposes in
vitron/model/region_extractor/builder.py:3-5 (called from llava_arch.py when a checkpoint is assembled).

The widths come from the LLaVA config: `mm_hidden_size` is the tower width (1024 for LanguageBind ViT-L/14),
`hidden_size` the decoder width (4096 for Vicuna-7B). `mm_region_image_size` is an extension: the side of the
pixel grid that region masks are drawn on (the reference hard-codes 224 in layer.py).
"""
from . import layer as _layer

_DEFAULT_MASK_SIDE = 224


def build_region_extractor(config, delay_load=False, **kwargs):
    del delay_load, kwargs  # accepted for call compatibility; packing happens lazily on first use
    widths = dict(in_dim=int(config.mm_hidden_size), out_dim=int(config.hidden_size))
    side = int(getattr(config, "mm_region_image_size", _DEFAULT_MASK_SIDE))
    return _layer.RegionExtractor(widths["in_dim"], widths["out_dim"], image_size=side)
