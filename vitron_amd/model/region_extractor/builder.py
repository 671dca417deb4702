"""reference vitron/model/region_extractor/builder.py:3-5."""
from .layer import RegionExtractor


def build_region_extractor(config, delay_load=False, **kwargs):
    return RegionExtractor(config.mm_hidden_size, config.hidden_size,
                           image_size=getattr(config, "mm_region_image_size", 224))
