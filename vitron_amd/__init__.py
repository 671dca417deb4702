"""vitron_amd -- MI355X (gfx950) implementation of Vitron's multimodal forward pass.

LanguageBind ViT image/video tower -> region_extractor -> mm_projector -> LLaMA decoder (prefill + paged-KV
decode) as hand-written HIP kernels in libvitron_hip.so (C ABI: include/vitron_hip.h), bound with ctypes and
wrapped in the reference's Python surface (vitron.model: load_pretrained_model, LlavaLlamaForCausalLM,
encode_images / encode_videos). See DESIGN.md and INTEGRATION.md.
"""
from .constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX, OBJS_TOKEN_INDEX  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):  # lazy: importing the package must not require torch.cuda or the shared library
    if name in ("LlavaLlamaForCausalLM", "VitronLlamaForCausalLM", "LlavaConfig", "load_pretrained_model"):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
