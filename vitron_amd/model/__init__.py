"""vitron_amd.model -- mirrors the export surface of the reference's vitron/model/__init__.py."""
from .language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM, VitronLlamaForCausalLM  # noqa: F401
from .builder import load_pretrained_model  # noqa: F401
