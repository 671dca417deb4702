"""Import the reference's OWN python modules (read-only, from /root/reference) in the build container.

TEST INFRASTRUCTURE ONLY -- used by tests/golden/make_golden.py to generate golden vectors from the reference
itself and by tests that cross-check the oracle when /root/reference is present. /root/reference does not exist
on the GPU box, so nothing on the -m gpu path, smoke() or bench.py imports this module.

The reference cannot be imported as-is here: vitron/__init__.py pulls in training code that
needs peft/deepspeed, modeling_video.py imports names removed after transformers 4.31, processing_*.py need
torchvision/decord/cv2/pytorchvideo. The shim stubs the missing third-party modules,
registers bare `vitron` / `vitron.model` packages so their __init__ side effects never run, and no-ops the
AutoConfig registration that collides with transformers 5.x. Nothing under /root/reference is modified.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("VITRON_REFERENCE_ROOT", "/root/reference")
# the staged archive of exactly the modules imported below (oracle/stage_ref.py, written by __graft_entry__.build() in the build
# container; git-ignored, travels to the GPU box): used when the reference tree itself is absent -- bench.py's cpu_baseline leg
STAGED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "vitron_ref.zip")
_unpacked = None


def _resolve_root() -> str:
    """REFERENCE_ROOT if the tree is there; else the staged archive unpacked into a temporary directory (once per process)."""
    global REFERENCE_ROOT, _unpacked
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "vitron", "model")):
        return REFERENCE_ROOT
    if os.path.exists(STAGED):
        if _unpacked is None:
            import atexit
            import shutil
            import tempfile
            import zipfile
            _unpacked = tempfile.mkdtemp(prefix="vitron_ref_")
            atexit.register(shutil.rmtree, _unpacked, ignore_errors=True)
            with zipfile.ZipFile(STAGED) as z:
                z.extractall(_unpacked)
        REFERENCE_ROOT = _unpacked
    return REFERENCE_ROOT


def available() -> bool:
    return os.path.isdir(os.path.join(_resolve_root(), "vitron", "model"))


def source() -> str:
    """'tree' (the reference tree itself), 'staged' (oracle/_ref/vitron_ref.zip) or 'absent'."""
    if not available():
        return "absent"
    return "staged" if _unpacked is not None and REFERENCE_ROOT == _unpacked else "tree"


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, n):
        return _Any()


_installed = False


def install():
    """Install the stubs once; returns a namespace with the reference modules."""
    global _installed
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    import transformers  # noqa: F401  (must be imported BEFORE torchvision is stubbed)
    import transformers.models.clip.modeling_clip as mc
    from transformers import AutoConfig, AutoModelForCausalLM

    if not _installed:
        def stub(name, **attrs):
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
            return m

        if "peft" not in sys.modules:
            stub("peft", LoraConfig=object, get_peft_model=lambda *a, **k: None)
        for n in ["cv2", "decord", "torchvision", "torchvision.transforms", "torchvision.transforms._transforms_video",
                  "pytorchvideo", "pytorchvideo.data", "pytorchvideo.data.encoded_video", "pytorchvideo.transforms"]:
            if n not in sys.modules:
                stub(n).__getattr__ = lambda name: _Any()
        sys.modules["decord"].bridge = _Any()
        if not hasattr(mc, "_expand_mask"):
            mc._expand_mask = lambda *a, **k: None
        if not hasattr(mc, "clip_loss"):
            mc.clip_loss = lambda *a, **k: None
        for pkg, path in [("vitron", os.path.join(REFERENCE_ROOT, "vitron")),
                          ("vitron.model", os.path.join(REFERENCE_ROOT, "vitron", "model"))]:
            m = types.ModuleType(pkg)
            m.__path__ = [path]
            sys.modules[pkg] = m
        AutoConfig.register = staticmethod(lambda *a, **k: None)
        AutoModelForCausalLM.register = staticmethod(lambda *a, **k: None)
        _installed = True

    ns = types.SimpleNamespace()
    ns.arch = importlib.import_module("vitron.model.llava_arch")
    ns.llava_llama = importlib.import_module("vitron.model.language_model.llava_llama")
    ns.modeling_video = importlib.import_module("vitron.model.multimodal_encoder.languagebind.video.modeling_video")
    ns.modeling_image = importlib.import_module("vitron.model.multimodal_encoder.languagebind.image.modeling_image")
    ns.configuration_image = importlib.import_module(
        "vitron.model.multimodal_encoder.languagebind.image.configuration_image")
    ns.configuration_video = importlib.import_module(
        "vitron.model.multimodal_encoder.languagebind.video.configuration_video")
    ns.mm_utils = importlib.import_module("vitron.mm_utils")
    ns.constants = importlib.import_module("vitron.constants")
    ns.region_layer = _load_standalone("vitron_ref_region_layer", "vitron/model/region_extractor/layer.py")
    ns.projector_builder = _load_standalone("vitron_ref_projector_builder", "vitron/model/multimodal_projector/builder.py")
    return ns


def _load_standalone(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
