"""Run the REFERENCE's own pre-processing code (languagebind/image/processing_image.py, languagebind/video/processing_video.py) in a
container that has none of the third-party libraries it composes.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): used by tests/golden/make_golden_preproc.py to generate
tests/golden/preproc_ref.npz in the build container. The product (vitron_amd/) never imports this module.

What this gives and what it cannot: the reference's processors are compositions of torchvision 0.15.2 (pyproject.toml:16),
pytorchvideo 0.1.5 (requirements.txt:40), decord 0.6.0 (requirements.txt:2) and opencv-python 4.9.0.80 (requirements.txt:32)
calls. None of the four is installed offline and none is vendored under /root/reference, so their code cannot run here. This module
restates the handful of primitives the reference calls -- each from the library's published behaviour at the pinned version, in a few
lines of torch -- registers them under the libraries' module names, and then imports and runs the reference's two files UNMODIFIED.
The fixture therefore pins everything the REFERENCE wrote: which transforms, in which order (the video path normalises BEFORE it
resizes), the hard-wired 224 of Resize / CenterCrop / ShortSideScale / CenterCropVideo, the OPENAI mean / std constants, the x / 255,
the (T,H,W,C) -> (C,T,H,W) permute, the uniform frame sampling np.linspace(0, duration - 1, num_frames, dtype=int) of the decord and
opencv back-ends (through fake readers that serve a synthetic clip), the BGR -> RGB swap, the inference-time
RandomHorizontalFlipVideo(p = 0.5), the stacking into `pixel_values`. The arithmetic INSIDE the primitives (size rules, crop
offsets, interpolate call) remains a restatement of the two libraries: "parity unpinned" for that part.

Primitives restated (library, version, the behaviour followed):
  torchvision.transforms 0.15.2
    ToTensor        PIL RGB / HWC uint8 ndarray -> CHW float32 / 255
    Resize(int)     TENSOR path: short side -> size, long side int(size * long / short) (_compute_resized_output_size); unchanged if the
                    size already matches; antialias="warn" means NO antialias for tensors in 0.15; torch.nn.functional.interpolate(
                    mode, align_corners=False); no clamp for float input
    CenterCrop      top = int(round((h - th) / 2.0)), left = int(round((w - tw) / 2.0))
    Normalize       (x - mean[:, None, None]) / std[:, None, None]
    Compose, Lambda
  torchvision.transforms._transforms_video 0.15.2 (clips are (C, T, H, W) float)
    NormalizeVideo  (clip - mean[:, None, None, None]) / std[:, None, None, None]
    CenterCropVideo same offsets as CenterCrop on the last two dimensions
    RandomHorizontalFlipVideo(p)   `if random.random() < p: clip = clip.flip(-1)`
  pytorchvideo.transforms 0.1.5
    ShortSideScale(size)           w < h: (floor(h / w * size), size) else (size, floor(w / h * size)); interpolate(bilinear, align_corners=False)
    UniformTemporalSubsample(n)    index_select at clamp(linspace(0, t - 1, n)).long() on dim -3
    ApplyTransformToKey
  decord 0.6.0 / cv2 4.9 (fake readers over an in-memory clip: VideoReader / len / get_batch with the torch bridge; VideoCapture /
    get(CAP_PROP_FRAME_COUNT) / set(1, i) / read (BGR) / release, cvtColor(COLOR_BGR2RGB))
"""
from __future__ import annotations

import importlib.util
import math
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

from . import ref_shim


# ---- torchvision.transforms -------------------------------------------------------------------------------------------
class InterpolationMode:
    NEAREST, BILINEAR, BICUBIC = "nearest", "bilinear", "bicubic"


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class Lambda:
    def __init__(self, lambd):
        self.lambd = lambd

    def __call__(self, x):
        return self.lambd(x)


class ToTensor:
    def __call__(self, pic):
        a = np.array(pic)                       # PIL (mode RGB) or HWC ndarray
        t = torch.from_numpy(a).permute(2, 0, 1).contiguous()
        return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t


class Resize:
    def __init__(self, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias="warn"):
        self.size, self.interpolation, self.antialias = size, interpolation, antialias

    def __call__(self, img):
        assert isinstance(img, torch.Tensor), "the reference resizes behind ToTensor: the tensor path"
        h, w = img.shape[-2:]
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = self.size, int(self.size * long / short)
        new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
        if (h, w) == (new_h, new_w):
            return img
        antialias = False if self.antialias == "warn" else bool(self.antialias)
        return F.interpolate(img[None], size=[new_h, new_w], mode=self.interpolation, align_corners=False, antialias=antialias)[0]


def _center_offsets(h, w, th, tw):
    return int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))


class CenterCrop:
    def __init__(self, size):
        self.size = (int(size), int(size)) if isinstance(size, (int, float)) else tuple(size)

    def __call__(self, img):
        th, tw = self.size
        h, w = img.shape[-2:]
        assert h >= th and w >= tw, "(the padding branch of CenterCrop is never reached behind Resize(224))"
        i, j = _center_offsets(h, w, th, tw)
        return img[..., i:i + th, j:j + tw]


class Normalize:
    def __init__(self, mean, std, inplace=False):
        self.mean, self.std = mean, std

    def __call__(self, t):
        mean = torch.as_tensor(self.mean, dtype=t.dtype)
        std = torch.as_tensor(self.std, dtype=t.dtype)
        return t.clone().sub_(mean[:, None, None]).div_(std[:, None, None])


# ---- torchvision.transforms._transforms_video -------------------------------------------------------------------------
class NormalizeVideo:
    def __init__(self, mean, std, inplace=False):
        self.mean, self.std = mean, std

    def __call__(self, clip):
        mean = torch.as_tensor(self.mean, dtype=clip.dtype)
        std = torch.as_tensor(self.std, dtype=clip.dtype)
        return clip.clone().sub_(mean[:, None, None, None]).div_(std[:, None, None, None])


class CenterCropVideo:
    def __init__(self, crop_size):
        self.crop_size = (int(crop_size), int(crop_size)) if isinstance(crop_size, (int, float)) else tuple(crop_size)

    def __call__(self, clip):
        th, tw = self.crop_size
        h, w = clip.shape[-2:]
        assert h >= th and w >= tw, "height and width must be no smaller than crop_size"
        i, j = _center_offsets(h, w, th, tw)
        return clip[..., i:i + th, j:j + tw]


class RandomHorizontalFlipVideo:
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, clip):
        if random.random() < self.p:
            clip = clip.flip(-1)
        return clip


class RandomCropVideo:      # imported by the reference (processing_video.py:12), never used
    def __init__(self, *a, **k):
        raise NotImplementedError


# ---- pytorchvideo.transforms ------------------------------------------------------------------------------------------
class ShortSideScale:
    def __init__(self, size, interpolation="bilinear", backend="pytorch"):
        self.size, self.interpolation = size, interpolation

    def __call__(self, x):
        assert x.dim() == 4 and x.dtype == torch.float32
        _, _, h, w = x.shape
        if w < h:
            new_h, new_w = int(math.floor((float(h) / w) * self.size)), self.size
        else:
            new_h, new_w = self.size, int(math.floor((float(w) / h) * self.size))
        return F.interpolate(x, size=(new_h, new_w), mode=self.interpolation, align_corners=False)


class UniformTemporalSubsample:
    def __init__(self, num_samples, temporal_dim=-3):
        self.n, self.dim = num_samples, temporal_dim

    def __call__(self, x):
        t = x.shape[self.dim]
        idx = torch.clamp(torch.linspace(0, t - 1, self.n), 0, t - 1).long()
        return torch.index_select(x, self.dim, idx)


class ApplyTransformToKey:
    def __init__(self, key, transform):
        self.key, self.transform = key, transform

    def __call__(self, x):
        x[self.key] = self.transform(x[self.key])
        return x


# ---- fake decoders over in-memory clips -------------------------------------------------------------------------------
CLIPS = {}          # "path" -> uint8 [frames, H, W, 3] RGB tensor (what a decoder would hand out)
READ_LOG = []       # frame indices the reference asked for, in order (per call of load_and_transform_video)


class _VideoReader:
    def __init__(self, path, ctx=None):
        self.clip = CLIPS[path]

    def __len__(self):
        return self.clip.shape[0]

    def get_batch(self, idx):
        idx = [int(i) for i in idx]
        READ_LOG.extend(idx)
        return self.clip[idx]          # the 'torch' bridge: a torch tensor (T, H, W, C)


class _VideoCapture:
    def __init__(self, path):
        self.clip, self.pos = CLIPS[path], 0

    def get(self, prop):
        assert prop == _cv2.CAP_PROP_FRAME_COUNT
        return float(self.clip.shape[0])

    def set(self, prop, value):
        assert prop == 1               # cv2.CAP_PROP_POS_FRAMES, as the reference writes it (processing_video.py:108)
        self.pos = int(value)

    def read(self):
        READ_LOG.append(self.pos)
        frame = self.clip[self.pos].numpy()[:, :, ::-1].copy()      # OpenCV hands out BGR
        self.pos += 1
        return True, frame

    def release(self):
        pass


_cv2 = types.SimpleNamespace(CAP_PROP_FRAME_COUNT=7, COLOR_BGR2RGB=4)


def _cvt_color(frame, code):
    assert code == _cv2.COLOR_BGR2RGB
    return np.ascontiguousarray(frame[:, :, ::-1])


# ---- registration + import of the reference's two files ----------------------------------------------------------------
def install():
    """Register the primitives under the third-party module names and import the reference's processing_image.py / processing_video.py
    (unmodified) as standalone modules. Returns (processing_image, processing_video)."""
    ref_shim.install()                  # transformers first, bare vitron packages, the other stubs

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tvt = mod("torchvision.transforms", Compose=Compose, Lambda=Lambda, ToTensor=ToTensor, Resize=Resize, CenterCrop=CenterCrop,
              Normalize=Normalize, InterpolationMode=InterpolationMode)
    tvv = mod("torchvision.transforms._transforms_video", NormalizeVideo=NormalizeVideo, RandomCropVideo=RandomCropVideo,
              RandomHorizontalFlipVideo=RandomHorizontalFlipVideo, CenterCropVideo=CenterCropVideo)
    tvt._transforms_video = tvv
    mod("torchvision", transforms=tvt)
    ptt = mod("pytorchvideo.transforms", ApplyTransformToKey=ApplyTransformToKey, ShortSideScale=ShortSideScale,
              UniformTemporalSubsample=UniformTemporalSubsample)
    ev = mod("pytorchvideo.data.encoded_video", EncodedVideo=None)
    mod("pytorchvideo.data", encoded_video=ev)
    mod("pytorchvideo", transforms=ptt)
    mod("decord", VideoReader=_VideoReader, cpu=lambda i=0: ("cpu", i), bridge=types.SimpleNamespace(set_bridge=lambda name: None))
    mod("cv2", VideoCapture=_VideoCapture, cvtColor=_cvt_color, CAP_PROP_FRAME_COUNT=_cv2.CAP_PROP_FRAME_COUNT,
        COLOR_BGR2RGB=_cv2.COLOR_BGR2RGB)
    root = ref_shim.REFERENCE_ROOT
    out = []
    for name, rel in (("vitron_ref_processing_image", "vitron/model/multimodal_encoder/languagebind/image/processing_image.py"),
                      ("vitron_ref_processing_video", "vitron/model/multimodal_encoder/languagebind/video/processing_video.py")):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, rel))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        out.append(m)
    return tuple(out)


def make_processor(cls, config):
    """Construct one of the reference's processor classes. transformers 4.31's ProcessorMixin.__init__ accepts `attributes = []` with no
    arguments; the 5.x build installed here insists on a tokenizer argument, so the mixin's __init__ is a no-op for the duration of the
    constructor (the reference's own __init__ body runs unchanged: config, transform, image_processor, tokenizer = None)."""
    from transformers import ProcessorMixin
    saved = ProcessorMixin.__init__
    ProcessorMixin.__init__ = lambda self, *a, **k: None
    try:
        return cls(config)
    finally:
        ProcessorMixin.__init__ = saved
