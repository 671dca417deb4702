"""Device-side pre-processing: mirror of the reference's LanguageBind image / video processors
(vitron/model/multimodal_encoder/languagebind/image/processing_image.py, .../video/processing_video.py) with the
tensor work (resize + centre crop + scale + normalise) fused into one HIP kernel (vt_preprocess).

Decoding stays on the host (PIL / decord / OpenCV, whatever the caller has); the processors take decoded uint8 frames.
The reference's RandomHorizontalFlipVideo(p=0.5) -- applied at inference, processing_video.py:52 -- is an explicit
`flip` argument here (default off) because it changes results.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)   # processing_image.py:7
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)   # processing_image.py:8


def sample_frame_indices(duration: int, num_frames: int) -> np.ndarray:
    """Uniform frame sampling of the decord / opencv back-ends (processing_video.py:96,104)."""
    return np.linspace(0, duration - 1, num_frames, dtype=int)


def preprocess_frames(frames: torch.Tensor, size: int, bicubic: bool, clip_layout: bool, flip: bool = False,
                      dtype=None, mean=OPENAI_DATASET_MEAN, std=OPENAI_DATASET_STD) -> torch.Tensor:
    """frames: [F,H,W,3] uint8 (or float in [0,1]... pass uint8 for the /255) on the GPU.
    Returns [F,3,S,S] (images) or [3,F,S,S] (clip_layout: the (C,T,H,W) clip the video tower takes) in `dtype`
    (bf16 -- the package default -- / fp16: the towers' operand dtype, or fp32)."""
    dtype = dtype if dtype is not None else _lib.torch_dtype()
    lib = _lib.load_any() if dtype == torch.float32 else _lib.lib_for(dtype)
    if not frames.is_cuda:
        raise _lib.VitronHipError("preprocess_frames: frames must be on the GPU")
    if frames.dim() != 4 or frames.shape[-1] != 3:
        raise _lib.VitronHipError(f"preprocess_frames: expected [F,H,W,3], got {tuple(frames.shape)}")
    if frames.dtype not in (torch.uint8, torch.float32):
        frames = frames.float()
    frames = frames.contiguous()
    F_, H, W, _ = frames.shape
    S = int(size)
    out = torch.empty((3, F_, S, S) if clip_layout else (F_, 3, S, S), device=frames.device, dtype=dtype)
    sc, sf = (F_ * S * S, S * S) if clip_layout else (S * S, 3 * S * S)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    dt = _lib.DTYPE_F32 if dtype == torch.float32 else _lib.DTYPE_BF16      # DTYPE_BF16 = "the library's 16-bit operand format"
    _lib.check(lib.vt_preprocess(frames.data_ptr(), int(frames.dtype == torch.uint8), 1, F_, H, W, int(bicubic), S, m, s, int(flip),
                                 out.data_ptr(), dt, sc, sf, torch.cuda.current_stream().cuda_stream), "vt_preprocess", lib)
    return out


def decode_video_frames(path: str, num_frames: int, backend: str = "opencv") -> torch.Tensor:
    """The decode + uniform sampling half of the reference's load_and_transform_video (processing_video.py:72-114, 'decord' and
    'opencv' back-ends; 'pytorchvideo' decodes through decord as well): ONLY the `num_frames` frames at
    linspace(0, duration-1, num_frames) are decoded and returned as one uint8 [T,H,W,3] host tensor, so one upload feeds the
    fused resize/crop/normalise kernel. Video codecs stay a host-library matter (no decoder library ships in the offline image:
    the import error says which one to install); frames that are already decoded go straight to LanguageBindVideoProcessor."""
    idx = None
    if backend in ("decord", "pytorchvideo"):
        try:
            import decord
        except ImportError as e:
            raise ImportError("video files need the `decord` package (or backend='opencv' with `opencv-python`); "
                              "already decoded frames can be passed as a uint8 [frames,H,W,3] tensor") from e
        vr = decord.VideoReader(path, ctx=decord.cpu(0))
        idx = sample_frame_indices(len(vr), num_frames)
        batch = vr.get_batch(idx.tolist())
        return torch.from_numpy(batch.asnumpy() if hasattr(batch, "asnumpy") else np.asarray(batch))
    if backend == "opencv":
        try:
            import cv2
        except ImportError as e:
            raise ImportError("video files need `opencv-python` (or backend='decord'); already decoded frames can be passed "
                              "as a uint8 [frames,H,W,3] tensor") from e
        cap = cv2.VideoCapture(path)
        idx = sample_frame_indices(int(cap.get(cv2.CAP_PROP_FRAME_COUNT)), num_frames)
        frames = []
        for i in idx.tolist():
            cap.set(1, i)
            ok, frame = cap.read()
            if not ok:
                cap.release()
                raise ValueError(f"could not read frame {i} of {path}")
            frames.append(torch.from_numpy(cv2.cvtColor(frame, cv2.COLOR_BGR2RGB)))
        cap.release()
        return torch.stack(frames)
    raise NameError("video_decode_backend should specify in (pytorchvideo, decord, opencv)")   # reference processing_video.py:113


def _to_u8_hwc(img, device) -> torch.Tensor:
    if isinstance(img, str):        # a file path, as inference_image.py:24 passes it (processing_image.py:29-31)
        from PIL import Image
        img = Image.open(img).convert("RGB")
    if isinstance(img, torch.Tensor):
        t = img
    else:
        t = torch.from_numpy(np.array(img.convert("RGB") if hasattr(img, "convert") else img))   # np.array: a writable copy
    if t.dim() != 3 or t.shape[-1] != 3:
        raise ValueError(f"expected an HxWx3 image, got {tuple(t.shape)}")
    return t.to(device)


class LanguageBindImageProcessor:
    """preprocess(images)['pixel_values'] -> [N,3,S,S]; same attributes app.py / mm_utils read (image_mean, crop_size)."""

    def __init__(self, config=None, image_size: int = 224, device="cuda", dtype=None):
        vc = getattr(config, "vision_config", config)
        self.size = int(getattr(vc, "image_size", image_size) or image_size)
        self.device, self.dtype = device, dtype
        self.image_mean = OPENAI_DATASET_MEAN
        self.crop_size = {"height": self.size, "width": self.size}

    def preprocess(self, images, return_tensors=None, **kwargs):
        images = images if isinstance(images, (list, tuple)) else [images]
        outs = [preprocess_frames(_to_u8_hwc(im, self.device).unsqueeze(0), self.size, True, False, dtype=self.dtype)[0] for im in images]
        return {"pixel_values": torch.stack(outs)}

    __call__ = preprocess


class LanguageBindVideoProcessor:
    """__call__(videos)['pixel_values'] -> [N,3,T,S,S]; a video is a decoded uint8 tensor [frames,H,W,3] (all frames of
    the file: `num_frames` are sampled uniformly like the reference) or an already sampled [T,H,W,3] one."""

    def __init__(self, config=None, image_size: int = 224, num_frames: int = 8, device="cuda", dtype=None, flip: bool = False,
                 video_decode_backend: str = "opencv"):
        vc = getattr(config, "vision_config", config)
        self.video_decode_backend = getattr(vc, "video_decode_backend", video_decode_backend) or video_decode_backend
        self.size = int(getattr(vc, "image_size", image_size) or image_size)
        self.num_frames = int(getattr(vc, "num_frames", num_frames) or num_frames)
        self.device, self.dtype, self.flip = device, dtype, flip
        self.image_mean = OPENAI_DATASET_MEAN
        self.crop_size = {"height": self.size, "width": self.size}

    def __call__(self, videos, return_tensors=None, **kwargs):
        videos = videos if isinstance(videos, (list, tuple)) else [videos]
        outs = []
        for v in videos:
            if isinstance(v, str):      # a file: decode only the sampled frames, one upload, one kernel
                v = decode_video_frames(v, self.num_frames, self.video_decode_backend)
            if not isinstance(v, torch.Tensor):
                v = torch.from_numpy(np.asarray(v))
            if v.shape[0] != self.num_frames:
                v = v[torch.from_numpy(sample_frame_indices(v.shape[0], self.num_frames))]
            outs.append(preprocess_frames(v.to(self.device), self.size, False, True, flip=self.flip, dtype=self.dtype))
        return {"pixel_values": torch.stack(outs)}
