"""Weight packing and the composite operators (ViT tower, projector, region extractor, LLaMA decoder + paged KV).

This is the "weight loader" side of the boundary: it takes state dicts in the REFERENCE's
parameter naming (what vitron/model/builder.py:53-86,149-163 would load from disk) and lays them out the way the
HIP kernels want them:
  * q/k/v projections fused into one [3D][D] matrix (ViT: q rows and bias pre-scaled by head_dim^-0.5, an exact
    power-of-two scaling in bf16),
  * LLaMA gate/up fused and row-interleaved in blocks of 16 so SwiGLU is a lane-local GEMM epilogue,
  * the patch-embedding conv flattened to [D][3*P*P] in (c,py,px) order and zero padded to a multiple of 64,
  * image-tower LoRA deltas (reference image/modeling_image.py:772-793) merged into the base weights,
  * biases / norm parameters / embeddings tables upcast to fp32 once.
Each Packed* object owns its device tensors and the ctypes struct that points at them, and exposes one
`forward` that is a single C-ABI call.

Operand format: every Packed* takes `dtype` (torch.bfloat16, the default, or torch.float16 -- the reference's inference dtype,
vitron/model/builder.py:47) and packs its GEMM weights in it; the dtype also selects the library build the object calls
(_lib.lib_for). fp16 checkpoints are kept bit for bit in fp16 mode (no detour through bf16); bf16 / fp32 weights are
rounded to fp16 once, saturating at +-65504.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from ._lib import PAGE_TOKENS

SD = Dict[str, torch.Tensor]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _op(t: torch.Tensor, device, dtype=torch.bfloat16) -> torch.Tensor:
    """A weight / activation in the operand format. fp16 saturates (a bf16 / fp32 value beyond +-65504 would become inf)."""
    if dtype == torch.float16 and t.dtype != torch.float16:
        return t.to(device=device, dtype=torch.float32).clamp_(-65504.0, 65504.0).to(torch.float16).contiguous()
    return t.to(device=device, dtype=dtype).contiguous()


def _dtype(dtype):
    dt = dtype if dtype is not None else _lib.torch_dtype()
    _lib.operand_of(dt)          # raises unless bf16 / fp16
    return dt


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


def merge_lora(sd: SD, lora_alpha: float = 16.0) -> SD:
    """Fold peft LoRA adapters (`X.base_layer.weight`, `X.lora_A.default.weight`, `X.lora_B.default.weight`) into
    plain `X.weight` entries: W += (alpha / r) * B @ A. Keys without adapters pass through."""
    out: SD = {}
    for k, v in sd.items():
        if ".lora_A." in k or ".lora_B." in k:
            continue
        k2 = k.replace("base_model.model.", "")
        if k2.endswith(".base_layer.weight"):
            stem = k[: -len(".base_layer.weight")]
            a = sd.get(stem + ".lora_A.default.weight")
            b = sd.get(stem + ".lora_B.default.weight")
            w = v.float()
            if a is not None and b is not None:
                w = w + (lora_alpha / a.shape[0]) * (b.float() @ a.float())
            out[k2[: -len(".base_layer.weight")] + ".weight"] = w.to(v.dtype)
        elif k2.endswith(".base_layer.bias"):
            out[k2[: -len(".base_layer.bias")] + ".bias"] = v
        else:
            out[k2] = v
    return out


class Workspace:
    """A grow-only device scratch buffer shared by the composite operators of one model."""

    def __init__(self, device):
        self.device = device
        self.buf: Optional[torch.Tensor] = None

    def get(self, nbytes: int) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = None
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self.buf


# =====================================================================================================================
def pair_lo(t):
    """The LOW half that travels with a tensor in precise level 2 (operand pairs: value = f32(t) + f32(pair_lo(t))); None otherwise."""
    return getattr(t, "_vt_lo", None)


def with_lo(t, lo):
    """Attach the low half of an operand pair to its high half (a plain attribute: views / copies made by the caller drop it)."""
    if lo is not None:
        t._vt_lo = lo
    return t


class PackedVit:
    """LanguageBind CLIP vision transformer in kernel layout (vt_vit_model)."""

    def __init__(self, sd: SD, cfg: dict, device, select_layer: int = -2, dtype=None):
        sd = merge_lora(sd, float(cfg.get("lora_alpha", 16.0))) if any(".lora_" in k for k in sd) else sd
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        self.dtype = dt = _dtype(dtype)
        self.lib = _lib.lib_for(dt)

        def _bf(t, dev):
            return _op(t, dev, dt)

        D, heads, P = cfg["hidden_size"], cfg["num_attention_heads"], cfg["patch_size"]
        if D != heads * 64:
            raise _lib.VitronHipError("PackedVit: kernels are specialised for head_dim 64")
        self.D, self.heads, self.P, self.I = D, heads, P, cfg["intermediate_size"]
        self.image_size = cfg["image_size"]
        self.G = self.image_size // P
        self.time_attn = bool(cfg.get("add_time_attn", False))
        self.num_frames = int(cfg.get("num_frames", 1))
        nl = cfg["num_hidden_layers"]
        self.run_layers = select_layer if select_layer >= 0 else nl + 1 + select_layer
        if not 0 <= self.run_layers <= nl:
            raise _lib.VitronHipError(f"PackedVit: select_layer {select_layer} out of range")
        self.k_pad = (3 * P * P + 63) // 64 * 64
        self._keep: List[torch.Tensor] = []
        dev = self.device
        scale = 64 ** -0.5

        def keep(t):
            self._keep.append(t)
            return t

        wp = torch.zeros((D, self.k_pad), dtype=dt, device=dev)
        wp[:, : 3 * P * P] = _bf(sd["embeddings.patch_embedding.weight"].reshape(D, 3 * P * P), dev)
        self.w_patch = keep(wp)
        self.cls = keep(_f32(sd["embeddings.class_embedding"].reshape(D), dev))
        pos = sd["embeddings.position_embedding.weight"]
        if pos.shape[0] != self.G * self.G + 1:
            raise _lib.VitronHipError(f"PackedVit: position table has {pos.shape[0]} rows, need {self.G * self.G + 1}")
        self.pos = keep(_f32(pos, dev))
        self.pre_g = keep(_f32(sd["pre_layrnorm.weight"], dev))
        self.pre_b = keep(_f32(sd["pre_layrnorm.bias"], dev))

        def fuse_qkv(prefix):
            w = torch.cat([sd[prefix + "q_proj.weight"].float() * scale, sd[prefix + "k_proj.weight"].float(),
                           sd[prefix + "v_proj.weight"].float()], 0)
            b = torch.cat([sd[prefix + "q_proj.bias"].float() * scale, sd[prefix + "k_proj.bias"].float(),
                           sd[prefix + "v_proj.bias"].float()], 0)
            return keep(_bf(w, dev)), keep(_f32(b, dev))

        self.layers = (_lib.VtVitLayer * max(self.run_layers, 1))()
        self._mlp_w: List[tuple] = []       # (fc1, fc2) per layer: precise level 1 makes their MX-FP4 images from these
        for l in range(self.run_layers):
            p = f"encoder.layers.{l}."
            L = self.layers[l]
            if self.time_attn:
                L.t_ln_g = keep(_f32(sd[p + "temporal_layer_norm1.weight"], dev)).data_ptr()
                L.t_ln_b = keep(_f32(sd[p + "temporal_layer_norm1.bias"], dev)).data_ptr()
                if self.num_frames != 1:
                    L.t_embed = keep(_f32(sd[p + "temporal_embedding"].reshape(-1, D)[: self.num_frames], dev)).data_ptr()
                w, b = fuse_qkv(p + "temporal_attn.")
                L.t_wqkv, L.t_bqkv = w.data_ptr(), b.data_ptr()
                L.t_wo = keep(_bf(sd[p + "temporal_attn.out_proj.weight"], dev)).data_ptr()
                L.t_bo = keep(_f32(sd[p + "temporal_attn.out_proj.bias"], dev)).data_ptr()
                if p + "temporal_mlp.fc1.weight" in sd:
                    # the IMAGE tower's add_time_attn variant (reference image/modeling_image.py:83-84,129-134): a temporal MLP behind the
                    # temporal attention; the video tower's layers (video/modeling_video.py) have none
                    L.t_ln2_g = keep(_f32(sd[p + "temporal_layer_norm2.weight"], dev)).data_ptr()
                    L.t_ln2_b = keep(_f32(sd[p + "temporal_layer_norm2.bias"], dev)).data_ptr()
                    L.t_w1 = keep(_bf(sd[p + "temporal_mlp.fc1.weight"], dev)).data_ptr()
                    L.t_b1 = keep(_f32(sd[p + "temporal_mlp.fc1.bias"], dev)).data_ptr()
                    L.t_w2 = keep(_bf(sd[p + "temporal_mlp.fc2.weight"], dev)).data_ptr()
                    L.t_b2 = keep(_f32(sd[p + "temporal_mlp.fc2.bias"], dev)).data_ptr()
            L.ln1_g = keep(_f32(sd[p + "layer_norm1.weight"], dev)).data_ptr()
            L.ln1_b = keep(_f32(sd[p + "layer_norm1.bias"], dev)).data_ptr()
            w, b = fuse_qkv(p + "self_attn.")
            L.wqkv, L.bqkv = w.data_ptr(), b.data_ptr()
            L.wo = keep(_bf(sd[p + "self_attn.out_proj.weight"], dev)).data_ptr()
            L.bo = keep(_f32(sd[p + "self_attn.out_proj.bias"], dev)).data_ptr()
            L.ln2_g = keep(_f32(sd[p + "layer_norm2.weight"], dev)).data_ptr()
            L.ln2_b = keep(_f32(sd[p + "layer_norm2.bias"], dev)).data_ptr()
            self._mlp_w.append((keep(_bf(sd[p + "mlp.fc1.weight"], dev)), keep(_bf(sd[p + "mlp.fc2.weight"], dev))))
            L.w1, L.w2 = self._mlp_w[-1][0].data_ptr(), self._mlp_w[-1][1].data_ptr()
            L.b1 = keep(_f32(sd[p + "mlp.fc1.bias"], dev)).data_ptr()
            L.b2 = keep(_f32(sd[p + "mlp.fc2.bias"], dev)).data_ptr()
        act = cfg.get("hidden_act", "quick_gelu")
        if act not in ("gelu", "quick_gelu"):
            raise _lib.VitronHipError(f"PackedVit: unsupported hidden_act {act}")
        m = _lib.VtVitModel()
        m.image_size, m.patch, m.hidden, m.heads, m.intermediate = self.image_size, P, D, heads, self.I
        m.num_layers, m.num_frames, m.add_time_attn = self.run_layers, self.num_frames, int(self.time_attn)
        m.act = _lib.ACT_GELU if act == "gelu" else _lib.ACT_QUICK_GELU
        m.ln_eps = float(cfg.get("layer_norm_eps", 1e-5))
        m.k_pad = self.k_pad
        m.w_patch, m.cls, m.pos = self.w_patch.data_ptr(), self.cls.data_ptr(), self.pos.data_ptr()
        m.pre_ln_g, m.pre_ln_b = self.pre_g.data_ptr(), self.pre_b.data_ptr()
        m.layers = C.cast(self.layers, C.POINTER(_lib.VtVitLayer))
        m.precise = int(cfg.get("precise", 0))
        self.model = m
        self.ws = Workspace(dev)

    def set_precise(self, level: int) -> None:
        """The MODEL's precise level -> the tower's (vt_vit_model.precise; include/vitron_hip.h). Model level 2: every GEMM A operand of the
        tower (norm outputs, q / k through the scores, attention and activation outputs) and the output features as operand pairs, the
        temporal attention in fp32 (tower level 2). Model level 3: the MLPs' operands with their rounding remainders (MX-FP4 images in the
        same launch at ViT-L width, 16-bit pairs otherwise) and the output features as a pair, the attention paths standard (tower
        level 1: the part of level 2 that carries the tower's distance from fp32). 0 / 1: standard."""
        level = int(level)
        self.model.precise = 2 if level == 2 else 1 if level >= 3 else 0
        if self.model.precise == 1 and self.D % 256 == 0 and self.I % 128 == 0 and not getattr(self, "_mx4_done", False):
            # tower level 1 on the MX pipe: fc1 / fc2's 4-bit images (vt_mx4_quant_weights), once; other widths keep 16-bit operand pairs
            from . import ops
            for l in range(self.run_layers):
                (w14, w1e), (w24, w2e) = ops.mx4_quant_weights(self._mlp_w[l][0]), ops.mx4_quant_weights(self._mlp_w[l][1])
                self._keep += [w14, w1e, w24, w2e]
                L = self.layers[l]
                L.w14, L.w1_e, L.w24, L.w2_e = w14.data_ptr(), w1e.data_ptr(), w24.data_ptr(), w2e.data_ptr()
            self._mx4_done = True

    def forward(self, pixels: torch.Tensor, return_hidden: bool = False):
        """pixels [B,3,H,W] or [B,3,T,H,W] (operand dtype or fp32, on device) -> patch features [B(,T),G*G,D] in the operand
        dtype (= feature_select of hidden_states[select_layer]); optionally also the full fp32 hidden state."""
        lib = self.lib
        if not pixels.is_cuda:
            raise _lib.VitronHipError("PackedVit.forward: pixels must live on the GPU")
        video = pixels.dim() == 5
        if video:
            B, Cc, T, H, W = pixels.shape
        else:
            B, Cc, H, W = pixels.shape
            T = 1
        if Cc != 3 or H != self.image_size or W != self.image_size:
            raise _lib.VitronHipError(f"PackedVit.forward: expected 3x{self.image_size}x{self.image_size} pixels, got {tuple(pixels.shape)}")
        if pixels.dtype not in (self.dtype, torch.float32):
            pixels = pixels.to(self.dtype)
        pixels = pixels.contiguous()
        dt = _lib.DTYPE_BF16 if pixels.dtype == self.dtype else _lib.DTYPE_F32
        G2 = self.G * self.G
        out = torch.empty((B * T * G2, self.D), device=self.device, dtype=self.dtype)
        hidden = torch.empty((B * T * (G2 + 1), self.D), device=self.device, dtype=torch.float32) if return_hidden else None
        out_lo = torch.empty_like(out) if self.model.precise >= 1 else None      # precise levels: the features leave as an operand pair
        call_model = _lib.VtVitModel.from_buffer_copy(self.model)     # per-call pointer in a copy: the shared struct stays untouched
        call_model.out_feats_lo = out_lo.data_ptr() if out_lo is not None else None
        nbytes = lib.vt_vit_workspace_bytes(C.byref(call_model), B, T)
        ws = self.ws.get(nbytes)
        _lib.check(lib.vt_vit_forward(C.byref(call_model), pixels.data_ptr(), dt, B, T, int(video), out.data_ptr(),
                                      None if hidden is None else hidden.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
                   "vt_vit_forward", lib)
        feats = out.view(B, T, G2, self.D) if video else out.view(B, G2, self.D)
        if out_lo is not None:
            with_lo(feats, out_lo.view(feats.shape))
        if return_hidden:
            return feats, hidden.view(B * T, G2 + 1, self.D)
        return feats


# =====================================================================================================================
class PackedProjector:
    """mm_projector ('linear', 'mlp2x_gelu', 'mlpNx_gelu'), reference multimodal_projector/builder.py:33-51. One and two layers
    are ONE C call (vt_projector_forward); deeper stacks (N > 2: Linear-GELU-...-Linear) chain the same GEMM with the GELU
    epilogue, one launch per layer."""

    def __init__(self, sd: SD, device, dtype=None, depth: Optional[int] = None):
        self.device = torch.device(device)
        self.dtype = dt = _dtype(dtype)
        self.lib = _lib.lib_for(dt)

        def _bf(t, dev):
            return _op(t, dev, dt)
        if "0.weight" in sd:
            n = 0
            while f"{2 * n}.weight" in sd:
                n += 1
            self.layers = [(_bf(sd[f"{2 * i}.weight"], device), _f32(sd[f"{2 * i}.bias"], device)) for i in range(n)]
        else:
            self.layers = [(_bf(sd["weight"], device), _f32(sd["bias"], device))]
        if depth is not None and len(self.layers) != depth:
            # a state dict with a hole (e.g. '0.*' and '4.*' but no '2.*') would otherwise run silently as a shallower MLP
            raise _lib.VitronHipError(f"PackedProjector: found {len(self.layers)} Linear layers, the projector type declares {depth}")
        for (w0, _), (w1, _) in zip(self.layers, self.layers[1:]):
            if w1.shape[1] != w0.shape[0]:
                raise _lib.VitronHipError(f"PackedProjector: layer widths do not chain ({tuple(w0.shape)} -> {tuple(w1.shape)})")
        self.w1, self.b1 = self.layers[0]
        self.w2, self.b2 = self.layers[1] if len(self.layers) > 1 else (None, None)
        self.din, self.dh = self.w1.shape[1], self.w1.shape[0]
        self.dout = self.layers[-1][0].shape[0]
        self.ws = Workspace(device)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lib = self.lib
        shp = x.shape
        x_lo = pair_lo(x)
        x2 = x.reshape(-1, shp[-1]).to(self.dtype).contiguous()
        M = x2.shape[0]
        if x_lo is not None and len(self.layers) == 2:
            # precise level 2: features arrive as an operand pair -> both Linear layers on pairs, the embeddings leave as a pair
            lo2 = x_lo.reshape(-1, shp[-1]).to(self.dtype).contiguous()
            out = torch.empty((M, self.dout), device=self.device, dtype=self.dtype)
            out_lo = torch.empty_like(out)
            ws = self.ws.get(lib.vt_projector_precise_workspace_bytes(M, self.dh, self.dout))
            _lib.check(lib.vt_projector_forward_precise(x2.data_ptr(), lo2.data_ptr(), M, self.din, self.w1.data_ptr(), self.b1.data_ptr(), self.dh,
                                                        self.w2.data_ptr(), self.b2.data_ptr(), self.dout, out.data_ptr(), out_lo.data_ptr(),
                                                        ws.data_ptr(), ws.numel(), _stream()), "vt_projector_forward_precise", lib)
            return with_lo(out.view(*shp[:-1], self.dout), out_lo.view(*shp[:-1], self.dout))
        if len(self.layers) > 2:
            from . import ops
            h = x2
            for i, (w, b) in enumerate(self.layers):
                h = ops.gemm(h, w, b, ops.EPI_BF16 if i + 1 == len(self.layers) else ops.EPI_BF16_GELU)
            return h.view(*shp[:-1], self.dout)
        out = torch.empty((M, self.dout), device=self.device, dtype=self.dtype)
        ws = self.ws.get(lib.vt_projector_workspace_bytes(M, self.dh))
        _lib.check(lib.vt_projector_forward(x2.data_ptr(), M, self.din, self.w1.data_ptr(), self.b1.data_ptr(), self.dh,
                                            None if self.w2 is None else self.w2.data_ptr(),
                                            None if self.b2 is None else self.b2.data_ptr(), self.dout, out.data_ptr(),
                                            ws.data_ptr(), ws.numel(), _stream()), "vt_projector_forward", lib)
        return out.view(*shp[:-1], self.dout)


# =====================================================================================================================
def resolve_region_slices(regions: Sequence[Sequence[float]], image_size: int) -> List[List[int]]:
    """Host half of transform_bbox_2_mask (reference region_extractor/layer.py:77-85): the mask assignment
    mask[int(x1):int(x2), int(y1):int(y2)] = 1 resolved with Python's own int() and slice semantics, so the
    integer bounds handed to the kernel are exactly the reference's (x -> rows, y -> columns)."""
    out = []
    for x1, y1, x2, y2 in regions:
        r0, r1, _ = slice(int(x1), int(x2)).indices(image_size)
        c0, c1, _ = slice(int(y1), int(y2)).indices(image_size)
        out.append([r0, max(r1, r0), c0, max(c1, c0)])
    return out


class PackedRegion:
    """RegionExtractor (reference region_extractor/layer.py:58-130) in kernel layout."""

    def __init__(self, sd: SD, device, image_size: int = 224, patch_size: int = 14, dtype=None):
        self.device = torch.device(device)
        self.dtype = dt = _dtype(dtype)
        self.lib = _lib.lib_for(dt)

        def _bf(t, dev):
            return _op(t, dev, dt)
        self.image_size, self.patch_size = image_size, patch_size
        self._keep = []
        w = _lib.VtRegionWeights()
        self.in_dim = sd["region_linear.layers.0.weight"].shape[1]
        self.out_dim = sd["region_linear.layers.2.weight"].shape[0]
        w.in_dim, w.out_dim = self.in_dim, self.out_dim
        for i in range(2):
            wt, bt = _bf(sd[f"region_linear.layers.{i}.weight"], device), _f32(sd[f"region_linear.layers.{i}.bias"], device)
            self._keep += [wt, bt]
            w.mlp_w[i], w.mlp_b[i] = wt.data_ptr(), bt.data_ptr()
        l0 = torch.zeros((self.out_dim // 2, 8), dtype=dt, device=device)  # K padded 4 -> 8
        l0[:, :4] = _bf(sd["loc_encoder.loc_encoder.0.weight"], device)
        l0b = _f32(sd["loc_encoder.loc_encoder.0.bias"], device)
        # region_linear's last layer and LocationEncoder's last layer are added (layer.py:129): one GEMM over [h2 | loc hidden]
        fw = torch.cat([_bf(sd["region_linear.layers.2.weight"], device), _bf(sd["loc_encoder.loc_encoder.2.weight"], device)], 1).contiguous()
        fb = (_f32(sd["region_linear.layers.2.bias"], device) + _f32(sd["loc_encoder.loc_encoder.2.bias"], device)).contiguous()
        self._keep += [l0, l0b, fw, fb]
        w.loc_w0, w.loc_b0, w.final_w, w.final_b = l0.data_ptr(), l0b.data_ptr(), fw.data_ptr(), fb.data_ptr()
        self.weights = w
        self.ws = Workspace(device)

    def forward(self, feats: torch.Tensor, regions: Sequence[Sequence[float]], return_mask: bool = False):
        """feats [B,G*G,C] (pre-projector patch features, operand dtype), regions: B boxes -> [B,1,H]."""
        lib = self.lib
        B, n, c = feats.shape
        G = int(math.sqrt(n))
        if len(regions) != B:  # the reference prints and carries on (layer.py:101-107); there is nothing sane to compute
            raise _lib.VitronHipError(f"region_extractor: {B} feature maps but {len(regions)} regions")
        feats = feats.to(self.dtype).contiguous()
        out = torch.empty((B, self.out_dim), device=self.device, dtype=self.dtype)
        mask = torch.empty((B, n), device=self.device, dtype=torch.int32) if return_mask else None
        count = torch.empty((B,), device=self.device, dtype=torch.int32) if return_mask else None
        for s in range(0, B, 16):
            e = min(B, s + 16)
            sl = torch.tensor(resolve_region_slices(regions[s:e], self.image_size), dtype=torch.int32, device=self.device)
            # LocationEncoder input in fp32 [B][4]: the reference builds torch.tensor(regions, dtype=model dtype) (layer.py:126);
            # bf16 cannot hold integers above 256, so the K = 4 layer takes the coordinates unrounded (include/vitron_hip.h)
            co = torch.tensor([list(map(float, r)) for r in regions[s:e]], dtype=torch.float32).reshape(e - s, 4).to(self.device)
            ws = self.ws.get(lib.vt_region_workspace_bytes(e - s, self.in_dim, self.out_dim))
            _lib.check(lib.vt_region_forward(C.byref(self.weights), feats[s:e].data_ptr(), sl.data_ptr(), co.data_ptr(),
                                             e - s, G, self.image_size, out[s:e].data_ptr(),
                                             None if mask is None else mask[s:e].data_ptr(),
                                             None if count is None else count[s:e].data_ptr(), ws.data_ptr(), ws.numel(),
                                             _stream()), "vt_region_forward", lib)
        if return_mask:
            return out.unsqueeze(1), mask, count
        return out.unsqueeze(1)


# =====================================================================================================================
def rope_tables(head_dim: int, max_pos: int, theta: float, device):
    """cos/sin [max_pos, head_dim/2] fp32, computed exactly as LlamaRotaryEmbedding does (fp32 outer product)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    t = torch.arange(max_pos, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq)
    return freqs.cos().to(device).contiguous(), freqs.sin().to(device).contiguous()


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """[I,H] gate, [I,H] up -> [2I,H] with rows gate[0:16], up[0:16], gate[16:32], up[16:32], ..."""
    I, H = gate.shape
    return torch.stack([gate.reshape(I // 16, 16, H), up.reshape(I // 16, 16, H)], dim=1).reshape(2 * I, H)


class PackedLlama:
    """LLaMA decoder weights in kernel layout (vt_llama_model) + embedding table."""

    def __init__(self, sd: SD, cfg: dict, device, rope_len: Optional[int] = None, dtype=None):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        self.dtype = dt = _dtype(dtype)
        self.lib = _lib.lib_for(dt)

        def _bf(t, dev):
            return _op(t, dev, dt)
        H, heads = cfg["hidden_size"], cfg["num_attention_heads"]
        self.H, self.heads, self.hd = H, heads, H // heads
        self.I, self.L = cfg["intermediate_size"], cfg["num_hidden_layers"]
        # the vocabulary is what the TENSORS say (a LoRA adapter's config can name more rows than a 32 000-row base has,
        # reference builder.py:57-61; resize_token_embeddings adds rows after loading, :146): never index past them
        emb_w, head_w = sd["model.embed_tokens.weight"], sd["lm_head.weight"]
        if emb_w.dim() != 2 or head_w.dim() != 2 or emb_w.shape[1] != H or head_w.shape[1] != H:
            raise _lib.VitronHipError(f"PackedLlama: embed_tokens {tuple(emb_w.shape)} / lm_head {tuple(head_w.shape)} do not match hidden_size {H}")
        self.V = int(head_w.shape[0])
        self.embed_rows = int(emb_w.shape[0])
        self.V_pad = (self.V + 3) // 4 * 4     # the logits GEMM wants N % 4 == 0: zero rows behind the vocabulary, never returned
        if self.hd not in (64, 128):
            raise _lib.VitronHipError("PackedLlama: head_dim must be 64 or 128")
        if self.I % 16:
            raise _lib.VitronHipError("PackedLlama: intermediate_size must be a multiple of 16")
        dev = self.device
        self._keep: List[torch.Tensor] = []

        def keep(t):
            self._keep.append(t)
            return t

        self.embed = keep(_bf(sd["model.embed_tokens.weight"], dev))
        self.final_norm = keep(_f32(sd["model.norm.weight"], dev))
        head = _bf(sd["lm_head.weight"], dev)
        if self.V_pad != self.V:
            head = torch.cat([head, torch.zeros((self.V_pad - self.V, H), dtype=dt, device=dev)], 0).contiguous()
        self.lm_head = keep(head)
        self.rope_len = int(rope_len or max(cfg.get("max_position_embeddings", 4096), 8192))
        self.rope_cos, self.rope_sin = rope_tables(self.hd, self.rope_len, float(cfg.get("rope_theta", 10000.0)), dev)
        self.layers = (_lib.VtLlamaLayer * self.L)()
        self.layer_tensors: List[dict] = []
        for l in range(self.L):
            p = f"model.layers.{l}."
            Ly = self.layers[l]
            want = {"self_attn.q_proj": (H, H), "self_attn.k_proj": (H, H), "self_attn.v_proj": (H, H), "self_attn.o_proj": (H, H),
                    "mlp.gate_proj": (self.I, H), "mlp.up_proj": (self.I, H), "mlp.down_proj": (H, self.I),
                    "input_layernorm": (H,), "post_attention_layernorm": (H,)}
            for name, shp in want.items():
                if tuple(sd[p + name + ".weight"].shape) != shp:
                    raise _lib.VitronHipError(f"PackedLlama: {p}{name}.weight is {tuple(sd[p + name + '.weight'].shape)}, config says {shp}")
            t = {"rms1": keep(_f32(sd[p + "input_layernorm.weight"], dev)), "rms2": keep(_f32(sd[p + "post_attention_layernorm.weight"], dev)),
                 "wqkv": keep(torch.cat([_bf(sd[p + "self_attn.q_proj.weight"], dev), _bf(sd[p + "self_attn.k_proj.weight"], dev),
                                         _bf(sd[p + "self_attn.v_proj.weight"], dev)], 0)),
                 "wo": keep(_bf(sd[p + "self_attn.o_proj.weight"], dev)),
                 "wgu": keep(interleave_gate_up(_bf(sd[p + "mlp.gate_proj.weight"], dev), _bf(sd[p + "mlp.up_proj.weight"], dev)).contiguous()),
                 "wdown": keep(_bf(sd[p + "mlp.down_proj.weight"], dev))}
            self.layer_tensors.append(t)      # the packed tensors by name (engine.padded_batch_fixup runs single rows through the primitive ops)
            Ly.rms1, Ly.rms2 = t["rms1"].data_ptr(), t["rms2"].data_ptr()
            Ly.wqkv, Ly.wo, Ly.wgu, Ly.wdown = t["wqkv"].data_ptr(), t["wo"].data_ptr(), t["wgu"].data_ptr(), t["wdown"].data_ptr()
        m = _lib.VtLlamaModel()
        m.hidden, m.heads, m.head_dim, m.intermediate, m.num_layers, m.vocab = H, heads, self.hd, self.I, self.L, self.V_pad
        m.rms_eps = float(cfg.get("rms_norm_eps", 1e-5))
        m.final_norm, m.lm_head = self.final_norm.data_ptr(), self.lm_head.data_ptr()
        m.rope_cos, m.rope_sin, m.rope_len = self.rope_cos.data_ptr(), self.rope_sin.data_ptr(), self.rope_len
        m.layers = C.cast(self.layers, C.POINTER(_lib.VtLlamaLayer))
        m.prefill_norm_fold = int(bool(cfg.get("prefill_norm_fold", False)))
        m.qkv_fuse = int(bool(cfg.get("qkv_fuse", False)))
        m.precise_qk = max(int(cfg.get("precise", 0) or 0), int(bool(cfg.get("precise_qk", False))))
        self.model = m
        self.ws = Workspace(dev)

    def set_qkv_fuse(self, on: bool) -> None:
        """Prefill: rotary + K / V^T page writes inside the QKV GEMM's epilogue instead of the separate vt_kv_tiles pass
        (vt_llama_model.qkv_fuse; bit-identical, measured slower at the benchmark shape: default off)."""
        self.model.qkv_fuse = int(bool(on))

    def set_precise(self, level: int) -> None:
        """0: standard; 1: precise_qk (below); 2: EVERY GEMM A operand of a prefill as an operand pair -- a verification mode at about
        twice the GEMM work that takes the fp16 build's full-depth logits below 1e-3 of the reference's fp32 (vt_llama_model.precise_qk = 2);
        3: every decoder Linear of a prefill adds the MX-FP4 product of the A operand's rounding remainder in the SAME launch (the weights'
        MX-FP4 images are made here, once: + 1/4 of the 16-bit weight bytes) -- the same 1e-3 at ~1.25x of the standard step."""
        level = int(level)
        if level not in (0, 1, 2, 3):
            raise _lib.VitronHipError(f"precise level must be 0, 1, 2 or 3, got {level}")
        if level and self.hd != 128:
            raise _lib.VitronHipError(f"precise modes need head_dim 128 (got {self.hd})")
        if level == 3:
            if self.H % 512 or self.I % 128:
                raise _lib.VitronHipError(f"precise level 3 needs hidden % 512 == 0 and intermediate % 128 == 0 (got {self.H}, {self.I})")
            self._pack_mx4()
        self.model.precise_qk = level

    def _pack_mx4(self) -> None:
        """The weights' MX-FP4 images (vt_mx4_quant_weights on the packed matrices: gate/up keep their interleaved row order)."""
        if getattr(self, "_mx4_done", False):
            return
        from . import ops
        for l in range(self.L):
            t, Ly = self.layer_tensors[l], self.layers[l]
            for name in ("wqkv", "wo", "wgu", "wdown"):
                w4, we = ops.mx4_quant_weights(t[name])
                t[name + "4"], t[name + "_e"] = w4, we
                setattr(Ly, name + "4", w4.data_ptr())
                setattr(Ly, name + "_e", we.data_ptr())
        self._mx4_done = True

    def set_precise_qk(self, on: bool) -> None:
        """Prefills carry q / k (and the norm output that feeds their projection) as hi + lo operand pairs: the attention scores see
        ~2^-20 of operand rounding instead of 2^-12 (fp16) / 2^-9 (bf16) (vt_llama_model.precise_qk; include/vitron_hip.h). The
        fp16 build's full-depth logits move from 1.3e-3 to below 1e-3 of the reference's fp32 output; costs ~10 % of
        the prefill time. head_dim 128 only; decode steps are unchanged. Default off."""
        if on and self.hd != 128:
            raise _lib.VitronHipError(f"precise_qk needs head_dim 128 (got {self.hd})")
        self.model.precise_qk = int(bool(on))

    def set_prefill_norm_fold(self, on: bool) -> None:
        """RMSNorm folded into the prefill tile GEMMs (vt_llama_model.prefill_norm_fold; measured neutral, default off)."""
        self.model.prefill_norm_fold = int(bool(on))


class PagedKVCache:
    """Paged KV pool: K pages [L][pages][heads][64][hd] in the operand dtype, V^T pages [L][pages][heads][hd][64] fp16; 64 tokens per page.
    A free list hands out pages; sequences own a list of page ids (their block table)."""

    def __init__(self, llama: PackedLlama, num_pages: int):
        self.llama = llama
        self.num_pages = int(num_pages)
        n = llama.L * self.num_pages * llama.heads * PAGE_TOKENS * llama.hd
        self.k = torch.zeros(n, dtype=llama.dtype, device=llama.device)
        self.vt = torch.zeros(n, dtype=torch.float16, device=llama.device)    # V^T pages hold fp16 (include/vitron_hip.h)
        self.free = list(range(self.num_pages - 1, -1, -1))
        c = _lib.VtKvCache()
        c.k, c.vt, c.num_pages = self.k.data_ptr(), self.vt.data_ptr(), self.num_pages
        self.struct = c

    def alloc(self, n: int) -> List[int]:
        if n > len(self.free):
            raise _lib.VitronHipError(f"PagedKVCache: out of pages (want {n}, have {len(self.free)})")
        return [self.free.pop() for _ in range(n)]

    def release(self, pages: Sequence[int]) -> None:
        self.free.extend(pages)


class SequenceState:
    """Host-side bookkeeping of one sequence in the paged cache."""

    def __init__(self):
        self.pages: List[int] = []
        self.length = 0  # tokens already in the cache


def llama_forward(llama: PackedLlama, kv: PagedKVCache, seqs: Sequence[SequenceState], embeds: torch.Tensor,
                  q_lens: Sequence[int], positions: Optional[torch.Tensor] = None, logit_rows: Optional[Sequence[int]] = None,
                  return_hidden: bool = False, return_all_hidden: bool = False, embeds_lo: Optional[torch.Tensor] = None):
    """One decoder pass over packed rows. embeds (operand dtype) [sum(q_lens), H]: the new tokens of every sequence, sequence by
    sequence. Appends their K/V to the cache, returns fp32 logits for `logit_rows` (default: last row of each
    sequence). positions default to cache position (length + i). Prefill and decode are the same call.
    return_hidden: also the fp32 residual stream behind the last layer [rows, H]; return_all_hidden: also the stream in front of
    every layer, fp32 [L, rows, H] (vt_llama_model.hidden_trace) -- returns (logits, hidden, trace).
    embeds_lo (precise level 2): the low half of `embeds` when the caller carries them as an operand pair (same shape and dtype)."""
    lib = llama.lib
    dev = llama.device
    rows = int(sum(q_lens))
    if embeds.shape[0] != rows or embeds.shape[1] != llama.H:
        raise _lib.VitronHipError(f"llama_forward: embeds {tuple(embeds.shape)} vs rows={rows}, H={llama.H}")
    if embeds_lo is None:
        embeds_lo = pair_lo(embeds)
    embeds = embeds.to(llama.dtype).contiguous()
    if embeds_lo is not None:
        if tuple(embeds_lo.shape) != tuple(embeds.shape):
            raise _lib.VitronHipError(f"llama_forward: embeds_lo {tuple(embeds_lo.shape)} vs embeds {tuple(embeds.shape)}")
        embeds_lo = embeds_lo.to(llama.dtype).contiguous()
    desc, table, pos_host = [], [], []
    row0 = 0
    max_new_tiles = 1
    for s, q in zip(seqs, q_lens):
        kv_len = s.length + q
        need = (kv_len + PAGE_TOKENS - 1) // PAGE_TOKENS
        if need > len(s.pages):
            s.pages += kv.alloc(need - len(s.pages))
        desc.append([row0, q, kv_len, len(table)])
        table += s.pages[:need]
        pos_host.append((s.length, kv_len))
        max_new_tiles = max(max_new_tiles, (kv_len - 1) // PAGE_TOKENS - s.length // PAGE_TOKENS + 1)
        row0 += q
    max_pos = max(e for _, e in pos_host) - 1
    if max_pos >= llama.rope_len and positions is None:
        raise _lib.VitronHipError(f"llama_forward: position {max_pos} beyond rope table ({llama.rope_len})")
    desc_t = torch.tensor(desc, dtype=torch.int32, device=dev)
    table_t = torch.tensor(table, dtype=torch.int32, device=dev)
    if positions is None:
        import numpy as np
        pos_t = torch.from_numpy(np.concatenate([np.arange(a, e, dtype=np.int32) for a, e in pos_host])).to(dev, non_blocking=True)
    else:
        pos_t = positions.to(device=dev, dtype=torch.int32).contiguous()
    if logit_rows is None:
        logit_rows = [d[0] + d[1] - 1 for d in desc]
    n_logit = len(logit_rows)
    lr_t = torch.tensor(list(logit_rows), dtype=torch.int32, device=dev) if n_logit else None
    logits = torch.empty((n_logit, llama.V_pad), dtype=torch.float32, device=dev) if n_logit else None
    hidden = torch.empty((rows, llama.H), dtype=torch.float32, device=dev) if (return_hidden or return_all_hidden) else None
    trace = torch.empty((llama.L, rows, llama.H), dtype=torch.float32, device=dev) if return_all_hidden else None
    max_kv = max(d[2] for d in desc)
    ws = llama.ws.get(lib.vt_llama_workspace_bytes(C.byref(llama.model), rows, n_logit, len(desc), max_kv))
    # the per-call pointers travel in a COPY of the model struct (two threads / streams sharing one PackedLlama must not see each
    # other's trace or lo buffer; the C call reads the struct and keeps nothing)
    call_model = _lib.VtLlamaModel.from_buffer_copy(llama.model)
    call_model.hidden_trace = trace.data_ptr() if trace is not None else None
    call_model.embeds_lo = embeds_lo.data_ptr() if embeds_lo is not None else None
    _llama_call(lib, call_model, kv, embeds, rows, pos_t, desc_t, len(desc), int(max(q_lens)), int(max_new_tiles), int(max_kv), table_t, lr_t,
                n_logit, logits, hidden, ws)
    for s, q in zip(seqs, q_lens):
        s.length += q
    if logits is not None and llama.V_pad != llama.V:
        logits = logits[:, :llama.V]               # rows keep the padded stride; the kernels take (V, row stride)
    if return_all_hidden:
        return logits, hidden, trace
    if return_hidden:
        return logits, hidden
    return logits


def _llama_call(lib, model_struct, kv, embeds, rows, pos_t, desc_t, nseq, max_q, max_new_tiles, max_kv, table_t, lr_t, n_logit, logits, hidden, ws):
    _lib.check(lib.vt_llama_forward(C.byref(model_struct), C.byref(kv.struct), embeds.data_ptr(), rows, pos_t.data_ptr(),
                                    desc_t.data_ptr(), nseq, max_q, max_new_tiles, max_kv, table_t.data_ptr(),
                                    None if lr_t is None else lr_t.data_ptr(), n_logit,
                                    None if logits is None else logits.data_ptr(),
                                    None if hidden is None else hidden.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
               "vt_llama_forward", lib)


def padded_batch_fixup(llama: PackedLlama, kv: PagedKVCache, seqs: Sequence[SequenceState], padded_len: int,
                       ids_valid: Sequence[int], ids_len: int) -> dict:
    """Turn the packed prefill of a RIGHT-padded batch into the state the REFERENCE's padded-batch generate() decodes from
    (`generate(..., padded_batch=True)`; reference llava_arch.py:196-205 + transformers 4.31 GenerationMixin, restated step by step in
    tests/golden/make_golden._padded_batch_greedy_hf431 and pinned by tests/golden/greedy_batch.npz):

      * the reference prefills the spliced batch padded to `padded_len` rows: a pad row (zero embedding, position 0) attends the sample's
        VALID keys only -- not itself, not the other pads -- and its K / V land in the cache like any row's. All pad rows of a sample are
        identical, so ONE row per sample is run here through the primitive operators (RMSNorm, weight-streaming GEMMs, single-query
        attention over the cache without append), layer by layer, and its K / V are written to the cache rows [valid, padded_len);
      * the first token of a sample shorter than the longest is read from its LAST PAD ROW (`logits[:, -1]`) -> returned as `pad_logits`;
      * every decode step attends the whole cache EXCEPT the rows [ids_valid[b], ids_len) -- the zeros of the ids-length mask, which the
        fix-up aligns with whatever spliced rows share their index -- at position (rows attended) - 1. That hole never changes, so the
        cache is COMPACTED once (rows before and behind the hole gathered into fresh pages: K rows carry their rotary position with
        them) and the ordinary decode kernels run on it: position = compacted length + step, exactly the reference's.

    seqs: the sequences after the packed prefill (length = valid spliced rows). Returns {"pad_logits": {b: fp32 [V]}, "holes": {b: n}}."""
    from . import ops
    dev, H, heads, hd, dt = llama.device, llama.H, llama.heads, llama.hd, llama.dtype
    B = len(seqs)
    valid = [s.length for s in seqs]
    pads = [b for b in range(B) if valid[b] < padded_len]
    kview = kv.k.view(llama.L, kv.num_pages, heads, PAGE_TOKENS, hd)
    vview = kv.vt.view(llama.L, kv.num_pages, heads, hd, PAGE_TOKENS)
    pad_logits = {}
    if pads:
        eps = float(llama.model.rms_eps)
        for b in pads:                                                    # pages for the pad rows
            need = (padded_len + PAGE_TOKENS - 1) // PAGE_TOKENS
            if need > len(seqs[b].pages):
                seqs[b].pages += kv.alloc(need - len(seqs[b].pages))
        table, desc_q, desc_w, rep = [], [], [], []
        row0 = 0
        for i, b in enumerate(pads):
            npad = padded_len - valid[b]
            desc_q.append([i, 1, valid[b], len(table)])                   # the pad row's attention: the VALID keys only
            desc_w.append([row0, npad, padded_len, len(table)])           # its K / V as "new tokens" [valid, padded_len)
            table += seqs[b].pages
            rep += [i] * npad
            row0 += npad
        table_t = torch.tensor(table, dtype=torch.int32, device=dev)
        desc_q_t = torch.tensor(desc_q, dtype=torch.int32, device=dev)
        desc_w_t = torch.tensor(desc_w, dtype=torch.int32, device=dev)
        rep_t = torch.tensor(rep, dtype=torch.long, device=dev)
        max_kv = max(valid[b] for b in pads)
        max_new_tiles = max((padded_len - 1) // PAGE_TOKENS - valid[b] // PAGE_TOKENS + 1 for b in pads)
        x = torch.zeros((len(pads), H), dtype=torch.float32, device=dev)  # the zero embedding of a pad row (llava_arch.py:520-533)
        lstride = kv.num_pages * heads * PAGE_TOKENS * hd
        for l, t in enumerate(llama.layer_tensors):
            k_l, vt_l = kv.k[l * lstride:(l + 1) * lstride], kv.vt[l * lstride:(l + 1) * lstride]
            y = ops.rmsnorm(x, t["rms1"], eps, dtype=dt)
            qkv = ops.gemm(y, t["wqkv"], None, ops.EPI_BF16)             # position 0: the rotary embedding is the identity
            att = ops.attn_decode(qkv[:, :H], k_l, vt_l, table_t, desc_q_t, heads, hd, 1.0 / math.sqrt(hd), max_kv)
            ops.kv_tiles(qkv.index_select(0, rep_t).contiguous(), 0, H, 2 * H, k_l, vt_l, table_t, desc_w_t, max_new_tiles, heads, hd)
            x = ops.gemm(att, t["wo"], None, ops.EPI_F32_RESID, out=x)
            h = ops.gemm(ops.rmsnorm(x, t["rms2"], eps, dtype=dt), t["wgu"], None, ops.EPI_SWIGLU_BF16)
            x = ops.gemm(h, t["wdown"], None, ops.EPI_F32_RESID, out=x)
        lg = ops.gemm(ops.rmsnorm(x, llama.final_norm, eps, dtype=dt), llama.lm_head, None, ops.EPI_F32)[:, :llama.V]
        for i, b in enumerate(pads):
            pad_logits[b] = lg[i]
            seqs[b].length = padded_len
    holes = {}
    for b in range(B):
        lo, hi = int(ids_valid[b]), min(int(ids_len), seqs[b].length)
        if hi <= lo:
            continue
        keep = list(range(0, lo)) + list(range(hi, seqs[b].length))
        old = seqs[b].pages
        new = kv.alloc((len(keep) + PAGE_TOKENS - 1) // PAGE_TOKENS + 1)
        try:     #  a failure while the rows move must hand the fresh pages back: generate() only releases seqs[*].pages
            kview[:, new] = 0
            vview[:, new] = 0
            src = torch.tensor(keep, dtype=torch.long, device=dev)
            dst = torch.arange(len(keep), dtype=torch.long, device=dev)
            oldp, newp = torch.tensor(old, dtype=torch.long, device=dev), torch.tensor(new, dtype=torch.long, device=dev)
            sp, ss, dp, ds = oldp[src // PAGE_TOKENS], src % PAGE_TOKENS, newp[dst // PAGE_TOKENS], dst % PAGE_TOKENS
            for l in range(kview.shape[0]):                                 # layer by layer: the gather's temporary stays one layer's rows
                kview[l, dp, :, ds, :] = kview[l, sp, :, ss, :]             # (row moves, no arithmetic: K rows carry their rotation)
                vview[l, dp, :, :, ds] = vview[l, sp, :, :, ss]
        except Exception:
            kv.release(new)
            raise
        kv.release(old)
        seqs[b].pages, seqs[b].length = new, len(keep)
        holes[b] = hi - lo
    return {"pad_logits": pad_logits, "holes": holes}


class DecodeState:
    """Device-resident state of a run of single-token decode steps over a fixed set of sequences.

    `llama_forward` rebuilds the step metadata (sequence descriptors, page table, positions) on the host and uploads it for
    every call, and the sampling loop around it needs each token on the host before it can enqueue the next pass: the GPU
    idles for that round trip once per token. Here the metadata lives on the device for the whole run -- pages for
    `max_steps` more tokens are reserved up front, one `vt_decode_feed` launch per token resolves pad / EOS, writes the next
    input rows and advances kv_len / positions -- so `forward()` needs nothing from the host and the caller can enqueue
    pass t+1 before it has read token t (read-back through pinned memory and an event, double-buffered).
    The caller owns the speculation: `rollback()` forgets the last pass when the token before it turned out to end the run."""

    def __init__(self, llama: PackedLlama, kv: PagedKVCache, seqs: Sequence[SequenceState], max_steps: int,
                 eos_ids: Sequence[int] = (), pad_id: int = 0):
        from . import ops
        self._ops = ops
        self.llama, self.kv, self.seqs = llama, kv, list(seqs)
        dev = llama.device
        B = self.B = len(self.seqs)
        desc, table = [], []
        for i, s in enumerate(self.seqs):
            if s.length <= 0:
                raise _lib.VitronHipError("DecodeState: every sequence needs a prefilled context")
            need = (s.length + max_steps + PAGE_TOKENS - 1) // PAGE_TOKENS
            if need > len(s.pages):
                s.pages += kv.alloc(need - len(s.pages))
            desc.append([i, 1, s.length, len(table)])        # kv_len / position are advanced by feed() BEFORE each pass
            table += s.pages
        self.max_len = max(s.length for s in self.seqs) + max_steps
        self.desc = torch.tensor(desc, dtype=torch.int32, device=dev)
        self.table = torch.tensor(table, dtype=torch.int32, device=dev)
        self.pos = torch.tensor([s.length - 1 for s in self.seqs], dtype=torch.int32, device=dev)
        self.rows = torch.arange(B, dtype=torch.int32, device=dev)
        self.eos = torch.tensor(sorted(int(e) for e in eos_ids), dtype=torch.int32, device=dev) if len(eos_ids) else None
        self.pad_id = int(pad_id)
        self.finished = torch.zeros(B, dtype=torch.int32, device=dev)
        self.x = torch.empty((B, llama.H), dtype=llama.dtype, device=dev)
        self.tok_dev = [torch.empty((2, B), dtype=torch.int32, device=dev) for _ in range(2)]
        self.tok_pin = [torch.empty((2, B), dtype=torch.int32).pin_memory() for _ in range(2)]
        self.events = [torch.cuda.Event() for _ in range(2)]
        self.fed = 0        # tokens handed to feed()
        self.passes = 0     # decoder passes enqueued

    def feed(self, next_ids: torch.Tensor) -> int:
        """Consume one sampled token id per sequence (device int32 [B]); returns the read-back slot for `read()`."""
        slot = self.fed & 1
        self._ops.decode_feed(self.llama.embed, next_ids.contiguous(), self.finished, self.eos, self.pad_id, self.tok_dev[slot],
                              self.x, self.desc, self.pos)
        self.tok_pin[slot].copy_(self.tok_dev[slot], non_blocking=True)
        self.events[slot].record(torch.cuda.current_stream(self.llama.device))
        self.fed += 1
        return slot

    def read(self, slot: int):
        """(tokens, finished flags) of the feed() that returned `slot`, as host lists; waits for that launch only."""
        self.events[slot].synchronize()
        t = self.tok_pin[slot].tolist()
        return t[0], [bool(f) for f in t[1]]

    def forward(self) -> torch.Tensor:
        """One decoder pass over the rows feed() just wrote; fp32 logits [B, V]. No host -> device traffic."""
        if self.passes >= self.fed:
            raise _lib.VitronHipError("DecodeState.forward: feed() the sampled tokens first")
        llama, B = self.llama, self.B
        lib = llama.lib
        for s in self.seqs:
            if s.length >= llama.rope_len:
                raise _lib.VitronHipError(f"llama_forward: position {s.length} beyond rope table ({llama.rope_len})")
            if s.length + 1 > len(s.pages) * PAGE_TOKENS:
                raise _lib.VitronHipError("DecodeState.forward: more steps than pages were reserved for")
        logits = torch.empty((B, llama.V_pad), dtype=torch.float32, device=llama.device)
        ws = llama.ws.get(lib.vt_llama_workspace_bytes(C.byref(llama.model), B, B, B, self.max_len))
        _lib.check(lib.vt_llama_forward(C.byref(llama.model), C.byref(self.kv.struct), self.x.data_ptr(), B, self.pos.data_ptr(),
                                        self.desc.data_ptr(), B, 1, 1, int(self.max_len), self.table.data_ptr(),
                                        self.rows.data_ptr(), B, logits.data_ptr(), None, ws.data_ptr(), ws.numel(), _stream()),
                   "vt_llama_forward", lib)
        for s in self.seqs:
            s.length += 1
        self.passes += 1
        return logits if llama.V_pad == llama.V else logits[:, :llama.V]

    def rollback(self) -> None:
        """Forget the last pass (its token should not have been fed): the cache slot is simply overwritten later."""
        if self.passes <= 0:
            return
        for s in self.seqs:
            s.length -= 1
        self.passes -= 1
