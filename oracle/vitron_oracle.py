"""CPU oracle for the Vitron multimodal forward pass -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as the
checker (or as the timed CPU baseline). The product path (vitron_amd/) never imports it and has no CPU fallback.

What it is: a plain-PyTorch fp32 restatement of the reference's algorithm for the hot path, one function per
SURVEY.md 8(a) row, each citing the reference file:line it follows. The arithmetic of the CLIP / LLaMA blocks
lives in the third-party dependency transformers==4.31.0 (pinned at /root/reference/pyproject.toml:17,
requirements.txt:63; not vendored), so those parts restate its published algorithm and
are anchored on the reference's call sites.

Pinning: the reference ships no tests or golden vectors for this path, so the oracle is pinned
against outputs of the reference's OWN modules run in the build container under an import shim
(oracle/ref_shim.py): tests/golden/make_golden.py generated tests/golden/*.npz from
  - vitron/model/multimodal_encoder/languagebind/video/modeling_video.py  CLIPVisionTransformer
  - vitron/model/region_extractor/layer.py                               RegionExtractor
  - vitron/model/multimodal_projector/builder.py                         build_vision_projector
  - vitron/model/llava_arch.py + language_model/llava_llama.py           prepare_inputs_labels_for_multimodal, forward
  - vitron/mm_utils.py                                                   tokenizer_image_token, preprocess_region ...
and tests/test_oracle_golden.py checks this file against them (fp32, rel-L2 <= 1e-5; integer outputs exact).

Numeric modes (the `emulate_bf16` argument of every function; the name predates the fp16 build):
  emulate_bf16=False   pure fp32 on the given (bf16-representable) inputs and weights.
  emulate_bf16=True    same maths, but values are rounded to bf16 at exactly the points where the HIP path
                       stores its 16-bit operands (norm outputs, fused-QKV, attention outputs, activations) and the
                       softmax weights to fp16 where the prefill attention kernel feeds them to its P.V MFMA;
                       accumulation, softmax, norms and the residual stream stay fp32 like the kernels.
  emulate_bf16="fp16"  the same storage points rounded to IEEE fp16 (saturating at +-65504): the emulation of
                       libvitron_hip_f16.so, the fp16-operand build (vitron_amd/csrc/vt_common.h).
State dicts use the reference's parameter names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

IGNORE_INDEX = -100        # reference vitron/constants.py:7
IMAGE_TOKEN_INDEX = -200   # reference vitron/constants.py:9
OBJS_TOKEN_INDEX = -300    # reference vitron/constants.py:24

SD = Dict[str, torch.Tensor]


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def fp16_round(x: torch.Tensor) -> torch.Tensor:
    """Softmax weights as the flash kernel feeds them to its P.V MFMA: fp16 (11 mantissa bits), values in (0, 1]."""
    return x.to(torch.float16).to(torch.float32)


def fp16_store(x: torch.Tensor) -> torch.Tensor:
    """An operand store of the fp16 build: round to nearest even, saturate at +-65504 (vt_common.h pack_op2)."""
    return x.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)


def _r(x: torch.Tensor, emulate) -> torch.Tensor:
    if isinstance(emulate, str):
        if emulate != "fp16":
            raise ValueError(f"emulate_bf16 must be False, True or 'fp16', got {emulate!r}")
        return fp16_store(x)
    return bf16_round(x) if emulate else x


def _pair(x: torch.Tensor, emulate) -> torch.Tensor:
    """An operand PAIR of the precise_qk mode (vt_llama_model.precise_qk): hi = store(x), lo = store(x - hi); the kernels multiply both
    halves, so what the arithmetic sees is hi + lo (2 x 11 / 2 x 8 mantissa bits). Identity when nothing is emulated."""
    if not emulate:
        return x
    hi = _r(x, emulate)
    return hi + _r(x - hi, emulate)


def _lin(x, w, b=None):
    return F.linear(x, w.float(), None if b is None else b.float())


# ---- MX-FP4 (OCP e2m1 elements, e8m0 power-of-two block scales): the low halves of precise level 3's operand pairs ------------------
# Restates what the kernels do (vitron_amd/csrc/vt_mx4.hip, vt_gemm8x.hip), not anything in the reference: the reference computes in one
# format throughout (vitron/model/builder.py:47). v_mfma_scale_f32_16x16x128_f8f6f4 multiplies 4-bit elements {0, .5, 1, 1.5, 2, 3, 4, 6}
# x sign and applies 2^(E - 127) per 32-element block of each operand (tools/mx_probe.hip pins layout and scale semantics on the GPU).
MX4_BLOCK = 32


def mx4_exponent(amax: torch.Tensor) -> torch.Tensor:
    """Unbiased power-of-two scale exponent e of a block whose largest magnitude is amax: the smallest e with amax / 2^e <= 6, i.e.
    amax / 2^e in (3, 6] -- nothing is clipped. From the fp32 fields, exactly as the kernels do it: floor(log2 amax) - 2, plus one when
    the mantissa exceeds 1.5. Clamped to [-126, 127] (2^e stays a normal fp32 number); amax = 0 gives -126 (every element quantises to 0)."""
    bits = amax.float().contiguous().view(torch.int32)
    ex = ((bits >> 23) & 0xFF) - 127
    e = ex - 2 + ((bits & 0x7FFFFF) > 0x400000).to(torch.int32)
    return e.clamp(-126, 127)


def mx4_round(x: torch.Tensor) -> torch.Tensor:
    """|x| <= 6 (already divided by the block scale) to the nearest e2m1 value, ties to the even mantissa: rint at a step of .5 below 2,
    1 below 4, 2 above (the e2m1 codes with mantissa bit 0 are 0, 1, 2, 4: rint's ties-to-even lands on exactly those)."""
    a = x.abs()
    q = torch.where(a < 2.0, torch.round(a * 2.0) * 0.5, torch.where(a < 4.0, torch.round(a), torch.round(a * 0.5) * 2.0))
    return torch.copysign(q.clamp(max=6.0), x)


def mx4_codes(q: torch.Tensor) -> torch.Tensor:
    """e2m1 values -> 4-bit codes (bit 3 = sign; -0 keeps its sign bit like the hardware conversion would)."""
    a = q.abs()
    c = torch.where(a <= 2.0, a * 2.0, torch.where(a <= 4.0, a + 2.0, torch.full_like(a, 7.0))).to(torch.int32)
    return c | (torch.signbit(q).to(torch.int32) << 3)


def mx4_quant(x: torch.Tensor, block: Optional[int] = MX4_BLOCK):
    """x [..., K] -> (dequantised fp32 image, e2m1 values, unbiased exponents [..., K / block]). block = None: ONE scale per row (the
    weights' format: `w4_scale` per output feature); block = 32: the activations' low halves (a scale per 32 consecutive k)."""
    x = x.float()
    K = x.shape[-1]
    b = K if block is None else block
    xb = x.reshape(*x.shape[:-1], K // b, b)
    e = mx4_exponent(xb.abs().amax(-1))
    sc = torch.ldexp(torch.ones_like(e, dtype=torch.float32), e).unsqueeze(-1)
    q = mx4_round(xb / sc)
    return (q * sc).reshape(x.shape), q.reshape(x.shape), e


def _lin_mx(x, w, emulate, w4=None):
    """A GEMM of precise level 3: A = hi + lo with hi = store(x) on the 16-bit MFMA against the 16-bit weights and lo = x - hi in MX-FP4
    (block 32) against the weights' MX-FP4 image (one scale per output feature), both accumulated in fp32."""
    hi = _r(x, emulate)
    lo4 = mx4_quant(x - hi, MX4_BLOCK)[0]
    if w4 is None:
        w4 = mx4_quant(w.float(), None)[0]
    return F.linear(hi, w.float()) + F.linear(lo4, w4)


# =====================================================================================================================
# ViT tower (image: add_time_attn=False, T=1; video: add_time_attn=True, T=num_frames)
# =====================================================================================================================
def quick_gelu(x):  # transformers ACT2FN['quick_gelu']
    return x * torch.sigmoid(1.702 * x)


def clip_attention(x, sd: SD, prefix: str, heads: int, emulate: bool, round_p: bool = True, precise: int = 0):
    """transformers-4.31 CLIPAttention.forward: q = q_proj(x)*hd^-0.5, bmm, softmax,
    bmm, out_proj -- no masks on the vision path (reference modeling_video.py:652-657 passes None).
    precise = 2 (with an emulation mode): the storage points of vt_vit_model.precise = 2 -- q and k reach the scores as operand PAIRS, v
    goes from fp32 straight into the V^T tiles' fp16, the output leaves as a pair; the temporal attention (round_p False) runs on the
    fp32 q | k | v and only its output is stored (as a pair)."""
    B, N, D = x.shape
    hd = D // heads
    pairs = bool(emulate) and int(precise) >= 2
    st = (lambda t: _pair(t, emulate)) if pairs else (lambda t: _r(t, emulate))
    qkv_st = (lambda t: t) if (pairs and not round_p) else st
    q = qkv_st(_lin(x, sd[prefix + "q_proj.weight"], sd[prefix + "q_proj.bias"])) * (hd ** -0.5)
    k = qkv_st(_lin(x, sd[prefix + "k_proj.weight"], sd[prefix + "k_proj.bias"]))
    v = _lin(x, sd[prefix + "v_proj.weight"], sd[prefix + "v_proj.bias"])
    if not pairs:
        v = _r(v, emulate)
    q = q.view(B, N, heads, hd).transpose(1, 2)
    k = k.view(B, N, heads, hd).transpose(1, 2)
    v = v.view(B, N, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if emulate and round_p:  # flash kernel: P (relative to the row max) is rounded to fp16 for the PV MFMA, row sum fp32
        p = torch.exp(s - s.amax(-1, keepdim=True))
        o = (fp16_round(p) @ fp16_store(v)) / p.sum(-1, keepdim=True)   # V^T pages hold fp16, saturating at +-65504 (vt_common.h)
    else:
        o = torch.softmax(s, dim=-1) @ v
    o = o.transpose(1, 2).reshape(B, N, D)
    return st(o)  # out_proj is applied by the caller (its output goes straight into the fp32 residual)


def vit_forward(sd: SD, cfg: dict, pixels: torch.Tensor, num_layers: Optional[int] = None, emulate_bf16: bool = False, precise: int = 0):
    """CLIPVisionTransformer.forward + CLIPEncoderLayer.forward of the reference:
      vitron/model/multimodal_encoder/languagebind/video/modeling_video.py:610-675 and :65-158
    pixels: [B,3,H,W] (image) or [B,3,T,H,W] (video). Returns the hidden state after `num_layers` encoder
    layers as [B*T, N, D] fp32 (hidden_states[num_layers] in HF numbering; select_layer=-2 <=> L-1 layers).
    precise = 2 (with an emulation mode): the storage points of the kernels' precise level 2 (vt_vit_model.precise) -- every GEMM A operand
    an operand pair instead of one 16-bit value. precise = 1: the tower's level 1 at ViT-L width -- the spatial MLP's two products as _lin_mx
    (16-bit operand + MX-FP4 image of its rounding remainder), the attention paths (and the image tower's temporal MLP) as in the standard mode."""
    pairs = bool(emulate_bf16) and int(precise) >= 2
    mlp_mx = bool(emulate_bf16) and int(precise) == 1         # tower level 1 (the towers' share of the model's precise level 3)
    _st = (lambda t: _pair(t, emulate_bf16)) if pairs else (lambda t: _r(t, emulate_bf16))
    D, heads, P = cfg["hidden_size"], cfg["num_attention_heads"], cfg["patch_size"]
    L = cfg["num_hidden_layers"] if num_layers is None else num_layers
    eps = cfg.get("layer_norm_eps", 1e-5)
    act = cfg.get("hidden_act", "quick_gelu")
    time_attn = bool(cfg.get("add_time_attn", False))
    if pixels.dim() == 5:  # 'b c t h w -> (b t) c h w'   modeling_video.py:637-640
        B, _, T, _, _ = pixels.shape
        pixels = pixels.permute(0, 2, 1, 3, 4).reshape(B * T, 3, pixels.shape[3], pixels.shape[4])
    else:
        B, T = pixels.shape[0], 1
    pixels = pixels.float()
    # CLIPVisionEmbeddings: conv(k=s=P, no bias) -> flatten(2).transpose(1,2) -> cat CLS -> + position   (Appendix A)
    pe = F.conv2d(pixels, sd["embeddings.patch_embedding.weight"].float(), stride=P)
    pe = pe.flatten(2).transpose(1, 2)
    cls = sd["embeddings.class_embedding"].float().expand(pe.shape[0], 1, -1)
    x = torch.cat([cls, pe], dim=1) + sd["embeddings.position_embedding.weight"].float()[: pe.shape[1] + 1]
    # PatchDropout is the identity at eval (modeling_video.py:31-32); pre_layrnorm :650
    x = F.layer_norm(x, (D,), sd["pre_layrnorm.weight"].float(), sd["pre_layrnorm.bias"].float(), eps)
    N = x.shape[1]
    for l in range(L):
        p = f"encoder.layers.{l}."
        if time_attn:
            t = cfg["num_frames"]
            assert t == T, "video tower built for num_frames frames"
            if t != 1:  # modeling_video.py:110-114 (the add changes the residual stream itself)
                x = (x.view(B, T, N, D) + sd[p + "temporal_embedding"].float()[:, :t, None, :]).view(B * T, N, D)
            res = x  # :117
            h = x.view(B, T, N, D).transpose(1, 2).reshape(B * N, T, D)  # '(b t) n d -> (b n) t d'
            h = _st(F.layer_norm(h, (D,), sd[p + "temporal_layer_norm1.weight"].float(),
                                 sd[p + "temporal_layer_norm1.bias"].float(), eps))
            h = clip_attention(h, sd, p + "temporal_attn.", heads, emulate_bf16, round_p=False, precise=precise)  # temporal kernel keeps P in fp32
            h = _lin(h, sd[p + "temporal_attn.out_proj.weight"], sd[p + "temporal_attn.out_proj.bias"])
            x = res + h.view(B, N, T, D).transpose(1, 2).reshape(B * T, N, D)  # :127
            if p + "temporal_mlp.fc1.weight" in sd:
                # the IMAGE tower's add_time_attn variant only (image/modeling_image.py:83-84,129-134): a row-wise MLP behind the temporal
                # attention (the '(b t) n d -> (b n) t d' rearrangement around it does not change a row-wise operation)
                res = x
                h = _st(F.layer_norm(x, (D,), sd[p + "temporal_layer_norm2.weight"].float(), sd[p + "temporal_layer_norm2.bias"].float(), eps))
                h = _lin(h, sd[p + "temporal_mlp.fc1.weight"], sd[p + "temporal_mlp.fc1.bias"])
                h = _st(F.gelu(h) if act == "gelu" else quick_gelu(h))
                x = res + _lin(h, sd[p + "temporal_mlp.fc2.weight"], sd[p + "temporal_mlp.fc2.bias"])
        res = x  # spatial attention :136-146
        h = _st(F.layer_norm(x, (D,), sd[p + "layer_norm1.weight"].float(), sd[p + "layer_norm1.bias"].float(), eps))
        h = clip_attention(h, sd, p + "self_attn.", heads, emulate_bf16, precise=precise)
        x = res + _lin(h, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        res = x  # MLP :148-151 (CLIPMLP: fc2(act(fc1(x))))
        hn = F.layer_norm(x, (D,), sd[p + "layer_norm2.weight"].float(), sd[p + "layer_norm2.bias"].float(), eps)
        if mlp_mx:      # tower level 1 on the MX pipe: both MLP products add the MX-FP4 product of their A operand's rounding remainder
            h = _lin_mx(hn, sd[p + "mlp.fc1.weight"], emulate_bf16) + sd[p + "mlp.fc1.bias"].float()
            h = F.gelu(h) if act == "gelu" else quick_gelu(h)
            x = res + _lin_mx(h, sd[p + "mlp.fc2.weight"], emulate_bf16) + sd[p + "mlp.fc2.bias"].float()
            continue
        h = _st(hn)
        h = _lin(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        h = F.gelu(h) if act == "gelu" else quick_gelu(h)
        h = _st(h)
        x = res + _lin(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x


def tower_features(sd: SD, cfg: dict, pixels: torch.Tensor, select_layer: int = -2, emulate_bf16: bool = False):
    """LanguageBind{Image,Video}Tower.forward + feature_select (reference languagebind/__init__.py:96-121,182-204):
    hidden_states[select_layer], CLS dropped. Image -> [B, G*G, D]; video -> [B, T, G*G, D]."""
    nl = cfg["num_hidden_layers"]
    layers = select_layer if select_layer >= 0 else nl + 1 + select_layer  # index into the (L+1)-long hidden_states list
    h = vit_forward(sd, cfg, pixels, layers, emulate_bf16)
    h = _r(h[:, 1:], emulate_bf16)
    if pixels.dim() == 5:
        B, T = pixels.shape[0], pixels.shape[2]
        return h.view(B, T, h.shape[1], h.shape[2])
    return h


# =====================================================================================================================
# mm_projector  (reference vitron/model/multimodal_projector/builder.py:33-51 -- 'linear' / 'mlpNx_gelu'; Vitron: 'mlp2x_gelu')
# =====================================================================================================================
def projector_forward(sd: SD, x: torch.Tensor, emulate_bf16: bool = False):
    if "0.weight" not in sd:  # 'linear'
        return _r(_lin(x.float(), sd["weight"], sd["bias"]), emulate_bf16)
    # 'mlpNx_gelu': nn.Sequential(Linear, GELU, Linear[, GELU, Linear ...]) -- Linear layers at the even indices (builder.py:39-46)
    n = 0
    while f"{2 * n}.weight" in sd:
        n += 1
    h = x.float()
    for i in range(n):
        h = _lin(h, sd[f"{2 * i}.weight"], sd[f"{2 * i}.bias"])
        if i + 1 < n:
            h = F.gelu(h)
        h = _r(h, emulate_bf16)
    return h


# =====================================================================================================================
# region_extractor  (reference vitron/model/region_extractor/layer.py)
# =====================================================================================================================
def region_mask(regions: Sequence[Sequence[float]], image_size: int) -> torch.Tensor:
    """transform_bbox_2_mask, layer.py:77-85: mask[int(x1):int(x2), int(y1):int(y2)] = 1 -- x indexes rows."""
    masks = []
    for bbox in regions:
        m = torch.zeros((image_size, image_size), dtype=torch.float32)
        x1, y1, x2, y2 = bbox
        m[int(x1):int(x2), int(y1):int(y2)] = 1
        masks.append(m)
    return torch.stack(masks, 0)


def region_forward(sd: SD, feats: torch.Tensor, regions: Sequence[Sequence[float]], image_size: int = 224,
                   emulate_bf16: bool = False, coords: Optional[torch.Tensor] = None):
    """RegionExtractor.forward, layer.py:87-130. feats [B, G*G, C]. Returns (region_feats [B,1,H], cell_mask
    [B,G*G] int, cell_count [B] int). `coords` overrides the LocationEncoder input (e.g. bf16-rounded coords)."""
    b, n, c = feats.shape
    g = int(math.sqrt(n))
    feats = feats.float()
    mask = region_mask(regions, image_size).unsqueeze(1)                                  # :112-114
    fm = feats.reshape(b, g, g, c).permute(0, 3, 1, 2)                                    # :116
    # MaskPooling.forward :27-43
    if fm.shape[-2:] != mask.shape[-2:]:
        mask = F.interpolate(mask, size=fm.shape[-2:], mode="bilinear", align_corners=False)
    mask = (mask > 0).to(mask.dtype)
    denorm = mask.sum(dim=(-1, -2), keepdim=True) + 1e-8
    pooled = torch.einsum("bchw,bqhw->bqc", fm, mask / denorm)                            # :38-42
    x = _r(pooled.reshape(-1, c), emulate_bf16)                                           # :122
    # MLP (3 layers, ReLU between) :17-20
    x = _r(F.relu(_lin(x, sd["region_linear.layers.0.weight"], sd["region_linear.layers.0.bias"])), emulate_bf16)
    x = _r(F.relu(_lin(x, sd["region_linear.layers.1.weight"], sd["region_linear.layers.1.bias"])), emulate_bf16)
    x = _lin(x, sd["region_linear.layers.2.weight"], sd["region_linear.layers.2.bias"])
    # LocationEncoder on the raw box coordinates :46-56,126
    loc_in = torch.tensor(regions, dtype=torch.float32) if coords is None else coords.float()
    l = _r(F.relu(_lin(loc_in, sd["loc_encoder.loc_encoder.0.weight"], sd["loc_encoder.loc_encoder.0.bias"])), emulate_bf16)
    l = _lin(l, sd["loc_encoder.loc_encoder.2.weight"], sd["loc_encoder.loc_encoder.2.bias"])
    out = _r(x + l, emulate_bf16).unsqueeze(1)                                            # :129-130
    cell_mask = mask.reshape(b, -1).to(torch.int32)
    return out, cell_mask, cell_mask.sum(-1).to(torch.int32)


# =====================================================================================================================
# prompt -> ids helpers  (reference vitron/mm_utils.py)
# =====================================================================================================================
def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, is_first=True):
    """mm_utils.py:80-99."""
    prompt_chunks = [tokenizer(chunk).input_ids for chunk in prompt.split("<image>")]

    def insert_separator(X, sep):
        return [ele for sublist in zip(X, [sep] * len(X)) for ele in sublist][:-1]

    input_ids = []
    offset = 0
    if len(prompt_chunks) > 0 and len(prompt_chunks[0]) > 0 and prompt_chunks[0][0] == tokenizer.bos_token_id and is_first:
        offset = 1
        input_ids.append(prompt_chunks[0][0])
    for x in insert_separator(prompt_chunks, [image_token_index] * (offset + 1)):
        input_ids.extend(x[offset:])
    return input_ids


def tokenizer_image_region_token(prompt, tokenizer, region_token_index=OBJS_TOKEN_INDEX):
    """mm_utils.py:102-117."""
    input_ids = []
    chunks = prompt.split("<objs>")
    for idx, ck in enumerate(chunks):
        input_ids.extend(tokenizer_image_token(ck, tokenizer, is_first=(idx == 0)))
        if idx < len(chunks) - 1:
            input_ids.extend([region_token_index])
    return input_ids


def preprocess_region(region, image_size, target_size):
    """mm_utils.py:121-135."""
    x1, y1, x2, y2 = region
    sx = target_size[0] / image_size[0]
    sy = target_size[1] / image_size[1]
    return [x1 * sx, y1 * sy, x2 * sx, y2 * sy]


# =====================================================================================================================
# multimodal glue  (reference vitron/model/llava_arch.py:189-573)
# =====================================================================================================================
def splice_embeddings(input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], embed_tokens: torch.Tensor,
                      image_features: List[torch.Tensor], region_features: Optional[List[torch.Tensor]],
                      max_length: Optional[int] = None, padding_side: str = "right"):
    """prepare_inputs_labels_for_multimodal after the encoders ran: the list surgery of llava_arch.py:306-398
    (region branch; the plain branch :479-558 is the same loop without the -300 case).
    image_features: flat list, one [P,H] block per image / per video FRAME (llava_arch.py:259-270);
    region_features: parallel list ([1,H] per image, dummies for frames) or None.
    Returns (inputs_embeds [B,S,H], attention_mask [B,S] bool, position_ids [B,S] long)."""
    B, Lmax = input_ids.shape
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
    else:
        attention_mask = attention_mask.bool()
    ids_list = [ids[m] for ids, m in zip(input_ids, attention_mask)]  # :302
    new_embeds = []
    cur_image_idx = 0
    for cur in ids_list:
        num_images = int((cur == IMAGE_TOKEN_INDEX).sum())
        if num_images == 0:  # :310-324 (consumes one feature slot, appends an empty slice)
            new_embeds.append(torch.cat([embed_tokens[cur], image_features[cur_image_idx][0:0]], dim=0))
            cur_image_idx += 1
            continue
        specials = sorted(torch.where(cur == IMAGE_TOKEN_INDEX)[0].tolist() +
                          (torch.where(cur == OBJS_TOKEN_INDEX)[0].tolist() if region_features is not None else []))
        bounds = [-1] + specials + [cur.shape[0]]
        parts = []
        for i in range(len(bounds) - 1):
            seg = cur[bounds[i] + 1: bounds[i + 1]]
            parts.append(embed_tokens[seg])
            if i < len(specials):
                tok = int(cur[specials[i]])
                if tok == IMAGE_TOKEN_INDEX:
                    parts.append(image_features[cur_image_idx])
                    cur_image_idx += 1
                else:  # -300 binds to the most recently consumed image  :350-351
                    parts.append(region_features[cur_image_idx - 1])
        new_embeds.append(torch.cat(parts, dim=0))
    if max_length is not None:  # :363-366
        new_embeds = [x[:max_length] for x in new_embeds]
    max_len = max(x.shape[0] for x in new_embeds)
    H = embed_tokens.shape[1]
    out = torch.zeros((B, max_len, H), dtype=new_embeds[0].dtype)
    mask = torch.zeros((B, max_len), dtype=torch.bool)
    pos = torch.zeros((B, max_len), dtype=torch.long)
    for i, e in enumerate(new_embeds):  # :375-396
        n = e.shape[0]
        if n == 0:
            continue
        if padding_side == "left":
            out[i, -n:] = e
            mask[i, -n:] = True
            pos[i, -n:] = torch.arange(n)
        else:
            out[i, :n] = e
            mask[i, :n] = True
            pos[i, :n] = torch.arange(n)
    return out, mask, pos


# =====================================================================================================================
# LLaMA decoder  (transformers-4.31 LlamaForCausalLM; driven by reference llava_llama.py:57-102)
# =====================================================================================================================
def rope_tables(head_dim: int, max_pos: int, theta: float = 10000.0):
    """LlamaRotaryEmbedding (Appendix A): inv_freq = 1/theta^(arange(0,hd,2)/hd); freqs = outer(t, inv_freq);
    returns cos, sin [max_pos, hd/2] fp32 (the second half of the HF table repeats the first)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(max_pos, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq)
    return freqs.cos(), freqs.sin()


def _rope(x, cos, sin):
    """apply_rotary_pos_emb with rotate_half = cat(-x2, x1) (half-split). x [.., S, hd]; cos/sin [S, hd/2]."""
    h = x.shape[-1] // 2
    x1, x2 = x[..., :h], x[..., h:]
    return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1)


def rmsnorm(x, w, eps):
    """LlamaRMSNorm: variance in fp32, x * rsqrt(var + eps), then * weight."""
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w.float() * (x.float() * torch.rsqrt(var + eps))


def llama_forward(sd: SD, cfg: dict, inputs_embeds: torch.Tensor, position_ids: Optional[torch.Tensor] = None,
                  attention_mask: Optional[torch.Tensor] = None, past: Optional[list] = None,
                  emulate_bf16: bool = False, num_layers: Optional[int] = None, return_hidden: bool = False,
                  precise_qk: bool = False):
    """LlamaModel + lm_head. inputs_embeds [B,S,H]; attention_mask [B, past+S] (1 = attend) or None;
    past = list of (k,v) per layer, each [B,heads,Sp,hd]. Returns (logits [B,S,V] fp32, new_past[, hidden]).
    precise_qk = 3 (with an emulation mode): precise level 3 -- every decoder GEMM's A operand is hi (the 16-bit store) plus an MX-FP4 low half
    multiplied with the weights' MX-FP4 image (_lin_mx); q / k / v / P / V^T are stored as in the standard mode; the lm_head's operand is a
    16-bit pair as in level 2.
    precise_qk = 2 (with an emulation mode): precise level 2 -- every GEMM A operand is an operand pair (norm outputs, attention output,
    SwiGLU output, final norm; v is computed from the pair and stored once).
    precise_qk = 1 / True (with an emulation mode): the storage points of the kernels' precise_qk prefill -- the input-norm output reaches the
    q / k projection as an operand pair, q / k stay fp32 through the rotary embedding and are stored once, as pairs; v, P, V^T and
    everything behind the attention are stored as usual. (The K_lo.Q_lo term the kernel drops is 2^-22 of a score: not modelled.)"""
    B, S, H = inputs_embeds.shape
    heads = cfg["num_attention_heads"]
    hd = H // heads
    L = cfg["num_hidden_layers"] if num_layers is None else num_layers
    eps = cfg.get("rms_norm_eps", 1e-5)
    Sp = 0 if past is None else past[0][0].shape[2]
    if position_ids is None:
        position_ids = torch.arange(Sp, Sp + S).unsqueeze(0).expand(B, S)
    cos_t, sin_t = rope_tables(hd, int(position_ids.max()) + 1, cfg.get("rope_theta", 10000.0))
    cos = cos_t[position_ids].unsqueeze(1)  # [B,1,S,hd/2]
    sin = sin_t[position_ids].unsqueeze(1)
    # _prepare_decoder_attention_mask: causal + padding, additive finfo.min
    kv_len = Sp + S
    neg = torch.finfo(torch.float32).min
    causal = torch.full((S, kv_len), neg)
    causal = torch.triu(causal, diagonal=Sp + 1)
    mask = causal[None, None]
    if attention_mask is not None:
        pad = (1.0 - attention_mask[:, None, None, :].float()) * neg
        mask = torch.clamp(mask + pad, min=neg)
    x = inputs_embeds.float()
    new_past = []
    for l in range(L):
        p = f"model.layers.{l}."
        full = bool(emulate_bf16) and int(precise_qk) == 2          # level 2: every GEMM A operand is a pair
        mx = bool(emulate_bf16) and int(precise_qk) == 3            # level 3: every GEMM A operand is hi + an MX-FP4 low half
        if mx:
            hn = rmsnorm(x, sd[p + "input_layernorm.weight"], eps)
            lin = lambda t, name: _lin_mx(t, sd[p + name], emulate_bf16)
            q = _r(lin(hn, "self_attn.q_proj.weight"), emulate_bf16).view(B, S, heads, hd).transpose(1, 2)
            k = _r(lin(hn, "self_attn.k_proj.weight"), emulate_bf16).view(B, S, heads, hd).transpose(1, 2)
            v = _r(lin(hn, "self_attn.v_proj.weight"), emulate_bf16).view(B, S, heads, hd).transpose(1, 2)
            q = _r(_rope(q, cos, sin), emulate_bf16)
            k = _r(_rope(k, cos, sin), emulate_bf16)
        elif precise_qk and emulate_bf16:
            hn = rmsnorm(x, sd[p + "input_layernorm.weight"], eps)
            h, hp = _r(hn, emulate_bf16), _pair(hn, emulate_bf16)
            q = _lin(hp, sd[p + "self_attn.q_proj.weight"]).view(B, S, heads, hd).transpose(1, 2)
            k = _lin(hp, sd[p + "self_attn.k_proj.weight"]).view(B, S, heads, hd).transpose(1, 2)
            v = _r(_lin(hp if full else h, sd[p + "self_attn.v_proj.weight"]), emulate_bf16).view(B, S, heads, hd).transpose(1, 2)
            q = _pair(_rope(q, cos, sin), emulate_bf16)
            k = _pair(_rope(k, cos, sin), emulate_bf16)
        else:
            h = _r(rmsnorm(x, sd[p + "input_layernorm.weight"], eps), emulate_bf16)
            q = _r(_lin(h, sd[p + "self_attn.q_proj.weight"]), emulate_bf16).view(B, S, heads, hd).transpose(1, 2)
            k = _r(_lin(h, sd[p + "self_attn.k_proj.weight"]), emulate_bf16).view(B, S, heads, hd).transpose(1, 2)
            v = _r(_lin(h, sd[p + "self_attn.v_proj.weight"]), emulate_bf16).view(B, S, heads, hd).transpose(1, 2)
            q = _r(_rope(q, cos, sin), emulate_bf16)
            k = _r(_rope(k, cos, sin), emulate_bf16)
        if past is not None:
            k = torch.cat([past[l][0], k], dim=2)
            v = torch.cat([past[l][1], v], dim=2)
        new_past.append((k, v))
        s = q @ k.transpose(-1, -2) / math.sqrt(hd) + mask
        if emulate_bf16:
            pr = torch.exp(s - s.amax(-1, keepdim=True))
            # P in fp16 for the PV MFMA; the V^T pages hold the fp16 image of V: exact for bf16 values of magnitude 2^-14 .. 65504,
            # SATURATING beyond (vt_common.h op2_to_f16x2) -- emulated, so that a checkpoint with outlier V activations shows up as a
            # parity difference against the fp32 mode instead of passing silently
            o = (fp16_round(pr) @ fp16_store(v)) / pr.sum(-1, keepdim=True)
        else:
            o = torch.softmax(s, dim=-1, dtype=torch.float32) @ v
        if mx:
            x = x + lin(o.transpose(1, 2).reshape(B, S, H), "self_attn.o_proj.weight")
            hn = rmsnorm(x, sd[p + "post_attention_layernorm.weight"], eps)
            a = F.silu(lin(hn, "mlp.gate_proj.weight")) * lin(hn, "mlp.up_proj.weight")
            x = x + lin(a, "mlp.down_proj.weight")
            continue
        st = (lambda t: _pair(t, emulate_bf16)) if full else (lambda t: _r(t, emulate_bf16))
        o = st(o.transpose(1, 2).reshape(B, S, H))
        x = x + _lin(o, sd[p + "self_attn.o_proj.weight"])
        h = st(rmsnorm(x, sd[p + "post_attention_layernorm.weight"], eps))
        g = _lin(h, sd[p + "mlp.gate_proj.weight"])
        u = _lin(h, sd[p + "mlp.up_proj.weight"])
        a = st(F.silu(g) * u)
        x = x + _lin(a, sd[p + "mlp.down_proj.weight"])
    hidden = x
    full = bool(emulate_bf16) and int(precise_qk) >= 2     # the lm_head's operand is a 16-bit pair in level 3 too
    xn = _pair(rmsnorm(x, sd["model.norm.weight"], eps), emulate_bf16) if full else _r(rmsnorm(x, sd["model.norm.weight"], eps), emulate_bf16)
    logits = _lin(xn, sd["lm_head.weight"]).float()
    if return_hidden:
        return logits, new_past, hidden
    return logits, new_past


def greedy_generate(sd: SD, cfg: dict, inputs_embeds: torch.Tensor, attention_mask: torch.Tensor,
                    position_ids: torch.Tensor, max_new_tokens: int, emulate_bf16: bool = False,
                    eos_token_id: Optional[int] = None, ids_mask: Optional[torch.Tensor] = None, return_logits: bool = False):
    """Greedy GenerationMixin loop with Vitron's decode-step fix-up (reference llava_arch.py:196-205): on every
    step the mask is extended to past_len+1 and position_ids = sum(mask) - 1. Returns [B, <=max_new_tokens] ids.

    ids_mask=None: the mask that is extended is the spliced one and the first token is read at every sample's last VALID row --
    for batch 1 (what app.py / inference_image.py run) that is exactly the reference, and for a padded batch it is what every
    sample gets when run alone (the behaviour vitron_amd keeps by packing).
    ids_mask=[B, L_ids] (the mask generate() was CALLED with): the reference's padded-batch behaviour to the letter -- transformers
    4.31's greedy_search reads logits[:, -1] (a pad row for a right-padded shorter sample) and carries the ids-length mask, which
    the fix-up extends with ones, so pad rows of the spliced batch are attended (pinned by tests/golden/greedy_batch.npz)."""
    B = inputs_embeds.shape[0]
    embed = sd["model.embed_tokens.weight"].float()
    logits, past = llama_forward(sd, cfg, inputs_embeds, position_ids, attention_mask, None, emulate_bf16)
    if ids_mask is None:
        last = attention_mask.long().sum(1) - 1  # right padding: last valid position
        row = logits[torch.arange(B), last]
        mask = attention_mask.clone()
    else:
        row = logits[:, -1]
        mask = ids_mask.clone()
    nxt = row.argmax(-1)
    out = [nxt]
    rows = [row]
    finished = torch.zeros(B, dtype=torch.bool)
    for _ in range(max_new_tokens - 1):
        if eos_token_id is not None:
            finished |= nxt == eos_token_id
            if bool(finished.all()):
                break
        target = past[0][0].shape[2] + 1
        mask = torch.cat([mask, torch.ones((B, target - mask.shape[1]), dtype=mask.dtype)], dim=1)
        pos = mask.long().sum(1, keepdim=True) - 1
        logits, past = llama_forward(sd, cfg, embed[nxt].unsqueeze(1), pos, mask, past, emulate_bf16)
        nxt = logits[:, -1].argmax(-1)
        out.append(nxt)
        rows.append(logits[:, -1])
    if return_logits:   # (ids [B, steps], the logits row every id was the arg-max of [B, steps, V])
        return torch.stack(out, dim=1), torch.stack(rows, dim=1)
    return torch.stack(out, dim=1)


# =====================================================================================================================
# whole multimodal prefill, as the reference's LlavaLlamaForCausalLM.forward runs it (llava_llama.py:57-102)
# =====================================================================================================================
def multimodal_prepare(weights: dict, cfgs: dict, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor],
                       images: Sequence[torch.Tensor], regions: Optional[Sequence[Sequence[float]]],
                       max_length: Optional[int] = None, padding_side: str = "right", emulate_bf16: bool = False,
                       region_image_size: int = 224):
    """prepare_inputs_labels_for_multimodal, llava_arch.py:189-573, encoders included.
    weights: {'image_tower','video_tower','projector','region','llama'} state dicts; cfgs: {'image','video','llama'}.
    `images` is the reference's mixed list: 3-D tensors are images, 4-D tensors (C,T,H,W) are videos (:234-237)."""
    use_regions = regions is not None and len(regions) > 0          # :233
    image_idx = [i for i, im in enumerate(images) if im.dim() == 3]
    video_idx = [i for i, im in enumerate(images) if im.dim() == 4]
    feats: List = [None] * len(images)
    regs: List = [None] * len(images)
    if image_idx:
        batch = torch.stack([images[i] for i in image_idx])
        f = tower_features(weights["image_tower"], cfgs["image"], batch, -2, emulate_bf16)      # encode_images :168-181
        if use_regions:
            rb = [regions[i] for i in image_idx]                                                # :241
            r, _, _ = region_forward(weights["region"], f, rb, region_image_size, emulate_bf16)   # coordinates stay fp32 (the ABI takes them so)
        pf = projector_forward(weights["projector"], f, emulate_bf16)
        for j, i in enumerate(image_idx):
            feats[i] = pf[j]
            regs[i] = r[j] if use_regions else None
    if video_idx:
        batch = torch.stack([images[i] for i in video_idx])
        f = tower_features(weights["video_tower"], cfgs["video"], batch, -2, emulate_bf16)      # encode_videos :183-187
        pf = projector_forward(weights["projector"], f, emulate_bf16)
        for j, i in enumerate(video_idx):
            feats[i] = [pf[j][t] for t in range(pf.shape[1])]                                   # :255-258
            regs[i] = [None] * pf.shape[1]
    flat_f, flat_r = [], []
    for f, r in zip(feats, regs):
        if isinstance(f, list):
            flat_f += f
            flat_r += r
        else:
            flat_f.append(f)
            flat_r.append(r)
    embed = weights["llama"]["model.embed_tokens.weight"].float()
    return splice_embeddings(input_ids, attention_mask, embed, flat_f, flat_r if use_regions else None,
                             max_length, padding_side)


def multimodal_forward(weights: dict, cfgs: dict, input_ids, attention_mask, images, regions, max_length=None,
                       padding_side="right", emulate_bf16=False):
    """Returns (logits [B,S,V], inputs_embeds, mask, position_ids) of the prefill pass."""
    embeds, mask, pos = multimodal_prepare(weights, cfgs, input_ids, attention_mask, images, regions, max_length,
                                           padding_side, emulate_bf16)
    # the reference hands attention_mask/position_ids back as None when the caller passed None (:400-411)
    am = mask if attention_mask is not None else None
    logits, _ = llama_forward(weights["llama"], cfgs["llama"], embeds, pos if attention_mask is not None else None,
                              am, None, emulate_bf16)
    return logits, embeds, mask, pos


# =====================================================================================================================
# pre-processing (reference languagebind/image/processing_image.py:15-25, video/processing_video.py:45-53)
# NOTE parity unpinned for this stage: the reference's transforms need torchvision / pytorchvideo, which are not
# installed, so no golden could be generated from the reference itself; this restates the documented algorithm of the
# two libraries (both call torch.nn.functional.interpolate) on top of the same PyTorch build.
# =====================================================================================================================
OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess_image(img_u8_hwc: torch.Tensor, size: int = 224) -> torch.Tensor:
    """ToTensor -> Resize(size, BICUBIC) [tensor path: F.interpolate bicubic, align_corners=False, no antialias; short
    side -> size, long side int(size*long/short)] -> CenterCrop(size) -> Normalize.  [H,W,3] uint8 -> [3,size,size]."""
    x = img_u8_hwc.permute(2, 0, 1).float() / 255.0
    h, w = x.shape[1:]
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nh, nw = size, int(size * w / h)
    x = F.interpolate(x[None], size=(nh, nw), mode="bicubic", align_corners=False)[0]
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
    x = x[:, top:top + size, left:left + size]
    mean = torch.tensor(OPENAI_DATASET_MEAN)[:, None, None]
    std = torch.tensor(OPENAI_DATASET_STD)[:, None, None]
    return (x - mean) / std


def preprocess_video(frames_u8_thwc: torch.Tensor, size: int = 224, flip: bool = False) -> torch.Tensor:
    """x/255 -> NormalizeVideo -> ShortSideScale(size) [bilinear, long side floor(long/short*size)] -> CenterCropVideo
    [-> horizontal flip].  [T,H,W,3] uint8 -> [3,T,size,size]."""
    x = frames_u8_thwc.permute(3, 0, 1, 2).float() / 255.0          # (C,T,H,W)
    mean = torch.tensor(OPENAI_DATASET_MEAN)[:, None, None, None]
    std = torch.tensor(OPENAI_DATASET_STD)[:, None, None, None]
    x = (x - mean) / std
    h, w = x.shape[2:]
    if w < h:
        nh, nw = int(math.floor((float(h) / w) * size)), size
    else:
        nh, nw = size, int(math.floor((float(w) / h) * size))
    x = F.interpolate(x, size=(nh, nw), mode="bilinear", align_corners=False)
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
    x = x[:, :, top:top + size, left:left + size]
    return x.flip(-1) if flip else x
