"""Torch-tensor front end of the primitive C-ABI operators (include/vitron_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every function below checks its tensors,
passes raw device pointers + the current stream to libvitron_hip.so and returns torch tensors. There is no
torch-op fallback: without the shared library, or on a CPU tensor, these raise.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import (CFG_AUTO, EPI_BF16, EPI_BF16_GELU, EPI_BF16_QGELU, EPI_BF16_RELU, EPI_F32, EPI_F32_RESID,  # noqa: F401
                   EPI_SWIGLU_BF16, PAGE_TOKENS)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _op16(t: torch.Tensor, name: str):
    """(library, dtype) for a 16-bit operand tensor: its dtype (bf16 / fp16) picks the library build."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.VitronHipError(f"{name}: expected a CUDA/HIP tensor (vitron_amd has no CPU path)")
    if t.dtype not in (torch.bfloat16, torch.float16):
        raise _lib.VitronHipError(f"{name}: expected a bf16 or fp16 tensor, got {t.dtype}")
    return _lib.lib_for(t), t.dtype


def _chk(t: torch.Tensor, dtype, name: str):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.VitronHipError(f"{name}: expected a CUDA/HIP tensor (vitron_amd has no CPU path)")
    if t.dtype != dtype:
        raise _lib.VitronHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.VitronHipError(f"{name}: tensor must be contiguous")


def gemm_plan(M: int, N: int, K: int, epi: int = EPI_BF16):
    """(cfg, rows_first) the dispatcher picks for an AUTO vt_gemm_bf16 of this shape (host logic, no launch; include/vitron_hip.h)."""
    import ctypes
    lib = _lib.load_any()
    cfg, rows = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.vt_gemm_plan_query(M, N, K, epi, ctypes.byref(cfg), ctypes.byref(rows)), "vt_gemm_plan_query", lib)
    return cfg.value, rows.value


def gemm_plan_cols(M: int, N: int, K: int, epi: int = EPI_BF16):
    """(cfg, rows_first, cols_first) of the dispatcher's plan (vt_gemm_plan_query2): cols_first > 0 = the column split of round 5."""
    import ctypes
    lib = _lib.load_any()
    cfg, rows, cols = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.vt_gemm_plan_query2(M, N, K, epi, ctypes.byref(cfg), ctypes.byref(rows), ctypes.byref(cols)), "vt_gemm_plan_query2", lib)
    return cfg.value, rows.value, cols.value


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epi: int = EPI_BF16,
         out: Optional[torch.Tensor] = None, cfg: int = CFG_AUTO, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = epi(row_scale[:, None] * (a[M,K] @ w[N,K]^T) + bias). EPI_F32_RESID accumulates into `out` (fp32, required)."""
    lib, dt = _op16(a, "gemm.a")
    _chk(a, dt, "gemm.a")
    _chk(w, dt, "gemm.w")
    M, K = a.shape
    N, K2 = w.shape
    if K != K2:
        raise _lib.VitronHipError(f"gemm: K mismatch {K} vs {K2}")
    if bias is not None:
        _chk(bias, torch.float32, "gemm.bias")
    n_out = N // 2 if epi == EPI_SWIGLU_BF16 else N
    odt = torch.float32 if epi in (EPI_F32, EPI_F32_RESID) else dt
    if out is None:
        if epi == EPI_F32_RESID:
            raise _lib.VitronHipError("gemm: EPI_F32_RESID needs `out` (the fp32 residual stream)")
        out = torch.empty((M, n_out), device=a.device, dtype=odt)
    _chk(out, odt, "gemm.out")
    if row_scale is not None:
        _chk(row_scale, torch.float32, "gemm.row_scale")
        if row_scale.numel() != M:
            raise _lib.VitronHipError(f"gemm: row_scale has {row_scale.numel()} elements for M={M}")
    _lib.check(lib.vt_gemm_bf16(_p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), _p(bias), M, N, K, epi,
                                cfg, _p(row_scale), _stream()), "vt_gemm_bf16", lib)
    return out


def gemm_resid_splitk(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, bias: Optional[torch.Tensor] = None, ksplit: int = 0,
                      partials: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out (fp32) += a @ w^T + bias, optionally as a two-pass split-K through the fp32 workspace `partials`."""
    lib, dt = _op16(a, "gemm_resid_splitk.a")
    _chk(a, dt, "gemm_resid_splitk.a")
    _chk(w, dt, "gemm_resid_splitk.w")
    _chk(out, torch.float32, "gemm_resid_splitk.out")
    M, K = a.shape
    N = w.shape[0]
    nbytes = 0 if partials is None else partials.numel() * partials.element_size()
    _lib.check(lib.vt_gemm_bf16_resid_splitk(_p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), _p(bias), M, N, K,
                                             int(ksplit), _p(partials), nbytes, _stream()), "vt_gemm_bf16_resid_splitk", lib)
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, temb: Optional[torch.Tensor] = None,
              tokens_per_frame: int = 0, dtype=None) -> torch.Tensor:
    """16-bit (`dtype`: bf16 default / fp16) LayerNorm of the fp32 rows of x; with temb ([T,D] fp32)
    x[row] += temb[(row//tokens_per_frame)%T] in place."""
    dtype = dtype or _lib.torch_dtype()
    lib = _lib.lib_for(dtype)
    _chk(x, torch.float32, "layernorm.x")
    rows, D = x.shape
    y = torch.empty((rows, D), device=x.device, dtype=dtype)
    T = 0 if temb is None else temb.shape[0]
    _lib.check(lib.vt_layernorm(_p(x), _p(temb), T, tokens_per_frame, _p(gamma), _p(beta), _p(y), rows, D, eps, _stream()),
               "vt_layernorm", lib)
    return y


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, idx: Optional[torch.Tensor] = None, dtype=None) -> torch.Tensor:
    dtype = dtype or _lib.torch_dtype()
    lib = _lib.lib_for(dtype)
    _chk(x, torch.float32, "rmsnorm.x")
    rows = x.shape[0] if idx is None else idx.shape[0]
    D = x.shape[1]
    y = torch.empty((rows, D), device=x.device, dtype=dtype)
    _lib.check(lib.vt_rmsnorm(_p(x), _p(idx), _p(w), _p(y), rows, D, eps, _stream()), "vt_rmsnorm", lib)
    return y


def seq_desc_tensor(rows: Sequence[Sequence[int]], device) -> torch.Tensor:
    """int32 [nseq,4] = (q_row0, q_len, kv_len, table_off)."""
    return torch.tensor(rows, dtype=torch.int32, device=device).reshape(-1, 4)


def kv_tiles(qkv: torch.Tensor, q_col0: int, k_col0: int, v_col0: int, k_tiles: torch.Tensor, vt_tiles: torch.Tensor,
             tile_table: torch.Tensor, seq_desc: torch.Tensor, max_new_tiles: int, heads: int, head_dim: int,
             rope_cos: Optional[torch.Tensor] = None, rope_sin: Optional[torch.Tensor] = None,
             positions: Optional[torch.Tensor] = None) -> None:
    lib, dt = _op16(qkv, "kv_tiles.qkv")
    _chk(qkv, dt, "kv_tiles.qkv")
    _chk(k_tiles, dt, "kv_tiles.k_tiles")
    if vt_tiles.element_size() != 2:        # fp16 bits in both builds; tests hand in raw 16-bit storage of either dtype
        raise _lib.VitronHipError("kv_tiles.vt_tiles: expected a 16-bit tensor (the V^T pages hold fp16 bits)")
    _lib.check(lib.vt_kv_tiles(_p(qkv), qkv.stride(0), q_col0, k_col0, v_col0, _p(k_tiles), _p(vt_tiles), _p(tile_table),
                               _p(seq_desc), seq_desc.shape[0], max_new_tiles, heads, head_dim, _p(rope_cos), _p(rope_sin),
                               _p(positions), _stream()), "vt_kv_tiles", lib)


def flash_attn(q: torch.Tensor, k_tiles: torch.Tensor, vt_tiles: torch.Tensor, tile_table: torch.Tensor,
               seq_desc: torch.Tensor, max_q_len: int, heads: int, head_dim: int, causal: bool, scale: float,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q: bf16 [rows, ldq] view whose first heads*head_dim columns are the queries (e.g. the fused QKV buffer)."""
    lib, dt = _op16(q, "flash_attn.q")
    if k_tiles.dtype != dt:
        raise _lib.VitronHipError(f"flash_attn: q is {dt} but the K tiles are {k_tiles.dtype}")
    if out is None:
        out = torch.empty((q.shape[0], heads * head_dim), device=q.device, dtype=dt)
    _lib.check(lib.vt_flash_attn(_p(q), q.stride(0), _p(k_tiles), _p(vt_tiles), _p(tile_table), _p(seq_desc),
                                 seq_desc.shape[0], max_q_len, _p(out), out.stride(0), heads, head_dim, int(causal),
                                 float(scale), _stream()), "vt_flash_attn", lib)
    return out


def flash_attn_select(kernel: int) -> None:
    """Process-wide choice of the head_dim-128 prefill kernel in BOTH operand builds (vt_flash_attn_select, include/vitron_hip.h):
    0 automatic, 1 two-waves-per-SIMD kernel, 2 one-wave-per-SIMD hand-placed kernel, 3 that kernel unplaced (its reference), 4 the
    hand-placed kernel in its persistent form (4 | (n << 8): at most n workgroups), 5 kernel 2 in the grid's natural block order
    instead of the planned dispatch order."""
    for op in ("bf16", "fp16"):
        lib = _lib.load(operand=op)
        _lib.check(lib.vt_flash_attn_select(int(kernel)), "vt_flash_attn_select", lib)


def attn_decode(q: torch.Tensor, k_tiles: torch.Tensor, vt_tiles: torch.Tensor, tile_table: torch.Tensor,
                seq_desc: torch.Tensor, heads: int, head_dim: int, scale: float, max_kv_len: int) -> torch.Tensor:
    lib, dt = _op16(q, "attn_decode.q")
    out = torch.empty((q.shape[0], heads * head_dim), device=q.device, dtype=dt)
    nseq = seq_desc.shape[0]
    scratch = torch.empty(lib.vt_attn_decode_scratch_bytes(nseq, heads, head_dim, int(max_kv_len)), dtype=torch.uint8, device=q.device)
    _lib.check(lib.vt_attn_decode(_p(q), q.stride(0), _p(k_tiles), _p(vt_tiles), _p(tile_table), _p(seq_desc), nseq, _p(out),
                                  out.stride(0), heads, head_dim, float(scale), int(max_kv_len), _p(scratch), scratch.numel(),
                                  _stream()), "vt_attn_decode", lib)
    return out


def attn_decode_fused(qkv: torch.Tensor, q_col0: int, k_col0: int, v_col0: int, k_tiles: torch.Tensor, vt_tiles: torch.Tensor,
                      tile_table: torch.Tensor, seq_desc: torch.Tensor, heads: int, head_dim: int, scale: float,
                      rope_cos: Optional[torch.Tensor] = None, rope_sin: Optional[torch.Tensor] = None,
                      positions: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One decode step's attention: rotary on the new q/k rows, k/v append to the paged tiles, attention, combine."""
    lib, dt = _op16(qkv, "attn_decode_fused.qkv")
    _chk(qkv, dt, "attn_decode_fused.qkv")
    _chk(k_tiles, dt, "attn_decode_fused.k_tiles")
    out = torch.empty((qkv.shape[0], heads * head_dim), device=qkv.device, dtype=dt)
    _lib.check(lib.vt_attn_decode_fused(_p(qkv), qkv.stride(0), q_col0, k_col0, v_col0, _p(k_tiles), _p(vt_tiles), _p(tile_table),
                                        _p(seq_desc), seq_desc.shape[0], _p(out), out.stride(0), heads, head_dim, float(scale),
                                        _p(rope_cos), _p(rope_sin), _p(positions), _stream()), "vt_attn_decode_fused", lib)
    return out


def attn_temporal(qkv: torch.Tensor, B: int, T: int, N: int, heads: int) -> torch.Tensor:
    lib, dt = _op16(qkv, "attn_temporal.qkv")
    _chk(qkv, dt, "attn_temporal.qkv")
    out = torch.empty((qkv.shape[0], heads * 64), device=qkv.device, dtype=dt)
    _lib.check(lib.vt_attn_temporal(_p(qkv), _p(out), B, T, N, heads, _stream()), "vt_attn_temporal", lib)
    return out


def im2col(pixels: torch.Tensor, patch: int, k_pad: int, dtype=None) -> torch.Tensor:
    """pixels fp32 or 16-bit; patches come out in `dtype` (default: the pixels' own 16-bit dtype, else the default operand)."""
    if dtype is None:
        dtype = pixels.dtype if pixels.dtype in (torch.bfloat16, torch.float16) else _lib.torch_dtype()
    lib = _lib.lib_for(dtype)
    if pixels.dtype not in (dtype, torch.float32):
        raise _lib.VitronHipError(f"im2col: pixels are {pixels.dtype}, expected {dtype} or fp32")
    video = pixels.dim() == 5
    if video:
        B, _, T, H, W = pixels.shape
    else:
        B, _, H, W = pixels.shape
        T = 1
    dt = _lib.DTYPE_F32 if pixels.dtype == torch.float32 else _lib.DTYPE_BF16
    out = torch.empty((B * T * (H // patch) * (W // patch), k_pad), device=pixels.device, dtype=dtype)
    _lib.check(lib.vt_im2col(_p(pixels.contiguous()), dt, _p(out), B, T, H, W, patch, k_pad, int(video), _stream()), "vt_im2col", lib)
    return out


def _chk_rows(t: torch.Tensor, dtype, name: str):
    """2-D tensor whose rows are contiguous (row stride free): logits narrowed to the true vocabulary of a padded lm_head."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.VitronHipError(f"{name}: expected a CUDA/HIP tensor (vitron_amd has no CPU path)")
    if t.dtype != dtype or t.dim() != 2 or t.stride(1) != 1:
        raise _lib.VitronHipError(f"{name}: expected a 2-D {dtype} tensor with contiguous rows")


def embed_splice(tok_table: torch.Tensor, vis: Optional[torch.Tensor], reg: Optional[torch.Tensor], plan: torch.Tensor) -> torch.Tensor:
    lib, dt = _op16(tok_table, "embed_splice.tok_table")
    _chk(tok_table, dt, "embed_splice.tok_table")
    for name, t in (("vis", vis), ("reg", reg)):
        if t is not None and t.dtype != dt:
            raise _lib.VitronHipError(f"embed_splice.{name}: {t.dtype} rows cannot be spliced into a {dt} sequence")
    rows, H = plan.shape[0], tok_table.shape[1]
    out = torch.empty((rows, H), device=tok_table.device, dtype=dt)
    _lib.check(lib.vt_embed_splice(_p(tok_table), tok_table.shape[0], _p(vis), 0 if vis is None else vis.shape[0], _p(reg),
                                   0 if reg is None else reg.shape[0], _p(plan), rows, H, _p(out), _stream()), "vt_embed_splice", lib)
    return out


def argmax(logits: torch.Tensor) -> torch.Tensor:
    lib = _lib.load_any()
    _chk_rows(logits, torch.float32, "argmax.logits")
    rows, V = logits.shape
    out = torch.empty((rows,), device=logits.device, dtype=torch.int32)
    _lib.check(lib.vt_argmax(_p(logits), rows, V, logits.stride(0), _p(out), _stream()), "vt_argmax", lib)
    return out


def decode_feed(tok_table: torch.Tensor, next_ids: torch.Tensor, finished: torch.Tensor, eos_ids: Optional[torch.Tensor],
                pad_id: int, tokens_out: torch.Tensor, x: torch.Tensor, seq_desc: torch.Tensor, positions: torch.Tensor) -> None:
    """In-place decode-loop feedback (vt_decode_feed): token resolution, EOS flags, next input rows, metadata advance."""
    lib, dt = _op16(tok_table, "decode_feed.tok_table")
    _chk(tok_table, dt, "decode_feed.tok_table")
    _chk(x, dt, "decode_feed.x")
    for name, t in (("next_ids", next_ids), ("finished", finished), ("tokens_out", tokens_out), ("seq_desc", seq_desc),
                    ("positions", positions)):
        _chk(t, torch.int32, f"decode_feed.{name}")
    nseq = next_ids.shape[0]
    if finished.numel() != nseq or tokens_out.numel() != 2 * nseq or seq_desc.numel() != 4 * nseq or positions.numel() != nseq \
            or tuple(x.shape) != (nseq, tok_table.shape[1]):
        raise _lib.VitronHipError("decode_feed: buffer sizes do not match the number of sequences")
    n_eos = 0 if eos_ids is None else int(eos_ids.numel())
    if n_eos:
        _chk(eos_ids, torch.int32, "decode_feed.eos_ids")
    _lib.check(lib.vt_decode_feed(_p(tok_table), tok_table.shape[1], tok_table.shape[0], _p(next_ids), _p(finished),
                                  _p(eos_ids) if n_eos else None, n_eos, int(pad_id), _p(tokens_out), _p(x), _p(seq_desc),
                                  _p(positions), nseq, _stream()), "vt_decode_feed", lib)


def sample_top_p(logits: torch.Tensor, temperature: float, top_p: float, seed: int, step: int, return_kept: bool = False,
                 top_k: int = 0):
    """One sampled token id per row (int32), on device; see vt_sample_top_p in include/vitron_hip.h."""
    lib = _lib.load_any()
    _chk_rows(logits, torch.float32, "sample_top_p.logits")
    rows, V = logits.shape
    out = torch.empty((rows,), device=logits.device, dtype=torch.int32)
    kept = torch.empty((rows,), device=logits.device, dtype=torch.int32) if return_kept else None
    _lib.check(lib.vt_sample_top_p(_p(logits), rows, V, logits.stride(0), float(temperature), int(top_k or 0),
                                   float(top_p if top_p else 1.0), int(seed) & (2 ** 64 - 1), int(step), _p(out), _p(kept),
                                   _stream()), "vt_sample_top_p", lib)
    return (out, kept) if return_kept else out


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """Mean cross entropy of fp32 logits [rows, V] against int labels [rows] (rows with `ignore_index` skipped); 0-dim tensor."""
    lib = _lib.load_any()
    _chk_rows(logits, torch.float32, "cross_entropy.logits")
    rows, V = logits.shape
    lab = labels.to(device=logits.device, dtype=torch.int32).contiguous()
    # torch.nn.functional.cross_entropy raises on a label outside [0, V) that is not ignore_index; the kernel would drop such a
    # row from the mean, i.e. turn a vocabulary mismatch into a plausible loss. (One read-back: this is the scoring path.)
    bad = (lab != int(ignore_index)) & ((lab < 0) | (lab >= V))
    if bool(bad.any()):
        raise IndexError(f"cross_entropy: label {int(lab[bad][0])} outside the vocabulary ({V} columns)")
    nll = torch.empty((rows,), device=logits.device, dtype=torch.float32)
    loss = torch.empty((1,), device=logits.device, dtype=torch.float32)
    _lib.check(lib.vt_cross_entropy(_p(logits), rows, V, logits.stride(0), _p(lab), int(ignore_index), _p(nll), _p(loss), _stream()),
               "vt_cross_entropy", lib)
    return loss[0]


# ---- precise level 3: MX-FP4 images of the rounding remainders (include/vitron_hip.h "precise level 3") ---------------------------------
def mx4_aexp_bytes(M: int, K: int) -> int:
    return int(_lib.load_any().vt_mx4_aexp_bytes(M, K))


def mx4_quant_weights(w: torch.Tensor):
    """(W4 uint8 [N][K/2], wexp uint8 [N]) of a 16-bit weight matrix [N][K]."""
    lib, dt = _op16(w, "mx4_quant_weights.w")
    _chk(w, dt, "mx4_quant_weights.w")
    N, K = w.shape
    w4 = torch.empty((N, K // 2), device=w.device, dtype=torch.uint8)
    wexp = torch.empty((N,), device=w.device, dtype=torch.uint8)
    _lib.check(lib.vt_mx4_quant_weights(_p(w), K, N, K, _p(w4), _p(wexp), _stream()), "vt_mx4_quant_weights", lib)
    return w4, wexp


def mx4_quant_lo(lo: torch.Tensor):
    """(A4 uint8 [M][K/2], aexp uint8 [mx4_aexp_bytes(M, K)]) of a 16-bit remainder [M][K]."""
    lib, dt = _op16(lo, "mx4_quant_lo.lo")
    _chk(lo, dt, "mx4_quant_lo.lo")
    M, K = lo.shape
    a4 = torch.empty((M, K // 2), device=lo.device, dtype=torch.uint8)
    aexp = torch.zeros((mx4_aexp_bytes(M, K),), device=lo.device, dtype=torch.uint8)
    _lib.check(lib.vt_mx4_quant_lo(_p(lo), K, M, K, _p(a4), _p(aexp), _stream()), "vt_mx4_quant_lo", lib)
    return a4, aexp


def rmsnorm_mx(x: torch.Tensor, w: torch.Tensor, eps: float, dtype: torch.dtype, idx: Optional[torch.Tensor] = None):
    """(y op16 [rows][D], A4, aexp): RMSNorm with the level 3 operand out."""
    _chk(x, torch.float32, "rmsnorm_mx.x")
    _chk(w, torch.float32, "rmsnorm_mx.w")
    lib = _lib.load(operand=_lib.operand_of(dtype))
    rows = x.shape[0] if idx is None else idx.shape[0]
    D = x.shape[1]
    y = torch.empty((rows, D), device=x.device, dtype=dtype)
    a4 = torch.empty((rows, D // 2), device=x.device, dtype=torch.uint8)
    aexp = torch.zeros((mx4_aexp_bytes(rows, D),), device=x.device, dtype=torch.uint8)
    _lib.check(lib.vt_rmsnorm_mx(_p(x), _p(idx), _p(w), _p(y), _p(a4), _p(aexp), rows, D, eps, _stream()), "vt_rmsnorm_mx", lib)
    return y, a4, aexp


def gemm_mx(a: torch.Tensor, a4: torch.Tensor, aexp: torch.Tensor, w: torch.Tensor, w4: torch.Tensor, wexp: torch.Tensor,
            bias: Optional[torch.Tensor] = None, epi: int = EPI_BF16, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = epi(a @ w^T + (a4 2^aexp) @ (w4 2^wexp)^T + bias) in one launch (vt_gemm_mx)."""
    lib, dt = _op16(a, "gemm_mx.a")
    _chk(a, dt, "gemm_mx.a")
    _chk(w, dt, "gemm_mx.w")
    for t, n in ((a4, "a4"), (aexp, "aexp"), (w4, "w4"), (wexp, "wexp")):
        _chk(t, torch.uint8, "gemm_mx." + n)
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K or tuple(a4.shape) != (M, K // 2) or tuple(w4.shape) != (N, K // 2) or wexp.numel() != N or \
            aexp.numel() < mx4_aexp_bytes(M, K):
        raise _lib.VitronHipError("gemm_mx: operand shapes do not match")
    if bias is not None:
        _chk(bias, torch.float32, "gemm_mx.bias")
    n_out = N // 2 if epi == EPI_SWIGLU_BF16 else N
    odt = torch.float32 if epi in (EPI_F32, EPI_F32_RESID) else dt
    if out is None:
        if epi == EPI_F32_RESID:
            raise _lib.VitronHipError("gemm_mx: EPI_F32_RESID needs `out` (the fp32 residual stream)")
        out = torch.empty((M, n_out), device=a.device, dtype=odt)
    _chk(out, odt, "gemm_mx.out")
    _lib.check(lib.vt_gemm_mx(_p(a), K, _p(a4), _p(aexp), _p(w), K, _p(w4), _p(wexp), _p(out), out.shape[1], _p(bias), M, N, K, epi,
                              _stream()), "vt_gemm_mx", lib)
    return out


def gemm_mx_swiglu(a, a4, aexp, w, w4, wexp):
    """(h op16 [M][N/2], h4 uint8 [M][N/4], hexp): level 3 gate/up -- the SwiGLU output and the MX-FP4 image of its rounding remainder."""
    lib, dt = _op16(a, "gemm_mx_swiglu.a")
    M, K = a.shape
    N = w.shape[0]
    h = torch.empty((M, N // 2), device=a.device, dtype=dt)
    h4 = torch.empty((M, N // 4), device=a.device, dtype=torch.uint8)
    hexp = torch.zeros((mx4_aexp_bytes(M, N // 2),), device=a.device, dtype=torch.uint8)
    _lib.check(lib.vt_gemm_mx_swiglu(_p(a), K, _p(a4), _p(aexp), _p(w), K, _p(w4), _p(wexp), _p(h), N // 2, _p(h4), _p(hexp), M, N, K, _stream()),
               "vt_gemm_mx_swiglu", lib)
    return h, h4, hexp


def gemm_mx_resid(a, a4, aexp, w, w4, wexp, out: torch.Tensor, partials: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out (fp32 [M][N]) += a @ w^T + (a4 2^aexp) @ (w4 2^wexp)^T; `partials` (fp32 scratch) lets a grid that spills over whole rounds run
    its trailing row blocks as K ranges (vt_gemm_mx_resid)."""
    lib, dt = _op16(a, "gemm_mx_resid.a")
    _chk(out, torch.float32, "gemm_mx_resid.out")
    M, K = a.shape
    N = w.shape[0]
    _lib.check(lib.vt_gemm_mx_resid(_p(a), K, _p(a4), _p(aexp), _p(w), K, _p(w4), _p(wexp), _p(out), N, M, N, K, _p(partials),
                                    0 if partials is None else partials.numel() * 4, _stream()), "vt_gemm_mx_resid", lib)
    return out


def flash_attn_mx(q, k_tiles, vt_tiles, tile_table, seq_desc, max_q_len: int, heads: int, scale: float):
    """Causal head_dim-128 attention with the level 3 operand out: (O op16 [rows][heads*128], O4, oexp). q: a [rows, ldq] view whose first
    heads*128 columns are the queries."""
    lib, dt = _op16(q, "flash_attn_mx.q")
    rows = q.shape[0]
    o = torch.empty((rows, heads * 128), device=q.device, dtype=dt)
    o4 = torch.empty((rows, heads * 64), device=q.device, dtype=torch.uint8)
    oexp = torch.zeros((mx4_aexp_bytes(rows, heads * 128),), device=q.device, dtype=torch.uint8)
    _lib.check(lib.vt_flash_attn_mx(_p(q), q.stride(0), _p(k_tiles), _p(vt_tiles), _p(tile_table), _p(seq_desc), seq_desc.shape[0], max_q_len,
                                    _p(o), heads * 128, _p(o4), _p(oexp), heads, scale, _stream()), "vt_flash_attn_mx", lib)
    return o, o4, oexp


def gemm_mx_gelu(a, a4, aexp, w, w4, wexp, bias: Optional[torch.Tensor] = None, quick: bool = False):
    """(h op16 [M][N], h4 uint8 [M][N/2], hexp): the towers' fc1 in precise level 1 -- gelu(a @ w^T + (a4..)(w4..)^T + bias) and the MX-FP4 image of
    its rounding remainder (vt_gemm_mx_gelu)."""
    lib, dt = _op16(a, "gemm_mx_gelu.a")
    M, K = a.shape
    N = w.shape[0]
    h = torch.empty((M, N), device=a.device, dtype=dt)
    h4 = torch.empty((M, N // 2), device=a.device, dtype=torch.uint8)
    hexp = torch.zeros((mx4_aexp_bytes(M, N),), device=a.device, dtype=torch.uint8)
    _lib.check(lib.vt_gemm_mx_gelu(_p(a), K, _p(a4), _p(aexp), _p(w), K, _p(w4), _p(wexp), _p(bias), _p(h), N, _p(h4), _p(hexp), M, N, K, int(quick),
                                   _stream()), "vt_gemm_mx_gelu", lib)
    return h, h4, hexp


def layernorm_mx(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, dtype: torch.dtype):
    """(y op16 [rows][D], A4, aexp): LayerNorm with the level 3 operand out (vt_layernorm_mx)."""
    _chk(x, torch.float32, "layernorm_mx.x")
    lib = _lib.load(operand=_lib.operand_of(dtype))
    rows, D = x.shape
    y = torch.empty((rows, D), device=x.device, dtype=dtype)
    a4 = torch.empty((rows, D // 2), device=x.device, dtype=torch.uint8)
    aexp = torch.zeros((mx4_aexp_bytes(rows, D),), device=x.device, dtype=torch.uint8)
    _lib.check(lib.vt_layernorm_mx(_p(x), _p(gamma), _p(beta), _p(y), _p(a4), _p(aexp), rows, D, eps, _stream()), "vt_layernorm_mx", lib)
    return y, a4, aexp
