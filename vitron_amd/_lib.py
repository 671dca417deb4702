"""ctypes binding of libvitron_hip.so / libvitron_hip_f16.so (C ABI: include/vitron_hip.h).

The same ABI exists in two operand formats (include/vitron_hip.h "OPERAND FORMAT"): bf16 (libvitron_hip.so, the benchmark's
dtype) and IEEE fp16 (libvitron_hip_f16.so, the reference's inference dtype, vitron/model/builder.py:47). A tensor's torch
dtype selects the library: `lib_for(t)`.

There is deliberately NO fallback: if the shared library is missing or fails to load, every operator of
vitron_amd raises. `load()` builds it with hipcc when the sources are newer than the binary (or the binary is
absent) and the toolchain is available.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libvitron_hip.so"
LIB_PATHS = {"bf16": LIB_PATH, "fp16": PKG_DIR / "libvitron_hip_f16.so"}
OPERAND_BF16, OPERAND_FP16 = 0, 1
ABI_VERSION = 113   # == VT_ABI_VERSION of include/vitron_hip.h; load() refuses a library that reports anything else

# ---- enums (mirror include/vitron_hip.h) ---------------------------------------------------------------------
EPI_BF16, EPI_BF16_GELU, EPI_BF16_QGELU, EPI_BF16_RELU, EPI_F32_RESID, EPI_F32, EPI_SWIGLU_BF16 = range(7)
CFG_AUTO, CFG_SKINNY, CFG_128x128, CFG_256x128, CFG_256x256, CFG_64x128, CFG_256x256_P8, _CFG_RESERVED_7, CFG_256x256_RP, CFG_SKINNY_REG, CFG_256x256_P4 = range(11)
CFG_256x256_W4 = 13
CFG_320x256_W4 = 14
CFG_160x128_W4 = 15
CFG_224x256_W4 = 16
DTYPE_BF16, DTYPE_F32 = 0, 1
ACT_GELU, ACT_QUICK_GELU = 0, 1
PAGE_TOKENS = 64

u16p = C.c_void_p  # all device pointers travel as void*
vp = C.c_void_p


class VtRegionWeights(C.Structure):
    _fields_ = [("in_dim", C.c_int), ("out_dim", C.c_int), ("mlp_w", vp * 2), ("mlp_b", vp * 2),
                ("loc_w0", vp), ("loc_b0", vp), ("final_w", vp), ("final_b", vp)]


class VtVitLayer(C.Structure):
    _fields_ = [(n, vp) for n in (
        "t_ln_g", "t_ln_b", "t_embed", "t_wqkv", "t_bqkv", "t_wo", "t_bo",
        "ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_g", "ln2_b", "w1", "b1", "w2", "b2",
        "t_ln2_g", "t_ln2_b", "t_w1", "t_b1", "t_w2", "t_b2", "w14", "w1_e", "w24", "w2_e")]


class VtVitModel(C.Structure):
    _fields_ = [("image_size", C.c_int), ("patch", C.c_int), ("hidden", C.c_int), ("heads", C.c_int),
                ("intermediate", C.c_int), ("num_layers", C.c_int), ("num_frames", C.c_int),
                ("add_time_attn", C.c_int), ("act", C.c_int), ("ln_eps", C.c_float), ("k_pad", C.c_int),
                ("w_patch", vp), ("cls", vp), ("pos", vp), ("pre_ln_g", vp), ("pre_ln_b", vp),
                ("layers", C.POINTER(VtVitLayer)), ("precise", C.c_int), ("out_feats_lo", vp)]


class VtLlamaLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("rms1", "wqkv", "wo", "rms2", "wgu", "wdown",
                                  "wqkv4", "wqkv_e", "wo4", "wo_e", "wgu4", "wgu_e", "wdown4", "wdown_e")]


class VtLlamaModel(C.Structure):
    _fields_ = [("hidden", C.c_int), ("heads", C.c_int), ("head_dim", C.c_int), ("intermediate", C.c_int),
                ("num_layers", C.c_int), ("vocab", C.c_int), ("rms_eps", C.c_float), ("final_norm", vp),
                ("lm_head", vp), ("rope_cos", vp), ("rope_sin", vp), ("rope_len", C.c_int),
                ("layers", C.POINTER(VtLlamaLayer)), ("prefill_norm_fold", C.c_int), ("qkv_fuse", C.c_int), ("precise_qk", C.c_int), ("embeds_lo", vp), ("hidden_trace", vp)]


class VtKvCache(C.Structure):
    _fields_ = [("k", vp), ("vt", vp), ("num_pages", C.c_int)]


_i, _f, _sz = C.c_int, C.c_float, C.c_size_t
# name -> (restype, argtypes); this table is also what tests check against the header
SIGNATURES = {
    "vt_version": (_i, []),
    "vt_operand_format": (_i, []),
    "vt_last_error": (_i, [C.c_char_p, _sz]),
    "vt_gemm_bf16": (_i, [vp, _i, vp, _i, vp, _i, vp, _i, _i, _i, _i, _i, vp, vp]),
    "vt_layernorm": (_i, [vp, vp, _i, _i, vp, vp, vp, _i, _i, _f, vp]),
    "vt_rmsnorm": (_i, [vp, vp, vp, vp, _i, _i, _f, vp]),
    "vt_flash_attn": (_i, [vp, _i, vp, vp, vp, vp, _i, _i, vp, _i, _i, _i, _i, _f, vp]),
    "vt_flash_attn_select": (_i, [_i]),
    "vt_attn_decode_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "vt_attn_decode": (_i, [vp, _i, vp, vp, vp, vp, _i, vp, _i, _i, _i, _f, _i, vp, _sz, vp]),
    "vt_gemm_plan_query": (_i, [_i, _i, _i, _i, vp, vp]),
    "vt_gemm_plan_query2": (_i, [_i, _i, _i, _i, vp, vp, vp]),
    "vt_gemm_bf16_resid_splitk": (_i, [vp, _i, vp, _i, vp, _i, vp, _i, _i, _i, _i, vp, _sz, vp]),
    "vt_attn_decode_fused": (_i, [vp, _i, _i, _i, _i, vp, vp, vp, vp, _i, vp, _i, _i, _i, _f, vp, vp, vp, vp]),
    "vt_kv_tiles": (_i, [vp, _i, _i, _i, _i, vp, vp, vp, vp, _i, _i, _i, _i, vp, vp, vp, vp]),
    "vt_attn_temporal": (_i, [vp, vp, _i, _i, _i, _i, vp]),
    "vt_im2col": (_i, [vp, _i, vp, _i, _i, _i, _i, _i, _i, _i, vp]),
    "vt_preprocess": (_i, [vp, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _i, vp, _i, C.c_long,
                           C.c_long, vp]),
    "vt_embed_splice": (_i, [vp, _i, vp, _i, vp, _i, vp, _i, _i, vp, vp]),
    "vt_argmax": (_i, [vp, _i, _i, _i, vp, vp]),
    "vt_decode_feed": (_i, [vp, _i, _i, vp, vp, vp, _i, _i, vp, vp, vp, vp, _i, vp]),
    "vt_cross_entropy": (_i, [vp, _i, _i, _i, vp, _i, vp, vp, vp]),
    "vt_sample_top_p": (_i, [vp, _i, _i, _i, _f, _i, _f, C.c_uint64, C.c_uint64, vp, vp, vp]),
    "vt_projector_workspace_bytes": (_sz, [_i, _i]),
    "vt_projector_forward": (_i, [vp, _i, _i, vp, vp, _i, vp, vp, _i, vp, vp, _sz, vp]),
    "vt_projector_precise_workspace_bytes": (_sz, [_i, _i, _i]),
    "vt_projector_forward_precise": (_i, [vp, vp, _i, _i, vp, vp, _i, vp, vp, _i, vp, vp, vp, _sz, vp]),
    "vt_region_workspace_bytes": (_sz, [_i, _i, _i]),
    "vt_region_forward": (_i, [C.POINTER(VtRegionWeights), vp, vp, vp, _i, _i, _i, vp, vp, vp, vp, _sz, vp]),
    "vt_vit_workspace_bytes": (_sz, [C.POINTER(VtVitModel), _i, _i]),
    "vt_vit_forward": (_i, [C.POINTER(VtVitModel), vp, _i, _i, _i, _i, vp, vp, vp, _sz, vp]),
    "vt_llama_workspace_bytes": (_sz, [C.POINTER(VtLlamaModel), _i, _i, _i, _i]),
    "vt_llama_forward": (_i, [C.POINTER(VtLlamaModel), C.POINTER(VtKvCache), vp, _i, vp, vp, _i, _i, _i, _i, vp, vp,
                              _i, vp, vp, vp, _sz, vp]),
    "vt_profile_begin": (_i, []),
    "vt_profile_end": (_i, [C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vt_probe_mfma": (_i, [vp, vp, vp, _i, vp]),
    "vt_probe_read": (_i, [vp, _sz, _i, vp, vp]),
    "vt_flash_attn_block_order": (_i, [_i, _i, _i, _i, C.POINTER(C.c_int), _i]),
    "vt_mx4_aexp_bytes": (_sz, [_i, _i]),
    "vt_mx4_quant_weights": (_i, [vp, _i, _i, _i, vp, vp, vp]),
    "vt_mx4_quant_lo": (_i, [vp, _i, _i, _i, vp, vp, vp]),
    "vt_rmsnorm_mx": (_i, [vp, vp, vp, vp, vp, vp, _i, _i, _f, vp]),
    "vt_gemm_mx": (_i, [vp, _i, vp, vp, vp, _i, vp, vp, vp, _i, vp, _i, _i, _i, _i, vp]),
    "vt_gemm_mx_swiglu": (_i, [vp, _i, vp, vp, vp, _i, vp, vp, vp, _i, vp, vp, _i, _i, _i, vp]),
    "vt_gemm_mx_gelu": (_i, [vp, _i, vp, vp, vp, _i, vp, vp, vp, vp, _i, vp, vp, _i, _i, _i, _i, vp]),
    "vt_layernorm_mx": (_i, [vp, vp, vp, vp, vp, vp, _i, _i, _f, vp]),
    "vt_gemm_mx_resid": (_i, [vp, _i, vp, vp, vp, _i, vp, vp, vp, _i, _i, _i, _i, vp, _sz, vp]),
    "vt_flash_attn_mx": (_i, [vp, _i, vp, vp, vp, vp, _i, _i, vp, _i, vp, vp, _i, _f, vp]),
}
PROF_CLASSES = ("gemm_tile", "flash_attn", "gemm_skinny", "attn_decode")

_libs = {}          # operand ("bf16" | "fp16" | "abl") -> CDLL
_default_operand = os.environ.get("VITRON_AMD_OPERAND", "bf16")


class VitronHipError(RuntimeError):
    pass


def operand_of(x) -> str:
    """'bf16' | 'fp16' for a torch dtype, a tensor, or one of those strings (anything else raises)."""
    import torch
    if isinstance(x, str):
        if x in LIB_PATHS:
            return x
        raise VitronHipError(f"operand format must be 'bf16' or 'fp16', got {x!r}")
    dt = x.dtype if isinstance(x, torch.Tensor) else x
    if dt == torch.bfloat16:
        return "bf16"
    if dt == torch.float16:
        return "fp16"
    raise VitronHipError(f"vitron_amd computes with bf16 or fp16 operands, got {dt}")


def torch_dtype(operand=None):
    import torch
    return {"bf16": torch.bfloat16, "fp16": torch.float16}[operand_of(operand or _default_operand)]


def default_operand() -> str:
    return _default_operand


def set_default_operand(operand) -> None:
    """Operand format of objects built without an explicit dtype ('bf16' unless VITRON_AMD_OPERAND says otherwise)."""
    global _default_operand
    _default_operand = operand_of(operand)


def _needs_build(path: Path) -> bool:
    if not path.exists():
        return True
    srcs = list((PKG_DIR / "csrc").glob("*.hip")) + list((PKG_DIR / "csrc").glob("*.h")) + list((PKG_DIR / "csrc").glob("*.inc")) + \
        list((PKG_DIR.parent / "include").glob("*.h"))
    if not srcs:
        return False
    return max(p.stat().st_mtime for p in srcs) > path.stat().st_mtime


def _bind(lib):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    return lib


def load(build_if_needed: bool = True, ablations: bool = False, operand=None):
    """Load (building first when stale and hipcc is present) the library for `operand` ('bf16' default / 'fp16', or a torch
    dtype / tensor). Raises if unavailable.
    ablations=True (tools/ only, must be the FIRST load of the process): the -DVT_ABLATIONS build of the same sources,
    libvitron_hip_abl.so, which also carries the timing-ablation kernels and their switches (vitron_amd/build.py); it then
    stands in for the bf16 library."""
    op = operand_of(operand if operand is not None else _default_operand)
    if op in _libs:
        return _libs[op]
    if ablations:
        from . import build as _build
        import torch  # noqa: F401  (same runtime-ordering reason as below)
        _libs["bf16"] = _bind(C.CDLL(str(_build.build(ablations=True))))
        return _libs["bf16"]
    path = LIB_PATHS[op]
    if build_if_needed and _needs_build(path) and os.environ.get("VITRON_AMD_NO_BUILD") != "1":
        from . import build as _build
        try:
            _build.build(operand=op)
        except Exception as e:  # stale binary is still usable; a missing one is fatal below
            if not path.exists():
                raise VitronHipError(f"{path.name} is missing and could not be built: {e}") from e
    if not path.exists():
        raise VitronHipError(f"{path} not found: run `python -m vitron_amd.build` (needs hipcc). "
                             "vitron_amd has no CPU or PyTorch fallback for its operators.")
    # PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's). It must be the
    # one already mapped when libvitron_hip.so resolves libamdhip64.so.7, or the process ends up with two runtimes
    # ("no ROCm-capable device is detected" at the first launch). Importing torch first guarantees that.
    import torch  # noqa: F401
    lib = _bind(C.CDLL(str(path)))
    want = OPERAND_FP16 if op == "fp16" else OPERAND_BF16
    if lib.vt_version() != ABI_VERSION:
        raise VitronHipError(f"{path.name} reports ABI {lib.vt_version()}, this package binds ABI {ABI_VERSION}: rebuild it "
                             "(python -m vitron_amd.build)")
    if lib.vt_operand_format() != want:
        raise VitronHipError(f"{path.name} reports operand format {lib.vt_operand_format()}, expected {want} ({op})")
    _libs[op] = lib
    return lib


def lib_for(x):
    """The library whose operand format is the dtype of tensor / dtype `x`."""
    return load(operand=operand_of(x))


def load_any():
    """A library for operators that touch no 16-bit tensor (arg-max, sampler, cross entropy, planner): whichever is already
    loaded, else the default one."""
    for lib in _libs.values():
        return lib
    return load()


def last_error(lib=None) -> str:
    buf = C.create_string_buffer(512)
    (lib or load_any()).vt_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(status: int, what: str = "", lib=None) -> None:
    """Raise on a non-zero status. `lib`: the handle the call went through (its thread-local message is the one to read);
    without it every loaded library is asked."""
    if status != 0:
        msgs = [last_error(lib)] if lib is not None else [m for m in (last_error(l) for l in _libs.values()) if m]
        raise VitronHipError(f"{what or 'libvitron_hip'} failed (status {status}): {' | '.join(msgs)}")


def profile_begin() -> None:
    if not _libs:
        load()            # nothing loaded yet: profile the default library (never a silent no-op)
    for lib in _libs.values():
        check(lib.vt_profile_begin(), "vt_profile_begin", lib)


def profile_end() -> dict:
    """{class: {'launches', 'ms', 'work'}} for the launches since profile_begin (device-side event timing), summed over the
    loaded libraries."""
    n = len(PROF_CLASSES)
    out = {c: {"launches": 0, "ms": 0.0, "work": 0.0} for c in PROF_CLASSES}
    for lib in _libs.values():
        launches, ms, work = (C.c_int * n)(), (C.c_double * n)(), (C.c_double * n)()
        check(lib.vt_profile_end(launches, ms, work), "vt_profile_end", lib)
        for i, c in enumerate(PROF_CLASSES):
            out[c]["launches"] += launches[i]
            out[c]["ms"] += ms[i]
            out[c]["work"] += work[i]
    return out
