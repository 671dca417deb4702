"""Build libvitron_hip.so (gfx950) in-tree with hipcc.

The shared library is the product: every compute call of vitron_amd goes through it (ctypes), and there is
no fallback when it is missing. hipcc cross-compiles for gfx950 without a GPU, so the build runs anywhere the
ROCm toolchain is installed. Objects are cached under vitron_amd/csrc/build/ keyed on source mtimes.
"""
from __future__ import annotations

import contextlib
import fcntl
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
INCLUDE = PKG_DIR.parent / "include"
LIB_PATH = PKG_DIR / "libvitron_hip.so"
# the SAME sources compiled with -DVT_OPERAND_F16=1: every GEMM / attention operand in IEEE fp16 instead of bf16 (vt_common.h,
# include/vitron_hip.h "OPERAND FORMAT"); same ABI, same symbol names. A product library like the one above.
F16_LIB_PATH = PKG_DIR / "libvitron_hip_f16.so"
# the SAME sources compiled with -DVT_ABLATIONS: timing-ablation / A-B variants of the GEMM and attention main loops (garbage
# results by construction) and their environment switches. Only tools/ load it (vitron_amd._lib.load(ablations=True)); the
# product library above contains none of that code.
ABL_LIB_PATH = PKG_DIR / "libvitron_hip_abl.so"
SOURCES = ["vt_api.hip", "vt_gemm.hip", "vt_gemm8.hip", "vt_mx4.hip", "vt_norm.hip", "vt_attn.hip", "vt_attn_w4.hip", "vt_vit.hip", "vt_region.hip", "vt_llama.hip", "vt_preproc.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-Wno-unused-but-set-variable", "-Wno-unused-variable"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: vitron_amd needs the ROCm toolchain to build libvitron_hip.so")
    return exe


def _deps_mtime() -> float:
    hdrs = list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + list(INCLUDE.glob("*.h"))
    return max(p.stat().st_mtime for p in hdrs)


# per-file flags. vt_attn_w4.hip: its hand-placed VALU stream must not be SLP-packed into v_pk_*_f32 (an anti-lever beside MFMAs)
FILE_FLAGS = {"vt_attn_w4.hip": ["-fno-slp-vectorize"]}


def _compile_one(src: Path, obj: Path, verbose: bool, extra=()) -> None:
    tmp = obj.with_suffix(f".tmp{os.getpid()}.o")
    cmd = [_hipcc(), *FLAGS, *FILE_FLAGS.get(src.name, []), *extra, "-c", str(src), "-o", str(tmp)]
    if verbose:
        print("[vitron_amd.build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        with contextlib.suppress(OSError):
            tmp.unlink()
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, obj)          # a reader never sees a half-written object
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)


@contextlib.contextmanager
def _build_lock(bdir: Path):
    """One builder at a time across processes (torchrun starts one rank per GPU, each of which may find a stale library):
    the others wait here, then see up-to-date objects and return without compiling."""
    with open(bdir / ".lock", "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def build(force: bool = False, verbose: bool = False, ablations: bool = False, operand: str = "bf16") -> Path:
    """Compile every HIP source for gfx950 and link libvitron_hip.so (operand="fp16": libvitron_hip_f16.so, the fp16-operand
    build; ablations=True: the -DVT_ABLATIONS test library libvitron_hip_abl.so instead). Returns the library path."""
    if operand not in ("bf16", "fp16"):
        raise ValueError(f"operand must be 'bf16' or 'fp16', got {operand!r}")
    if ablations and operand != "bf16":
        raise ValueError("the ablation library exists for bf16 operands only")
    bdir = CSRC / ("build_abl" if ablations else "build_f16" if operand == "fp16" else "build")
    bdir.mkdir(exist_ok=True)
    with _build_lock(bdir):
        return _build_locked(bdir, force, verbose, ablations, operand)


def build_all(force: bool = False, verbose: bool = False):
    """Both product libraries (bf16 and fp16 operands), compiled concurrently."""
    with ThreadPoolExecutor(max_workers=2) as ex:
        futs = [ex.submit(build, force, verbose, False, op) for op in ("bf16", "fp16")]
        return [f.result() for f in futs]


def lib_path(operand: str = "bf16") -> Path:
    return F16_LIB_PATH if operand == "fp16" else LIB_PATH


def _build_locked(bdir: Path, force: bool, verbose: bool, ablations: bool = False, operand: str = "bf16") -> Path:
    LIB_PATH = ABL_LIB_PATH if ablations else lib_path(operand)
    extra = ("-DVT_ABLATIONS",) if ablations else ("-DVT_OPERAND_F16=1",) if operand == "fp16" else ()
    hdr_m = _deps_mtime()
    todo = []
    objs = []
    for name in SOURCES:
        src = CSRC / name
        if not src.exists():
            raise RuntimeError(f"missing source {src}")
        obj = bdir / (src.stem + ".o")
        objs.append(obj)
        if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, hdr_m):
            todo.append((src, obj))
    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(lambda so: _compile_one(so[0], so[1], verbose, extra), todo))
    if todo or not LIB_PATH.exists() or any(o.stat().st_mtime > LIB_PATH.stat().st_mtime for o in objs):
        tmp = LIB_PATH.with_suffix(f".tmp{os.getpid()}.so")
        # -Bsymbolic: the library's own references bind to its own definitions, so the bf16 and the fp16 build (same symbol names)
        # can live in one process
        cmd = [_hipcc(), "-shared", "-fPIC", "-Wl,-Bsymbolic", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(tmp)]
        if verbose:
            print("[vitron_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            with contextlib.suppress(OSError):
                tmp.unlink()
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        os.replace(tmp, LIB_PATH)   # processes that already mapped the old file keep it; new loads see a complete library
    return LIB_PATH


if __name__ == "__main__":
    if "--ablations" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, ablations=True))
    else:
        for p in build_all(force="--force" in sys.argv, verbose=True):
            print(p)
