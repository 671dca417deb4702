"""Micro-benchmark of the flash attention kernel on the C3 shapes (LLM causal prefill S=5120, ViT spatial N=577)."""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def run(S, heads, hd, nseq, causal):
    dev = torch.device("cuda:0")
    D = heads * hd
    dt = torch.float16 if os.environ.get("ATTN_BENCH_DTYPE") == "fp16" else torch.bfloat16
    qkv = torch.randn((nseq * S, 3 * D), device=dev).to(dt)
    if os.environ.get("ATTN_BENCH_DATA") == "zeros":      # operand values change the sustained clock on this part
        qkv.zero_()
    elif os.environ.get("ATTN_BENCH_DATA") == "small":
        qkv.mul_(0.05)
    nt = (S + 63) // 64
    desc = torch.tensor([[i * S, S, S, i * nt] for i in range(nseq)], dtype=torch.int32, device=dev)
    table = torch.arange(nseq * nt, dtype=torch.int32, device=dev)
    kt = torch.zeros(nseq * nt * heads * 64 * hd, dtype=dt, device=dev)
    vt = torch.zeros_like(kt)
    ops.kv_tiles(qkv, 0, D, 2 * D, kt, vt, table, desc, nt, heads, hd)
    out = torch.empty((nseq * S, D), dtype=dt, device=dev)
    ms = timeit(lambda: ops.flash_attn(qkv, kt, vt, table, desc, S, heads, hd, causal, 1 / math.sqrt(hd), out=out))
    flops = nseq * heads * 4 * hd * (S * (S + 1) / 2 if causal else S * S)
    ms_kv = timeit(lambda: ops.kv_tiles(qkv, 0, D, 2 * D, kt, vt, table, desc, nt, heads, hd))
    print(json.dumps({"S": S, "heads": heads, "hd": hd, "nseq": nseq, "causal": causal, "ms": round(ms, 4),
                      "tflops": round(flops / ms / 1e9, 1), "kv_tiles_ms": round(ms_kv, 4)}), flush=True)


if __name__ == "__main__" and os.environ.get("VT_W4_PROF"):
    # phase clocks of the persistent attention kernel (test library): one workgroup's waves, summed over its blocks, in shader cycles
    import ctypes
    lib = _lib.load(ablations=True)
    names = ["seam", "set-up", "wait operands", "barrier", "hand-over+reset", "prologue stages", "steady state", "tail", "barrier", "epilogue"]
    for nseq, kern in ((1, 2), (1, 4), (8, 2), (8, 4)):
        lib.vt_flash_attn_select(kern)
        lib.vt_debug_w4_prof(None, 1)
        run(5120, 32, 128, nseq, True)
        buf = (ctypes.c_ulonglong * 64)()
        lib.vt_debug_w4_prof(buf, 0)
        for w in range(4):
            v = buf[16 * w:16 * w + 16]
            calls = 12                      # timeit: 2 warm-up + 10 timed launches of flash_attn (kv_tiles launches follow but do not touch the counters)
            tot = sum(v[:10])
            print(f"kernel {kern} nseq {nseq} wave {w}: blocks/launch {v[10] / calls:.1f} tiles/launch {v[11] / calls:.0f} warm seams {v[12] / calls:.1f}; cycles per block: "
                  + ", ".join(f"{n} {v[i] / max(v[10], 1):.0f}" for i, n in enumerate(names)) + f"; total/launch {tot / calls:.0f}", flush=True)
    sys.exit(0)
if __name__ == "__main__" and os.environ.get("VT_W4_ABL"):
    # timing ablations of the one-wave-per-SIMD kernel's placed loop (test library; results are garbage): VT_W4_ABL=<bits>
    _lib.load(ablations=True)
    _lib.load().vt_flash_attn_select(2)
    run(5120, 32, 128, 1, True)
    run(5120, 32, 128, 8, True)
    sys.exit(0)
if __name__ == "__main__" and os.environ.get("VT_W4_ORDER_FIXED"):
    # cost model of the planned dispatch order (test library): VT_W4_ORDER_FIXED=<cycles per block outside the tile loop>
    _lib.load(ablations=True).vt_flash_attn_select(2)
    for _ in range(12):                         # (the first windows of a fresh process run at a lower clock: read the last ones)
        run(5120, 32, 128, 1, True)
    sys.exit(0)
if __name__ == "__main__" and os.environ.get("ATTN_BENCH_KERNELS"):
    # A/B of the head_dim-128 prefill kernels (vt_flash_attn_select): ATTN_BENCH_KERNELS=1,2,3 python tools/attn_bench.py
    _lib.load()
    for rep in range(2):
        for k in [int(x) for x in os.environ["ATTN_BENCH_KERNELS"].split(",")]:
            ops.flash_attn_select(k)
            print(f"# kernel {k} (pass {rep})", flush=True)
            run(5120, 32, 128, 1, True)
            run(2048, 32, 128, 4, True)
            run(5120, 32, 128, 8, True)
    ops.flash_attn_select(0)
    sys.exit(0)
if __name__ == "__main__" and os.environ.get("ATTN_BENCH_SHAPES"):
    # A/B of the head_dim-128 prefill kernels on chosen sequence lengths: ATTN_BENCH_SHAPES=768,1088,2560 ATTN_BENCH_SEL=1,2,5 (round 5:
    # the reference-native 224 px shapes S = 768 / 2560 against the automatic choice's threshold)
    _lib.load()
    for rep in range(2):
        for k in [int(x) for x in os.environ.get("ATTN_BENCH_SEL", "1,2,5").split(",")]:
            ops.flash_attn_select(k)
            print(f"# kernel {k} (pass {rep})", flush=True)
            for S_ in [int(x) for x in os.environ["ATTN_BENCH_SHAPES"].split(",")]:
                run(S_, 32, 128, 1, True)
    ops.flash_attn_select(0)
    sys.exit(0)
if __name__ == "__main__" and os.environ.get("ATTN_BENCH_ONLY_C3"):
    _lib.load(ablations=True)
    run(5120, 32, 128, 1, True)
    sys.exit(0)
if __name__ == "__main__":
    _lib.load(ablations=any(os.environ.get(k) for k in ("VT_FLASH_ABL", "VT_FLASH_QBLK256")))   # the switches exist only in the test library
    run(5120, 32, 128, 1, True)
    run(1088, 32, 128, 1, True)
    run(577, 16, 64, 8, False)
    run(2048, 32, 128, 4, True)
    run(5120, 32, 128, 8, True)         # C4: eight clips packed into one pass
