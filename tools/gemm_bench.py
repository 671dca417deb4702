"""Micro-benchmark of the MFMA tile GEMM on the shapes of the C3 workload, per tile configuration, next to
torch.matmul (hipBLASLt) on the same random data as a same-chip reference point. Run on the GPU box:
    python tools/gemm_bench.py [--iters 20] > gpurun_out/gemm_bench.json
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402

SHAPES = [  # (M, N, K, epi, label)
    (5120, 12288, 4096, "BF16", "llm qkv"), (5120, 4096, 4096, "F32_RESID", "llm o_proj"),
    (5120, 22016, 4096, "SWIGLU_BF16", "llm gate_up"), (5120, 4096, 11008, "F32_RESID", "llm down"),
    (4616, 3072, 1024, "BF16", "vit qkv"), (4616, 4096, 1024, "BF16_GELU", "vit fc1"), (4616, 1024, 4096, "F32_RESID", "vit fc2"),
    (4608, 4096, 1024, "BF16_GELU", "proj 1"), (4096, 4096, 4096, "F32_RESID", "o_proj rows 0..4095"),
    (1024, 4096, 4096, "F32_RESID", "o_proj rows 4096.."), (4096, 4096, 11008, "F32_RESID", "down rows 0..4095"),
    (1024, 4096, 11008, "F32_RESID", "down rows 4096.."), (4608, 1024, 4096, "F32_RESID", "vit fc2 4608"), (4608, 3072, 1024, "BF16", "vit qkv 4608"), (1088, 12288, 4096, "BF16", "llm qkv C2"), (577, 3072, 1024, "BF16", "vit qkv image"),
]
CFGS = {"auto": 0, "128x128": 2, "256x128": 3, "256x256": 4, "64x128": 5, "256x256_p8": 6, "256x256_p4": 10, "256x256_rp": 8, "rp_v1": 301, "rp_v2": 302, "rp_v3": 303}


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cfgs", default="128x128,256x128,256x256")
    args = ap.parse_args()
    _lib.load()
    dev = torch.device("cuda:0")
    res = []
    for (M, N, K, epi_name, label) in SHAPES:
        epi = getattr(ops, "EPI_" + epi_name)
        a = (torch.randn((M, K), device=dev)).bfloat16()
        w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
        out = torch.zeros((M, N), device=dev, dtype=torch.float32) if epi_name == "F32_RESID" else None
        flops = 2.0 * M * N * K
        row = {"shape": [M, N, K], "epi": epi_name, "label": label}
        for name in args.cfgs.split(","):
            try:
                ms = timeit(lambda: ops.gemm(a, w, None, epi, out=out, cfg=CFGS[name]), args.iters)
                row[name] = round(flops / ms / 1e9, 1)
            except Exception as e:  # noqa: BLE001
                row[name] = f"error: {e}"
        ms = timeit(lambda: torch.matmul(a, w.t()), args.iters)
        row["torch.matmul(hipBLASLt)"] = round(flops / ms / 1e9, 1)
        res.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
