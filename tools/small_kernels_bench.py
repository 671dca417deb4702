"""The memory-bound passes between the GEMMs, one at a time, at the benchmark's shapes (measurement helper, MI355X):
    kv_tiles   S = 5120, 32 heads x 128, rotary (the decoder's K / V^T page write + q rotation; 252 MB per launch)
    rmsnorm    5120 x 4096 fp32 -> 16 bit (126 MB)
    layernorm  4616 x 1024 fp32 -> 16 bit (the towers; 28 MB)
Inputs rotate over 3 copies (> the 256 MB Infinity Cache for the big ones), 20 launches per window, median of 7 windows.
    python tools/small_kernels_bench.py [abl]        abl: load the -DVT_ABLATIONS test library (VT_KV_TILES_VARIANT etc. are read there)
Prints one JSON line per kernel with us per launch, GB/s, and a checksum of the outputs (to compare variants across processes)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402

abl = len(sys.argv) > 1 and sys.argv[1] == "abl"
_lib.load(ablations=abl)
dev = torch.device("cuda:0")
dt = torch.bfloat16


def timed(fn, n=20, windows=7):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    out = []
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(out)[len(out) // 2]


def csum(*ts):
    return [int(t.view(torch.int16).to(torch.int64).sum().item()) for t in ts]


def kv_tiles_case():
    S, heads, hd = 5120, 32, 128
    D = heads * hd
    g = torch.Generator(device=dev).manual_seed(99)
    src = [torch.randn((S, 3 * D), generator=g, device=dev).to(dt) for _ in range(3)]
    ntile = S // 64
    kt = [torch.zeros((ntile * heads * 64 * hd,), dtype=dt, device=dev) for _ in range(3)]
    vt = [torch.zeros_like(k) for k in kt]
    table = torch.randperm(ntile, generator=torch.Generator().manual_seed(1)).to(torch.int32).to(dev)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    ang = torch.arange(S, dtype=torch.float32)[:, None] * inv[None, :]
    cd, sd_ = ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous()
    pos = torch.arange(S, dtype=torch.int32, device=dev)
    desc = torch.tensor([[0, S, S, 0]], dtype=torch.int32, device=dev)
    x = [s.clone() for s in src]
    ops.kv_tiles(x[0], 0, D, 2 * D, kt[0], vt[0], table, desc, ntile, heads, hd, cd, sd_, pos)
    torch.cuda.synchronize()
    sums = csum(x[0][:, :D].contiguous(), kt[0], vt[0])
    us = timed(lambda i: ops.kv_tiles(x[i % 3], 0, D, 2 * D, kt[i % 3], vt[i % 3], table, desc, ntile, heads, hd, cd, sd_, pos))
    by = 2 * S * 3 * D * 2
    print(json.dumps({"kernel": "kv_tiles", "variant": os.environ.get("VT_KV_TILES_VARIANT", "0"), "us": round(us, 2), "GBps": round(by / us / 1e3, 1), "checksum": sums}), flush=True)


def rmsnorm_case():
    rows, D = 5120, 4096
    x = [torch.randn((rows, D), device=dev) for _ in range(3)]
    w = torch.ones((D,), device=dev)
    us = timed(lambda i: ops.rmsnorm(x[i % 3], w, 1e-5, dtype=dt))
    print(json.dumps({"kernel": "rmsnorm", "us": round(us, 2), "GBps": round(rows * D * 6 / us / 1e3, 1)}), flush=True)


def layernorm_case():
    rows, D = 4616, 1024
    x = [torch.randn((rows, D), device=dev) for _ in range(8)]
    g, b = torch.ones((D,), device=dev), torch.zeros((D,), device=dev)
    y = ops.layernorm(x[0], g, b, 1e-5, dtype=dt)
    us = timed(lambda i: ops.layernorm(x[i % 8], g, b, 1e-5, dtype=dt))
    print(json.dumps({"kernel": "layernorm", "variant": os.environ.get("VT_LN_VARIANT", "0"), "us": round(us, 2), "GBps": round(rows * D * 6 / us / 1e3, 1), "checksum": csum(y)}), flush=True)


for case in (kv_tiles_case, rmsnorm_case, layernorm_case):
    case()
