"""The towers' spatial attention alone at the benchmark's shape (8 frames x 577 tokens, 16 heads x 64, not causal): us per launch of
vt_flash_attn (+ the kv_tiles pass beside it) on the test library; VT_FA64_WAVES selects exploration variants of the block shape.
    python tools/vit_attn_probe.py [frames=8]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402

_lib.load(ablations=True)
dev = torch.device("cuda:0")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, heads, hd = 577, 16, 64
D = heads * hd
g = torch.Generator(device=dev).manual_seed(3)
qkv = [torch.randn((F * N, 3 * D), generator=g, device=dev).bfloat16() for _ in range(3)]
tpf = (N + 63) // 64
kt = torch.zeros((F * tpf * heads * 64 * hd,), dtype=torch.bfloat16, device=dev)
vt = torch.zeros_like(kt)
table = torch.arange(F * tpf, dtype=torch.int32, device=dev)
desc = torch.tensor([[f * N, N, N, f * tpf] for f in range(F)], dtype=torch.int32, device=dev)
scale = hd ** -0.5
out = torch.empty((F * N, D), device=dev, dtype=torch.bfloat16)


def timed(fn, n=20, windows=7):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return round(sorted(ts)[len(ts) // 2], 2)


ops.kv_tiles(qkv[0], 0, D, 2 * D, kt, vt, table, desc, tpf, heads, hd)
ops.flash_attn(qkv[0], kt, vt, table, desc, N, heads, hd, False, scale, out=out)
torch.cuda.synchronize()
cs = int(out.view(torch.int16).to(torch.int64).sum().item())
r = {"variant": os.environ.get("VT_FA64_WAVES", "4"), "frames": F,
     "kv_tiles_us": timed(lambda i: ops.kv_tiles(qkv[i % 3], 0, D, 2 * D, kt, vt, table, desc, tpf, heads, hd)),
     "flash_attn_us": timed(lambda i: ops.flash_attn(qkv[i % 3], kt, vt, table, desc, N, heads, hd, False, scale, out=out)), "checksum": cs}
print(json.dumps(r), flush=True)
