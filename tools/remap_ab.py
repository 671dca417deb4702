"""Four-wave GEMM shapes whose tile grid is whole rounds of the 256 CUs, under VT_W4_NOREMAP of an exploration build of the test library
(0 = one contiguous eighth of the tiles per XCD, 2 = rounds of 8 x 32): us per launch (weights rotated over 3 copies, median of 7 windows) and
an output checksum.   python tools/remap_ab.py   (EXPERIMENTS.md round 6)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402

_lib.load(ablations=True)
dev = torch.device("cuda:0")
H, I = 4096, 11008
res = {"remap": os.environ.get("VT_W4_NOREMAP", "0")}
for name, M, N, K, epi in (("qkv_5120", 5120, 3 * H, H, ops.EPI_BF16), ("qkv_40960", 40960, 3 * H, H, ops.EPI_BF16), ("o_40960", 40960, H, H, ops.EPI_F32_RESID),
                           ("down_40960", 40960, H, I, ops.EPI_F32_RESID), ("qkv_10240", 10240, 3 * H, H, ops.EPI_BF16)):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn((M, K), device=dev, generator=g).bfloat16()
    ws = [(torch.randn((N, K), device=dev, generator=g) * 0.02).bfloat16() for _ in range(3)]
    out = torch.zeros((M, N), device=dev, dtype=torch.float32) if epi == ops.EPI_F32_RESID else None
    y = ops.gemm(a, ws[0], None, epi, out=out)
    cs = int(y.view(torch.int32 if epi == ops.EPI_F32_RESID else torch.int16).to(torch.int64).sum().item())
    f = lambda i: ops.gemm(a, ws[i % 3], None, epi, out=out)  # noqa: E731
    for i in range(3):
        f(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(6):
            f(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 6 * 1e3)
    res[name] = [round(sorted(ts)[3], 1), ops.gemm_plan(M, N, K, epi)[0], cs]
    del a, ws, out, y
print(json.dumps(res), flush=True)
