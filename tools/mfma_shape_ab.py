import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops
_lib.load(ablations=True)
dev = torch.device("cuda:0")
shapes = [(5120, 12288, 4096), (5120, 22016, 4096), (5120, 4096, 11008)]
for M, N, K in shapes:
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn((M, K), device=dev, generator=g).bfloat16()
    ws = [(torch.randn((N, K), device=dev, generator=g) * 0.02).bfloat16() for _ in range(3)]
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    f = lambda i: ops.gemm(a, ws[i % 3], None, ops.EPI_BF16, out=out, cfg=13)
    for i in range(4): f(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12): f(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 12 * 1e3)
    us = sorted(ts)[3]
    print(json.dumps({"abl": os.environ.get("VT_W4_ABL", "0"), "shape": [M, N, K], "us": round(us, 1), "pflops": round(2.0 * M * N * K / us / 1e9, 3)}), flush=True)
