"""gate/up of the C3 step (5120 x 22016 x 4096, SwiGLU epilogue) on the test library: the dispatcher's kernel against the seamless
persistent form (VT_W4_SEAMLESS=<workgroups>, read once per process; the kernel lived in the test library for this measurement only -- EXPERIMENTS.md round 6 -- and is no longer in the tree: the script documents how the numbers in profiles/r6_gemm_seamless_persistent.jsonl were taken): µs per launch (weights rotated over 4 copies, 7 windows of 12
launches, median) and a checksum of the output.   python tools/gemm_seamless_ab.py [rows=5120]   (exploration helper, MI355X)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402

_lib.load(ablations=True)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
dev = torch.device("cuda:0")
H, I = 4096, 11008
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn((rows, H), device=dev, generator=g).bfloat16()
ws = [(torch.randn((2 * I, H), device=dev, generator=g) * 0.02).bfloat16() for _ in range(4)]
out = ops.gemm(a, ws[0], None, ops.EPI_SWIGLU_BF16)
torch.cuda.synchronize()
cs = int(out.view(torch.int16).to(torch.int64).sum().item())
nan = bool(torch.isnan(out.float()).any().item())
for i in range(4):
    ops.gemm(a, ws[i % 4], None, ops.EPI_SWIGLU_BF16)
torch.cuda.synchronize()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(12):
        ops.gemm(a, ws[i % 4], None, ops.EPI_SWIGLU_BF16)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 12 * 1e3)
us = sorted(ts)[3]
print(json.dumps({"seamless": os.environ.get("VT_W4_SEAMLESS", "0"), "rows": rows, "us": round(us, 1), "pflops": round(2.0 * rows * 2 * I * H / us / 1e9, 3),
                  "checksum": cs, "nan": nan}), flush=True)
