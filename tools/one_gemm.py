"""Run one GEMM shape/config a few times (for rocprofv3 --pmc runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402

M, N, K, cfg = (int(x) for x in sys.argv[1:5])
epi = getattr(ops, "EPI_" + (sys.argv[5] if len(sys.argv) > 5 else "BF16"))
_lib.load()
dev = torch.device("cuda:0")
a = torch.randn((M, K), device=dev).bfloat16()
w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
out = torch.zeros((M, N), device=dev, dtype=torch.float32) if epi == ops.EPI_F32_RESID else None
for _ in range(5):
    ops.gemm(a, w, None, epi, out=out, cfg=cfg)
torch.cuda.synchronize()
