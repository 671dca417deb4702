"""Residual GEMMs with a small 256x256 grid: plain dispatcher vs the two-pass split-K (ksplit 2..8)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import timeit  # noqa: E402
from vitron_amd import _lib, ops  # noqa: E402

_lib.load()
dev = torch.device("cuda:0")
for (M, N, K) in [(1088, 4096, 4096), (1088, 4096, 11008), (1024, 4096, 4096), (1024, 4096, 11008), (577, 1024, 4096),
                  (577, 1024, 1024), (4616, 1024, 4096), (4616, 1024, 1024), (300, 4096, 4096), (300, 4096, 11008)]:
    a = torch.randn((M, K), device=dev).bfloat16()
    w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
    out = torch.zeros((M, N), device=dev)
    part = torch.empty((8, M, N), device=dev)
    f = 2.0 * M * N * K
    row = f"{M}x{N}x{K}: plain {timeit(lambda: ops.gemm(a, w, None, ops.EPI_F32_RESID, out=out), 20) * 1e3:7.1f} us"
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    for ks in (2, 3, 4, 8):
        if (K >> 7) // ks < 2 or tiles * ks > 512:
            continue
        t = timeit(lambda: ops.gemm_resid_splitk(a, w, out, None, ks, part), 20)
        row += f" | ks={ks}: {t * 1e3:7.1f} us ({f / t / 1e9:5.0f} TF)"
    print(row, flush=True)
