"""How much of the GEMM's time is set by the operand VALUES (sustained clock under the power limit) rather than by the
instruction schedule: the same launch on N(0,1)·0.02 operands, on all-zero operands and on a constant. python tools/gemm_data_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import timeit  # noqa: E402
from vitron_amd import _lib, ops  # noqa: E402


def main():
    _lib.load()
    dev = torch.device("cuda:0")
    M, N, K = 5120, 12288, 4096
    flops = 2.0 * M * N * K
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    for name, mk in (("randn", lambda r, c, s: (torch.randn((r, c), device=dev) * s).bfloat16()),
                     ("zeros", lambda r, c, s: torch.zeros((r, c), device=dev, dtype=torch.bfloat16)),
                     ("ones", lambda r, c, s: torch.full((r, c), 1.0, device=dev, dtype=torch.bfloat16)),
                     ("randn", lambda r, c, s: (torch.randn((r, c), device=dev) * s).bfloat16())):
        a, w = mk(M, K, 1.0), mk(N, K, 0.02)
        row = {"data": name}
        for cfg_name, cfg in (("p4", _lib.CFG_256x256_P4), ("128x128", _lib.CFG_128x128)):
            ms = timeit(lambda: ops.gemm(a, w, None, ops.EPI_BF16, out=out, cfg=cfg), 20)
            row[cfg_name] = {"us": round(ms * 1e3, 1), "tflops": round(flops / ms / 1e9, 1)}
        ms = timeit(lambda: torch.matmul(a, w.t()), 20)
        row["hipBLASLt"] = {"us": round(ms * 1e3, 1), "tflops": round(flops / ms / 1e9, 1)}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
