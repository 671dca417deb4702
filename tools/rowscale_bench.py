"""What does the per-row factor of vt_gemm_bf16 (consumer side of the folded RMSNorm) cost on the decoder's consumer GEMMs?
Same operands with and without `row_scale`, back-to-back launches. Run on the GPU box: python tools/rowscale_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import timeit  # noqa: E402
from vitron_amd import _lib, ops  # noqa: E402

SHAPES = [(5120, 22016, 4096, "SWIGLU_BF16", "gate_up"), (5120, 12288, 4096, "BF16", "qkv"), (1088, 22016, 4096, "SWIGLU_BF16", "gate_up C2")]


def main():
    _lib.load()
    dev = torch.device("cuda:0")
    for (M, N, K, epi_name, label) in SHAPES:
        epi = getattr(ops, "EPI_" + epi_name)
        a = torch.randn((M, K), device=dev).bfloat16()
        w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
        rs = torch.rand((M,), device=dev) + 0.5
        out = torch.empty((M, N // 2 if epi_name == "SWIGLU_BF16" else N), device=dev, dtype=torch.bfloat16)
        row = {"shape": [M, N, K], "label": label}
        for rep in range(2):
            for name, r in (("plain", None), ("row_scale", rs)):
                ms = timeit(lambda: ops.gemm(a, w, None, epi, out=out, row_scale=r), 20)
                row[f"{name}_{rep}"] = round(ms * 1e3, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
