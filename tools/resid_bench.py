"""Residual-epilogue GEMMs of the decoder on the 4-phase kernel (whole rounds of tiles), back-to-back launches.
python tools/resid_bench.py"""
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
    for (M, N, K) in ((4096, 4096, 4096), (4096, 4096, 11008)):
        a = torch.randn((M, K), device=dev).bfloat16()
        w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
        out = torch.zeros((M, N), device=dev, dtype=torch.float32)
        r = [round(timeit(lambda: ops.gemm(a, w, None, ops.EPI_F32_RESID, out=out, cfg=_lib.CFG_256x256_P4), 30) * 1e3, 1) for _ in range(3)]
        print(json.dumps({"lib": os.path.dirname(os.path.abspath(_lib.__file__)), "shape": [M, N, K], "us": r}), flush=True)


if __name__ == "__main__":
    main()
