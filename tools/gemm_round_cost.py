"""Fixed cost of one ROUND of tiles of the four-wave GEMM (launch + prologue + epilogue, nothing of it overlapped with the matrix
pipe: one workgroup per CU) against the K-step slope, on decoder shapes that fill exactly one round of the 256 CUs:
    5120 x 4096 (320-row tiles: 16 x 16 = 256), 4096 x 4096 (256-row tiles), 3584 x 4096 (224-row tiles)
for the three epilogue families (bf16 store through LDS, SwiGLU, fp32 residual read-modify-write), K = 256 .. 8192, weights rotated.
The intercept of us(K) is what a persistent kernel that overlaps tile i's stores with tile i+1's first loads could win per round.
Run on the GPU box: python tools/gemm_round_cost.py   (measurement helper, not part of the product path)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402
from tools.vit_gemm_diag import timeit  # noqa: E402


def main():
    _lib.load()
    dev = torch.device("cuda:0")
    for M in (5120, 4096, 3584):
        for epi_name, N in (("BF16", 4096), ("SWIGLU_BF16", 8192), ("F32_RESID", 4096)):
            epi = getattr(ops, "EPI_" + epi_name)
            pts = []
            for K in (256, 512, 1024, 2048, 4096, 8192):
                a = torch.randn((M, K), device=dev).bfloat16()
                ws = [(torch.randn((N, K), device=dev) * 0.02).bfloat16() for _ in range(4)]
                out = (torch.zeros((M, N), device=dev, dtype=torch.float32) if epi_name == "F32_RESID"
                       else torch.empty((M, N // 2 if epi_name == "SWIGLU_BF16" else N), device=dev, dtype=torch.bfloat16))
                i = [0]

                def ours():
                    i[0] = (i[0] + 1) % 4
                    ops.gemm(a, ws[i[0]], None, epi, out=out)
                pts.append((K // 64, timeit(ours, 40)))
                cfg = ops.gemm_plan(M, N, K, epi)[0]
            ks, us = np.array([p[0] for p in pts], float), np.array([p[1] for p in pts], float)
            slope, icpt = np.polyfit(ks[2:], us[2:], 1)
            rounds = 2 if epi_name == "SWIGLU_BF16" else 1
            print(json.dumps({"M": M, "N": N, "epi": epi_name, "cfg": cfg, "rounds": rounds,
                              "us_by_ksteps": {int(k): round(float(u), 1) for k, u in pts},
                              "us_per_kstep_per_round": round(float(slope) / rounds, 3),
                              "fixed_us_per_round": round(float(icpt) / rounds, 1)}), flush=True)


if __name__ == "__main__":
    main()
