"""Weight-streaming GEMM (M <= 16) on the decode shapes: LDS-DMA ring kernel (cfg 1, default) vs register-operand MFMA kernel (cfg 9); [us, TB/s].
Prints achieved weight-stream bandwidth (2*N*K bytes / time). Buffers are rotated so weights come from HBM, not cache."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402


def main():
    # SKINNY_ABL=1 VT_SKINNY_VARIANT=<waves * 100 + ring slots>: the test library's other (waves, ring) shapes of the LDS-DMA kernel
    abl = os.environ.get("SKINNY_ABL") == "1"
    _lib.load(ablations=abl)
    dev = torch.device("cuda:0")
    shapes = [("qkv", 12288, 4096, ops.EPI_BF16), ("o_proj", 4096, 4096, ops.EPI_F32_RESID),
              ("gate_up", 22016, 4096, ops.EPI_SWIGLU_BF16), ("down", 4096, 11008, ops.EPI_F32_RESID),
              ("lm_head", 32000, 4096, ops.EPI_F32)]
    Ms = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "4", "16"])]
    for name, N, K, epi in shapes:
        nbuf = max(2, int(600e6 // (N * K * 2)) + 1)    # > 256 MiB MALL
        ws = [(torch.randn((N, K), device=dev) * 0.02).bfloat16() for _ in range(nbuf)]
        for M in Ms:
            a = torch.randn((M, K), device=dev).bfloat16()
            n_out = N // 2 if epi == ops.EPI_SWIGLU_BF16 else N
            out = torch.zeros((M, n_out), device=dev, dtype=torch.float32 if epi in (ops.EPI_F32, ops.EPI_F32_RESID) else torch.bfloat16)
            row = {"shape": name, "M": M, "N": N, "K": K}
            if abl:
                row["variant"] = os.environ.get("VT_SKINNY_VARIANT", "0")
            for label, cfg in ((("auto", 0),) if abl else (("auto", 0), ("dma_ring", 1), ("tile_64x128", 5))):
                for w in ws:
                    ops.gemm(a, w, None, epi, out=out, cfg=cfg)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 5
                e0.record()
                for _ in range(reps):
                    for w in ws:
                        ops.gemm(a, w, None, epi, out=out, cfg=cfg)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / (reps * nbuf)
                row[label] = [round(us, 2), round(N * K * 2 / us / 1e6, 2)]
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
