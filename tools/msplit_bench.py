"""Would the M-split of vt_gemm_launch (whole rounds of 256x256 tiles on the ping-pong kernel + the remaining rows on small
tiles) also pay for the K = 1024 shapes of the ViT / projector (today: K >= 2048 only)? Times `auto`, the whole problem on the
4-phase kernel, and a manual split at `m0` rows. Run on the GPU box: python tools/msplit_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import timeit  # noqa: E402
from vitron_amd import _lib, ops  # noqa: E402

SHAPES = [(4616, 4096, 1024, "BF16_QGELU", 4096, "vit fc1"), (4608, 4096, 1024, "BF16_GELU", 4096, "projector fc1"),
          (4616, 3072, 1024, "BF16", 4096, "vit qkv"), (4616, 1024, 1024, "F32_RESID", 4096, "vit o_proj"),
          (4616, 1024, 4096, "F32_RESID", 4096, "vit fc2")]


def main():
    _lib.load()
    dev = torch.device("cuda:0")
    for (M, N, K, epi_name, m0, label) in SHAPES:
        epi = getattr(ops, "EPI_" + epi_name)
        a = torch.randn((M, K), device=dev).bfloat16()
        w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
        resid = epi_name == "F32_RESID"
        out = torch.zeros((M, N), device=dev, dtype=torch.float32 if resid else torch.bfloat16)
        flops = 2.0 * M * N * K
        row = {"shape": [M, N, K], "epi": epi_name, "label": label}

        def split():
            ops.gemm(a[:m0], w, None, epi, out=out[:m0], cfg=_lib.CFG_256x256_P4)
            ops.gemm(a[m0:], w, None, epi, out=out[m0:], cfg=_lib.CFG_64x128)

        for name, fn in (("auto", lambda: ops.gemm(a, w, None, epi, out=out)),
                         ("p4", lambda: ops.gemm(a, w, None, epi, out=out, cfg=_lib.CFG_256x256_P4)),
                         ("64x128", lambda: ops.gemm(a, w, None, epi, out=out, cfg=_lib.CFG_64x128)),
                         ("128x128", lambda: ops.gemm(a, w, None, epi, out=out, cfg=_lib.CFG_128x128)),
                         (f"p4[:{m0}]+64x128", split)):
            try:
                ms = timeit(fn, 30)
                row[name] = {"us": round(ms * 1e3, 1), "tflops": round(flops / ms / 1e9, 1)}
            except Exception as e:  # noqa: BLE001
                row[name] = f"error: {e}"
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
