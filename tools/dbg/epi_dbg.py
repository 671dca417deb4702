import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vitron_amd import _lib, ops
from tests.util import randn
_lib.load(); dev = torch.device("cuda:0")
M, N, K = 300, 384, 256
a, w, b = randn((M, K), 41), randn((N, K), 42, 0.05), randn((N,), 43)
ad, wd = a.to(dev).bfloat16(), w.to(dev).bfloat16()
for epi, nm in ((ops.EPI_BF16, "bf16"), (ops.EPI_BF16_RELU, "relu"), (ops.EPI_BF16_QGELU, "qgelu")):
    for cfg in (13, 14):
        got = ops.gemm(ad, wd, b.to(dev), epi, cfg=cfg).float().cpu()
        ref = ops.gemm(ad, wd, b.to(dev), epi, cfg=10).float().cpu()
        bad = (got != ref)
        rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
        print(nm, cfg, "mismatch", int(bad.sum()), "rows", rows[:12], "..", rows[-4:], "cols", cols[:12], "..", cols[-4:])
        if bad.any():
            i = bad.nonzero()[:6]
            for r, c in i.tolist():
                print("   ", r, c, float(got[r, c]), float(ref[r, c]))
