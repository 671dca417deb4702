import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops
from tools.gemm_bench import timeit
_lib.load(); dev = torch.device("cuda:0")
M = N = 4096
for K in (256, 512, 1024, 2048, 4096, 8192):
    a = torch.randn((M, K), device=dev).bfloat16(); w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
    out32 = torch.zeros((M, N), device=dev, dtype=torch.float32)
    r = {}
    for nm, cfg in (("p8", 6), ("rp", 8), ("256", 4), ("128", 2)):
        r[nm + "_bf16"] = round(timeit(lambda: ops.gemm(a, w, None, ops.EPI_BF16, cfg=cfg), 20) * 1e3, 1)
    r["p8_resid"] = round(timeit(lambda: ops.gemm(a, w, None, ops.EPI_F32_RESID, out=out32, cfg=6), 20) * 1e3, 1)
    r["hipblaslt"] = round(timeit(lambda: torch.matmul(a, w.t()), 20) * 1e3, 1)
    print(K, r, flush=True)
