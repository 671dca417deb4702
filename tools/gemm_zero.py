import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops
from tools.gemm_bench import timeit
_lib.load(); dev = torch.device("cuda:0")
for (M, N, K) in [(4096, 4096, 4096), (5120, 12288, 4096), (8192, 8192, 8192)]:
    for fill in ("randn", "zeros", "ones_small"):
        if fill == "randn":
            a = torch.randn((M, K), device=dev).bfloat16(); w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
        elif fill == "zeros":
            a = torch.zeros((M, K), device=dev).bfloat16(); w = torch.zeros((N, K), device=dev).bfloat16()
        else:
            a = torch.randint(0, 3, (M, K), device=dev).bfloat16(); w = torch.randint(0, 3, (N, K), device=dev).bfloat16()
        r = {}
        for name, cfg in (("p8", 6), ("rp", 8), ("256", 4)):
            ms = timeit(lambda: ops.gemm(a, w, None, ops.EPI_BF16, cfg=cfg), 10)
            r[name] = round(2.0 * M * N * K / ms / 1e9, 1)
        ms = timeit(lambda: torch.matmul(a, w.t()), 10)
        r["hipblaslt"] = round(2.0 * M * N * K / ms / 1e9, 1)
        print(M, N, K, fill, r, flush=True)
