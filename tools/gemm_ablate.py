"""Timing ablations of the 8-phase GEMM (results are garbage by construction): which resource paces the K loop?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops
from tools.gemm_bench import timeit
_lib.load(ablations=True); dev = torch.device("cuda:0")
names = {10: "P4 full", 111: "P4 no ds_read", 112: "P4 no DMA", 113: "P4 no ds_read, no DMA", 114: "P4 no MFMA", 115: "P4 no MFMA, no ds_read", 116: "P4 no MFMA, no DMA", 117: "P4 barriers only", 6: "full", 101: "no ds_read", 102: "no DMA", 103: "no ds_read, no DMA", 104: "no MFMA", 105: "no MFMA, no ds_read", 106: "no MFMA, no DMA", 107: "barriers only"}
for (M, N, K) in [(4096, 4096, 4096)]:
    a = torch.randn((M, K), device=dev).bfloat16(); w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
    for cfg, nm in names.items():
        ms = timeit(lambda: ops.gemm(a, w, None, ops.EPI_BF16, cfg=cfg), 10)
        print(f"{M}x{N}x{K} {nm:22s} {ms*1e3:8.1f} us  ({2.0*M*N*K/ms/1e9:7.1f} TF-equivalent)", flush=True)
