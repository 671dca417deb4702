"""Flash attention of this repo against torch.nn.functional.scaled_dot_product_attention (PyTorch-ROCm's own flash / AOTriton
backend) on the prefill shape of the benchmark (S = 5120, 32 heads, head dim 128, causal, bf16), 1-s alternating windows.
A reference point only (like tools/gemm_sustained.py for the GEMMs); torch's layout is [B, H, S, D] contiguous, ours reads the
fused-QKV rows and the paged K / V^T tiles.   python tools/attn_vs_sdpa.py [S]"""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402


def run(fn, seconds):
    fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    return (time.perf_counter() - t0) / n


def main():
    _lib.load()
    dev = torch.device("cuda:0")
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
    heads, hd = 32, 128
    D = heads * hd
    qkv = torch.randn((S, 3 * D), device=dev).bfloat16()
    nt = (S + 63) // 64
    desc = torch.tensor([[0, S, S, 0]], dtype=torch.int32, device=dev)
    table = torch.arange(nt, dtype=torch.int32, device=dev)
    kt = torch.zeros(nt * heads * 64 * hd, dtype=torch.bfloat16, device=dev)
    vt = torch.zeros_like(kt)
    ops.kv_tiles(qkv, 0, D, 2 * D, kt, vt, table, desc, nt, heads, hd)
    out = torch.empty((S, D), dtype=torch.bfloat16, device=dev)
    q = qkv[:, :D].view(S, heads, hd).transpose(0, 1).unsqueeze(0).contiguous()
    k = qkv[:, D:2 * D].view(S, heads, hd).transpose(0, 1).unsqueeze(0).contiguous()
    v = qkv[:, 2 * D:].view(S, heads, hd).transpose(0, 1).unsqueeze(0).contiguous()
    flops = heads * 4 * hd * (S * (S + 1) / 2)
    ours = lambda: ops.flash_attn(qkv, kt, vt, table, desc, S, heads, hd, True, 1 / math.sqrt(hd), out=out)   # noqa: E731
    sdpa = lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True)                  # noqa: E731
    ref = sdpa()[0].transpose(0, 1).reshape(S, D).float()
    ours()
    err = ((out.float() - ref).norm() / ref.norm()).item()
    for rep in range(3):
        row = {"S": S, "heads": heads, "hd": hd, "rep": rep, "rel_l2_ours_vs_sdpa": round(err, 5)}
        for name, fn in (("ours", ours), ("torch_sdpa", sdpa)):
            s = run(fn, 1.0)
            row[name] = {"us": round(s * 1e6, 1), "tflops": round(flops / s / 1e12, 1)}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
