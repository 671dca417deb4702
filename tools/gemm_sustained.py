"""Sustained (>= 1 s per measurement, alternating) throughput of the dispatcher's choice (AUTO: four-wave kernel on 256- or 320-row
tiles), the 4-phase ping-pong GEMM and torch.matmul (hipBLASLt) on decoder shapes (bf16 store epilogue on all three): 8-ms micro-benchmarks ride on the thermal / power state the previous kernel left behind, so
kernel-vs-kernel comparisons are made on windows long enough for the power limit to settle. python tools/gemm_sustained.py [M N K | all]"""
import json
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
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    return (time.perf_counter() - t0) / n


def main():
    _lib.load()
    dev = torch.device("cuda:0")
    if len(sys.argv) > 1 and sys.argv[1] == "all":   # the four big GEMM shapes of the benchmark step
        shapes = [(5120, 12288, 4096), (5120, 22016, 4096), (5120, 4096, 11008), (5120, 4096, 4096)]
    else:
        shapes = [(5120, 12288, 4096)] if len(sys.argv) < 4 else [tuple(int(x) for x in sys.argv[1:4])]
    secs = float(os.environ.get("GEMM_SUSTAINED_SECONDS", "1.5"))
    for (M, N, K) in shapes:
        a = torch.randn((M, K), device=dev).bfloat16()
        w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
        out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
        flops = 2.0 * M * N * K
        for rep in range(3):
            row = {"shape": [M, N, K], "rep": rep, "plan": list(ops.gemm_plan(M, N, K, ops.EPI_BF16))}
            for name, fn in (("auto", lambda: ops.gemm(a, w, None, ops.EPI_BF16, out=out)),
                             ("p4", lambda: ops.gemm(a, w, None, ops.EPI_BF16, out=out, cfg=_lib.CFG_256x256_P4)),
                             ("hipBLASLt", lambda: torch.matmul(a, w.t(), out=out))):
                s = run(fn, secs)
                row[name] = {"us": round(s * 1e6, 1), "tflops": round(flops / s / 1e12, 1)}
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
