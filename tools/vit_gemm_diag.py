"""Where the time of the ViT's short-K GEMMs goes (0.25 of the MFMA peak on K = 1024 shapes).

For the four GEMM shapes of a ViT-L/14 layer on the 8-frame 336 px clip (4616 rows), with the weights ROTATED over 8 copies so that
every launch streams its weights from HBM as in the tower (one shape re-run from the Infinity Cache flatters short launches):
  * us per launch of the dispatcher's choice at K / 4, K / 2, K, 2K: fixed cost (launch + prologue + epilogue) vs K-step slope;
  * the same shape through torch.matmul (hipBLASLt, plain bf16 store) as the same-chip reference point;
  * one whole layer's launch sequence (7 GEMMs: temporal qkv / out, spatial qkv / out, fc1, fc2 + their norms are NOT included) back to
    back, ours vs the sum of the isolated launches: what the boundaries between short launches cost.
Run on the GPU box: python tools/vit_gemm_diag.py   (measurement helper, not part of the product path)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402

R = 4616
SHAPES = [("qkv", 3072, 1024, "BF16"), ("fc1", 4096, 1024, "BF16_GELU"), ("o_proj", 1024, 1024, "F32_RESID"), ("fc2", 1024, 4096, "F32_RESID")]
NW = 8


def timeit(fn, iters=40):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def decoder_shapes(dev):
    """The two bf16-store GEMMs of a decoder layer at S = 5120 (qkv, gate/up SwiGLU), weights rotated over 3 copies."""
    for name, N, K, epi_name in (("llm_qkv", 12288, 4096, "BF16"), ("llm_gate_up", 22016, 4096, "SWIGLU_BF16")):
        epi = getattr(ops, "EPI_" + epi_name)
        a = torch.randn((5120, K), device=dev).bfloat16()
        ws = [(torch.randn((N, K), device=dev) * 0.02).bfloat16() for _ in range(3)]
        out = torch.empty((5120, N // 2 if epi_name == "SWIGLU_BF16" else N), device=dev, dtype=torch.bfloat16)
        i = [0]

        def ours():
            i[0] = (i[0] + 1) % 3
            ops.gemm(a, ws[i[0]], None, epi, out=out)
        us = timeit(ours, 60)
        print(json.dumps({"gemm": name, "M": 5120, "N": N, "K": K, "epi": epi_name, "us": round(us, 1),
                          "tflops": round(2.0 * 5120 * N * K / us / 1e6, 1), "cfg": ops.gemm_plan(5120, N, K, epi)[0]}), flush=True)


def main():
    _lib.load(ablations=any(os.environ.get(k) for k in ("VT_W4_EPI_DIRECT", "VT_W4_ABL")))     # A/B switches live in the test library
    dev = torch.device("cuda:0")
    if "--decoder" in sys.argv:
        decoder_shapes(dev)
        return
    layer = []
    for name, N, K0, epi_name in SHAPES:
        epi = getattr(ops, "EPI_" + epi_name)
        row = {"gemm": name, "M": R, "N": N, "epi": epi_name}
        for K in (K0 // 4, K0 // 2, K0, 2 * K0):
            a = torch.randn((R, K), device=dev).bfloat16()
            ws = [(torch.randn((N, K), device=dev) * 0.02).bfloat16() for _ in range(NW)]
            bias = torch.zeros((N,), device=dev)
            out = torch.zeros((R, N), device=dev, dtype=torch.float32) if epi_name == "F32_RESID" else torch.empty((R, N), device=dev, dtype=torch.bfloat16)
            i = [0]

            def ours():
                i[0] = (i[0] + 1) % NW
                ops.gemm(a, ws[i[0]], bias, epi, out=out)

            def lib():
                i[0] = (i[0] + 1) % NW
                torch.matmul(a, ws[i[0]].t())

            us, us_lib = timeit(ours), timeit(lib)
            cfg, rows_first = ops.gemm_plan(R, N, K, epi)
            row[f"K{K}"] = {"us": round(us, 1), "tflops": round(2.0 * R * N * K / us / 1e6, 1), "hipblaslt_us": round(us_lib, 1), "cfg": cfg}
            if K == K0:
                layer.append((a, ws, bias, epi, out))
        k = [row[f"K{K0 // 4}"]["us"], row[f"K{K0 // 2}"]["us"], row[f"K{K0}"]["us"], row[f"K{2 * K0}"]["us"]]
        slope = (k[3] - k[2]) / (K0 / 64)                        # us per 64-wide K step between K and 2K
        row["us_per_k_step"] = round(slope, 3)
        row["fixed_us_at_K"] = round(k[2] - slope * (K0 / 64), 1)
        print(json.dumps(row), flush=True)
    # one layer's GEMM sequence back to back: temporal qkv, temporal out, spatial qkv, spatial out, fc1, fc2
    seq = [layer[0], layer[2], layer[0], layer[2], layer[1], layer[3]]
    j = [0]

    def run_layer():
        j[0] = (j[0] + 1) % NW
        for a, ws, bias, epi, out in seq:
            ops.gemm(a, ws[j[0]], bias, epi, out=out)

    us_layer = timeit(run_layer, 20)
    iso = 0.0
    for a, ws, bias, epi, out in seq:
        jj = [0]

        def one():
            jj[0] = (jj[0] + 1) % NW
            ops.gemm(a, ws[jj[0]], bias, epi, out=out)
        iso += timeit(one, 20)
    print(json.dumps({"layer_gemm_sequence_us": round(us_layer, 1), "sum_of_isolated_us": round(iso, 1),
                      "algorithmic_gflop": round(sum(2.0 * a.shape[0] * ws[0].shape[0] * a.shape[1] for a, ws, *_ in seq) / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
