"""vt_gemm_mx (16-bit product + MX-FP4 product in one launch: precise level 3) against vt_gemm_bf16 AUTO on the decoder's prefill shapes:
weights rotated over 4 copies (cold, as in the step), arms alternated in rounds, median of the rounds.
    python tools/gemm_mx_bench.py [rows=5120] [fp16|bf16] [iters=20]      -> one JSON line per shape
Measurement helper (MI355X), not part of the product path."""
import json
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402
if __import__("os").environ.get("VT_GEMM_MX_BENCH_ABL") == "1":     # the test library (bf16), for its exploration knobs
    _lib.load(ablations=True)

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[2] if len(sys.argv) > 2 else "fp16"]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
H, I = 4096, 11008
SHAPES = [("qkv", 3 * H, H, ops.EPI_BF16), ("o_proj", H, H, ops.EPI_F32_RESID), ("gate_up", 2 * I, H, ops.EPI_SWIGLU_BF16),
          ("down", H, I, ops.EPI_F32_RESID)]
for name, N, K, epi in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    a32 = torch.randn((rows, K), device=dev, generator=g)
    a = a32.to(dt)
    lo = (a32 - a.float()).to(dt)
    a4, aexp = ops.mx4_quant_lo(lo)
    ws = [(torch.randn((N, K), device=dev, generator=g) * 0.02).to(dt) for _ in range(4)]
    w4s = [ops.mx4_quant_weights(w) for w in ws]
    out = torch.zeros((rows, N), device=dev, dtype=torch.float32) if epi == ops.EPI_F32_RESID else None

    def run(arm, i):
        w = ws[i % 4]
        if arm == "base":
            return ops.gemm(a, w, None, epi, out=out)
        return ops.gemm_mx(a, a4, aexp, w, w4s[i % 4][0], w4s[i % 4][1], None, epi, out=out)
    res = {"base": [], "mx": []}
    for r in range(5):
        for arm in ("base", "mx"):
            for i in range(3):
                run(arm, i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                run(arm, i)
            e1.record()
            torch.cuda.synchronize()
            res[arm].append(e0.elapsed_time(e1) / iters * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    fl = 2.0 * rows * N * K
    print(json.dumps({"shape": name, "M": rows, "N": N, "K": K, "dtype": str(dt).split(".")[-1], "base_us": round(med["base"], 1),
                      "mx_us": round(med["mx"], 1), "ratio": round(med["mx"] / med["base"], 3),
                      "base_pflops": round(fl / med["base"] * 1e-9, 3), "mx_pflops_16bit_work": round(fl / med["mx"] * 1e-9, 3),
                      "plan": ops.gemm_plan(rows, N, K, epi)}), flush=True)
