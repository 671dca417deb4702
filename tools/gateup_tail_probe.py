import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops
_lib.load()
dev = torch.device("cuda:0")
H, I = 4096, 11008
def timed(fn, n=12, windows=7):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return round(sorted(ts)[len(ts)//2], 1)
for M in (768, 1088):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn((M, H), device=dev, generator=g).bfloat16()
    ws = [(torch.randn((2 * I, H), device=dev, generator=g) * 0.02).bfloat16() for _ in range(4)]
    r = {"M": M, "plan": ops.gemm_plan_cols(M, 2 * I, H, ops.EPI_SWIGLU_BF16)}
    r["auto_us"] = timed(lambda i: ops.gemm(a, ws[i % 4], None, ops.EPI_SWIGLU_BF16))
    for ntail in (256, 512):
        n1 = 2 * I - ntail
        r[f"main_{n1}_us"] = timed(lambda i: ops.gemm(a, ws[i % 4][:n1], None, ops.EPI_SWIGLU_BF16, cfg=13))
        r[f"tail_{ntail}_auto_us"] = timed(lambda i: ops.gemm(a, ws[i % 4][n1:], None, ops.EPI_SWIGLU_BF16))
        outf = torch.zeros((M, ntail), device=dev, dtype=torch.float32)
        for ks in (8, 16):
            part = torch.empty((ks * M * ntail,), device=dev, dtype=torch.float32)
            r[f"tail_{ntail}_splitk{ks}_us"] = timed(lambda i: ops.gemm_resid_splitk(a, ws[i % 4][n1:], outf, None, ks, part))
    print(json.dumps(r), flush=True)
