import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops
_lib.load(ablations=True)
dev = torch.device("cuda:0")
H, I = 4096, 11008
res = {"group_m": os.environ.get("VT_W4_GROUP_M", "8(default)")}
for name, M, N, K, epi in (("gate_up", 5120, 2 * I, H, ops.EPI_SWIGLU_BF16), ("qkv", 5120, 3 * H, H, ops.EPI_BF16)):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn((M, K), device=dev, generator=g).bfloat16()
    ws = [(torch.randn((N, K), device=dev, generator=g) * 0.02).bfloat16() for _ in range(4)]
    out = ops.gemm(a, ws[0], None, epi)
    cs = int(out.view(torch.int16).to(torch.int64).sum().item())
    f = lambda i: ops.gemm(a, ws[i % 4], None, epi)
    for i in range(4): f(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12): f(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 12 * 1e3)
    res[name] = [round(sorted(ts)[3], 1), cs]
print(json.dumps(res), flush=True)
