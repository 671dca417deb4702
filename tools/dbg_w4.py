import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops
_lib.load(); _lib.load(operand="fp16")
dev = torch.device("cuda:0")
hd, heads, L = 128, 2, 700
D = heads * hd
g = torch.Generator().manual_seed(5)
qkv = torch.randn((L, 3 * D), generator=g)
mult = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
qkv[:, :2 * D] *= mult
qd = qkv.to(dev).bfloat16()
npages = (L + 63) // 64
kt = torch.zeros(npages * heads * 64 * hd, dtype=torch.bfloat16, device=dev)
vt = torch.zeros(npages * heads * 64 * hd, dtype=torch.float16, device=dev)
table, desc = torch.arange(npages, dtype=torch.int32, device=dev), torch.tensor([[0, L, L, 0]], dtype=torch.int32, device=dev)
ops.kv_tiles(qd, 0, D, 2 * D, kt, vt, table, desc, npages, heads, hd)
scale = 1.0 / math.sqrt(hd)
x = qd.double().cpu()
q, k, v = (x[:, i * D:(i + 1) * D].view(L, heads, hd) for i in range(3))
s = torch.einsum("qhd,khd->hqk", q, k) * scale
s = s.masked_fill((torch.arange(L)[None, :] > torch.arange(L)[:, None])[None], float("-inf"))
ref = torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), v).reshape(L, D).float()
for kernel in (1, 3, 2):
    ops.flash_attn_select(kernel)
    out = ops.flash_attn(qd, kt, vt, table, desc, L, heads, hd, True, scale).float().cpu()
    bad = (~torch.isfinite(out)).view(L, heads, hd).any(-1)
    err = ((out - ref).view(L, heads, hd).norm(dim=-1) / ref.view(L, heads, hd).norm(dim=-1))
    err = torch.where(torch.isfinite(err), err, torch.tensor(99.0))
    worst = torch.topk(err.flatten(), 12)
    print("kernel", kernel, "nonfinite (row,head):", bad.nonzero().tolist()[:20], "n", int(bad.sum()))
    print("   worst rows:", [(int(i) // heads, int(i) % heads, round(float(e), 4)) for e, i in zip(worst.values, worst.indices)])
    print("   rel_l2 finite rows:", float(((out - ref)[~bad.any(-1)]).norm() / ref[~bad.any(-1)].norm()))
