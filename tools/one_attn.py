import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops
S, heads, hd = 5120, 32, 128
_lib.load(); dev = torch.device("cuda:0"); D = heads * hd
qkv = torch.randn((S, 3 * D), device=dev).bfloat16(); nt = (S + 63) // 64
desc = torch.tensor([[0, S, S, 0]], dtype=torch.int32, device=dev); table = torch.arange(nt, dtype=torch.int32, device=dev)
kt = torch.zeros(nt * heads * 64 * hd, dtype=torch.bfloat16, device=dev); vt = torch.zeros_like(kt)
ops.kv_tiles(qkv, 0, D, 2 * D, kt, vt, table, desc, nt, heads, hd)
for _ in range(3):
    ops.flash_attn(qkv, kt, vt, table, desc, S, heads, hd, True, 1 / math.sqrt(hd))
torch.cuda.synchronize()
