"""Step-shaped A/B of GEMM tile configurations: the decoder's per-layer launch sequence (RMSNorm, QKV GEMM, RoPE / page write,
flash attention, o_proj, RMSNorm, gate/up SwiGLU GEMM, down_proj) over `layers` layers with DISTINCT weights (13 GB at 7B: every
layer's 400 MB arrive cold from HBM, as inside the prefill step), each arm selecting its own tile configuration for the four big
GEMMs, arms alternated pass by pass on one box. tools/gemm_ab.cpp re-runs ONE shape from the Infinity Cache and overstates what a
kernel change is worth inside the step (+5 % there, +1 % in the step); this harness measures the step.

    python tools/step_ab.py [--rows 5120] [--layers 32] [--passes 6] --arm auto=0,0,0,0 --arm p4=10,10,10,10
    (an arm = name=cfg_qkv,cfg_o,cfg_gateup,cfg_down; cfg numbers as in include/vitron_hip.h, 0 = the dispatcher's choice)

Synthetic weights N(0, 0.02^2) and inputs; the numbers are times, not results (the residual stream is re-seeded every pass)."""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402
from vitron_amd.engine import rope_tables  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=5120)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--passes", type=int, default=6)
    ap.add_argument("--arm", action="append", default=[])
    args = ap.parse_args()
    arms = [(a.split("=")[0], [int(c) for c in a.split("=")[1].split(",")]) for a in (args.arm or ["auto=0,0,0,0", "p4=10,10,10,10"])]
    _lib.load()
    dev = torch.device("cuda:0")
    S, H, I, heads, hd, L = args.rows, 4096, 11008, 32, 128, args.layers
    g = torch.Generator(device=dev).manual_seed(1)

    def w(n, k):
        return (torch.randn((n, k), generator=g, device=dev) * 0.02).to(torch.bfloat16)

    layers = [dict(qkv=w(3 * H, H), o=w(H, H), gu=w(2 * I, H), down=w(H, I), n1=torch.ones(H, device=dev), n2=torch.ones(H, device=dev))
              for _ in range(L)]
    x0 = torch.randn((S, H), generator=g, device=dev) * 0.02
    x = x0.clone()
    nt = (S + 63) // 64
    desc = torch.tensor([[0, S, S, 0]], dtype=torch.int32, device=dev)
    table = torch.arange(nt, dtype=torch.int32, device=dev)
    kt = torch.zeros(nt * heads * 64 * hd, dtype=torch.bfloat16, device=dev)
    vt = torch.zeros_like(kt)
    cos, sin = rope_tables(hd, S, 10000.0, dev)
    pos = torch.arange(S, dtype=torch.int32, device=dev)
    qkv = torch.empty((S, 3 * H), dtype=torch.bfloat16, device=dev)
    att = torch.empty((S, H), dtype=torch.bfloat16, device=dev)
    act = torch.empty((S, I), dtype=torch.bfloat16, device=dev)
    scale = 1.0 / math.sqrt(hd)

    def one_pass(cfg):
        x.copy_(x0)
        for lw in layers:
            y = ops.rmsnorm(x, lw["n1"], 1e-5)
            ops.gemm(y, lw["qkv"], None, ops.EPI_BF16, out=qkv, cfg=cfg[0])
            ops.kv_tiles(qkv, 0, H, 2 * H, kt, vt, table, desc, nt, heads, hd, cos, sin, pos)
            ops.flash_attn(qkv, kt, vt, table, desc, S, heads, hd, True, scale, out=att)
            ops.gemm(att, lw["o"], None, ops.EPI_F32_RESID, out=x, cfg=cfg[1])
            y = ops.rmsnorm(x, lw["n2"], 1e-5)
            ops.gemm(y, lw["gu"], None, ops.EPI_SWIGLU_BF16, out=act, cfg=cfg[2])
            ops.gemm(act, lw["down"], None, ops.EPI_F32_RESID, out=x, cfg=cfg[3])

    times = {name: [] for name, _ in arms}
    for name, cfg in arms:   # warm-up (and the plan each arm's AUTO entries resolve to)
        one_pass(cfg)
    torch.cuda.synchronize()
    for _ in range(args.passes):
        for name, cfg in arms:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            one_pass(cfg)
            b.record()
            torch.cuda.synchronize()
            times[name].append(a.elapsed_time(b))
    plans = {"qkv": ops.gemm_plan(S, 3 * H, H, ops.EPI_BF16), "o_proj": ops.gemm_plan(S, H, H, ops.EPI_F32_RESID),
             "gate_up": ops.gemm_plan(S, 2 * I, H, ops.EPI_SWIGLU_BF16), "down_proj": ops.gemm_plan(S, H, I, ops.EPI_F32_RESID)}
    out = {"rows": S, "layers": L, "passes": args.passes, "auto_plans": {k: list(v) for k, v in plans.items()}, "arms": {}}
    for name, cfg in arms:
        t = sorted(times[name])
        out["arms"][name] = {"cfg_qkv_o_gateup_down": cfg, "ms_median": round(t[len(t) // 2], 3), "ms_min": round(t[0], 3), "ms_max": round(t[-1], 3)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
