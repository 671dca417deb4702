import os
import sys

import torch


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    v = float((a - b).norm() / b.norm().clamp_min(1e-30))
    out = os.environ.get("VT_TOL_REPORT")      # measurement aid: every distance with its call site, to set bounds from measurements
    if out:
        f = sys._getframe(1)
        with open(out, "a") as fh:
            fh.write(f"{os.path.basename(f.f_code.co_filename)}:{f.f_lineno}\t{v:.6e}\n")
    return v


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def randn(shape, seed, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf16r(torch.randn(shape, generator=g) * std)


def f32(sd):
    return {k: v.float().cpu() for k, v in sd.items()}
