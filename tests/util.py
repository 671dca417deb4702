import torch


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def randn(shape, seed, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf16r(torch.randn(shape, generator=g) * std)


def f32(sd):
    return {k: v.float().cpu() for k, v in sd.items()}
