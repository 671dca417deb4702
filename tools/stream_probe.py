import torch, json
dev = torch.device("cuda:0")
S, D = 5120, 4096
def timed(fn, n=20, windows=7):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    out = []
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(out)[len(out)//2]
xs = [torch.randn((S, 3*D), device=dev).bfloat16() for _ in range(3)]
ys = [torch.empty((S, 3*D), device=dev, dtype=torch.bfloat16) for _ in range(3)]
ks = [torch.empty((S, D), device=dev, dtype=torch.bfloat16) for _ in range(3)]
r = {}
r["full_copy_126r_126w"] = timed(lambda i: ys[i%3].copy_(xs[i%3]))
r["inplace_q_42r_42w"] = timed(lambda i: xs[i%3][:, :D].mul_(1.5))
r["strided_to_contig_42r_42w"] = timed(lambda i: ks[i%3].copy_(xs[i%3][:, D:2*D]))
r["read_only_sum_126r"] = timed(lambda i: xs[i%3].sum())
r["fill_42w"] = timed(lambda i: ks[i%3].fill_(1.0))
r["fill_126w"] = timed(lambda i: ys[i%3].fill_(1.0))
print(json.dumps({k: round(v, 2) for k, v in r.items()}))
