"""Continuous batching (vitron_amd.serving.ServingEngine) vs one-request-at-a-time `generate` on the 7B-shaped model:
R requests of (336 px image + box + short prompt), N new tokens each, greedy."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, synth  # noqa: E402
from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM  # noqa: E402
from vitron_amd.serving import ServingEngine  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    _lib.load()
    dev = torch.device("cuda:0")
    model = LlavaLlamaForCausalLM(LlavaConfig(**synth.VICUNA_7B, mm_hidden_size=1024, mm_region_image_size=336, kv_prefix_reuse=False))
    model.init_synthetic(dev, seed=1234, vit_image=dict(synth.VIT_L14, image_size=336), vit_video=None)
    gen = synth.make_generator(4321, dev)
    reqs = []
    for i in range(R):
        ids = torch.tensor([[1, -200] + torch.randint(3, 32000, (6,), generator=gen, device=dev).tolist() + [-300, 1] +
                            torch.randint(3, 32000, (24,), generator=gen, device=dev).tolist()], device=dev)
        reqs.append((ids, [torch.randn((3, 336, 336), generator=gen, device=dev).bfloat16()], [[10.0 + i, 20.0, 200.0, 300.0]]))
    model.generate(reqs[0][0], images=reqs[0][1], regions=reqs[0][2], do_sample=False, max_new_tokens=4, eos_token_id=-1)   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for ids, im, rg in reqs:
        model.generate(ids, images=im, regions=rg, do_sample=False, max_new_tokens=N, eos_token_id=-1)
    torch.cuda.synchronize()
    t_seq = time.perf_counter() - t0
    for mb, bp in ((4, False), (16, False), (16, True), (32, False), (32, True)):
        if mb > R and mb != 4:
            continue
        eng = ServingEngine(model, max_batch=mb, kv_pages=R * 14 + 8, batch_prefill=bp)
        for ids, im, rg in reqs:
            eng.submit(ids, im, rg, N, eos_token_id=-1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.run()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        assert all(len(v) == N for v in out.values())
        print(json.dumps({"requests": R, "new_tokens_each": N, "max_batch": mb, "batch_prefill": bp, "seconds": round(t, 3),
                          "generated_tokens_per_s": round(R * N / t, 1), "one_at_a_time_seconds": round(t_seq, 3),
                          "one_at_a_time_tokens_per_s": round(R * N / t_seq, 1), "speedup": round(t_seq / t, 2)}), flush=True)


if __name__ == "__main__":
    main()
