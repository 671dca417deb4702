"""BASELINE configs[1] alone: one 336x336 image + 512-token prompt (S = 1088), prefill + first token, N timed repetitions.
Used under rocprofv3 --kernel-trace --stats to see where a single-image turn spends its time."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, synth  # noqa: E402
from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    _lib.load()
    dev = torch.device("cuda:0")
    model = LlavaLlamaForCausalLM(LlavaConfig(**synth.VICUNA_7B, mm_hidden_size=1024, kv_prefix_reuse=False))
    model.init_synthetic(dev, seed=1234, vit_image=dict(synth.VIT_L14, image_size=336), vit_video=None)
    gen = synth.make_generator(4321, dev)
    image = torch.randn((3, 336, 336), generator=gen, device=dev).bfloat16()
    ids = torch.cat([torch.tensor([1, -200], device=dev), torch.randint(3, 32000, (511,), generator=gen, device=dev)]).unsqueeze(0)
    run = lambda: model.generate(ids, images=[image], do_sample=False, max_new_tokens=1, eos_token_id=-1)  # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"config": "C2 336px image + 512 tokens", "S": 1088, "ms": dt * 1e3, "tokens_per_s": 1088 / dt}))


if __name__ == "__main__":
    main()
