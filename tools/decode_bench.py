"""BASELINE configs[1] and configs[4] on one MI355X (parity-test cases, reported in DESIGN.md -- not the bench line):
  C2: single 336x336 image + 512-token prompt prefill (image tower + projector + decoder, S = 1088)
  C5: 4 x (336 image + box) through region_extractor + projector, prefill, then greedy decode with the paged-KV kernels.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops, synth  # noqa: E402
from vitron_amd.engine import SequenceState, llama_forward  # noqa: E402
from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    _lib.load()
    dev = torch.device("cuda:0")
    vit_image = dict(synth.VIT_L14, image_size=336)
    model = LlavaLlamaForCausalLM(LlavaConfig(**synth.VICUNA_7B, mm_hidden_size=1024, mm_region_image_size=336))
    model.init_synthetic(dev, seed=1234, vit_image=vit_image, vit_video=None)
    model.config.kv_prefix_reuse = False     # every timed call must do the whole job: no multi-turn reuse of towers / KV pages
    gen = synth.make_generator(4321, dev)
    img = lambda: torch.randn((3, 336, 336), generator=gen, device=dev).bfloat16()  # noqa: E731

    # ---- C2 ----
    ids = torch.cat([torch.tensor([1, -200], device=dev), torch.randint(3, 32000, (511,), generator=gen, device=dev)]).unsqueeze(0)
    image = img()
    def c2():
        return model.generate(ids, images=[image], do_sample=False, max_new_tokens=1, eos_token_id=-1)
    for _ in range(2):
        c2()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20 if steps <= 0 else 5
    for _ in range(reps):
        c2()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    S = 576 + 512
    print(json.dumps({"config": "C2 336px image + 512 tokens prefill", "S": S, "ms": dt * 1e3, "tokens_per_s": S / dt}), flush=True)
    if steps <= 0:      # C2 only (python tools/decode_bench.py 0): the per-kernel profile of the single-image prefill
        return

    # ---- C5 ----
    B = 4
    boxes = [[0, 0, 336, 336], [0, 88.4, 176.8, 176.8], [150, 30, 270, 300], [10.5, 10.5, 12, 12]]
    prompt = [1, -200] + torch.randint(3, 32000, (6,), generator=gen, device=dev).tolist() + [-300, 1] + torch.randint(3, 32000, (24,), generator=gen, device=dev).tolist()
    ids5 = torch.tensor([prompt] * B, device=dev)
    images = [img() for _ in range(B)]
    model.generate(ids5, images=images, regions=boxes, do_sample=False, max_new_tokens=4, eos_token_id=-1)   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate(ids5, images=images, regions=boxes, do_sample=False, max_new_tokens=steps, eos_token_id=-1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.profile_begin()   # separate pass: the per-launch events perturb the wall clock
    model.generate(ids5, images=images, regions=boxes, do_sample=False, max_new_tokens=steps, eos_token_id=-1)
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    # decode-only timing: second call with 1 token gives the prefill cost
    t1 = time.perf_counter()
    model.generate(ids5, images=images, regions=boxes, do_sample=False, max_new_tokens=1, eos_token_id=-1)
    torch.cuda.synchronize()
    tp = time.perf_counter() - t1
    dec = (dt - tp) / (steps - 1)
    ctx = 576 + len(prompt) - 1
    wbytes = 6_738_415_616 * 2
    print(json.dumps({"config": f"C5 4x(336 image + box), context {ctx}, {steps} greedy steps, batch {B}", "prefill_ms": tp * 1e3,
                      "ms_per_decode_step": dec * 1e3, "decode_tokens_per_s": B / dec,
                      "weight_stream_GBps": wbytes / dec / 1e9, "hbm_frac_of_8TBps": wbytes / dec / 8e12,
                      "skinny_gemm_ms_per_step": prof["gemm_skinny"]["ms"] / steps, "attn_decode_ms_per_step": prof["attn_decode"]["ms"] / steps,
                      "new_tokens_shape": list(out.shape)}), flush=True)


if __name__ == "__main__":
    main()
