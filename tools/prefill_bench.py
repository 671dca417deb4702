"""One prefill configuration alone, N timed repetitions: tools/prefill_bench.py {image|clip} IMAGE_SIZE [reps] [dtype] [precise_qk|precise2|precise3]
  image 224 -> C2-224 (S = 768)    image 336 -> C2 (S = 1088)    clip 224 -> C3-224 (S = 2560)    clip 336 -> C3 (S = 5120)
Full-work step as in bench.py (tower, projector, splice, 32 layers, last-position logits, arg-max). Used under
rocprofv3 --kernel-trace --stats to see where each shape spends its time (profiles/r5_*_rocprofv3_kernel_stats.csv)."""
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
    kind, size = sys.argv[1], int(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    op = sys.argv[4] if len(sys.argv) > 4 else "bf16"
    precise = {"precise": 1, "precise_qk": 1, "precise2": 2, "precise3": 3}.get(sys.argv[5], 0) if len(sys.argv) > 5 else 0
    if os.environ.get("VT_PREFILL_BENCH_ABL") == "1":      # the -DVT_ABLATIONS test library (bf16 only): step-level A/B of its exploration switches
        _lib.load(ablations=True)
    else:
        _lib.load(operand=op)
    odt = _lib.torch_dtype(op)
    dev = torch.device("cuda:0")
    clip = kind == "clip"
    vcfg = dict(synth.VIT_L14, image_size=size, add_time_attn=clip, num_frames=8 if clip else 1)
    model = LlavaLlamaForCausalLM(LlavaConfig(**synth.VICUNA_7B, mm_hidden_size=1024, kv_prefix_reuse=False))
    model.init_synthetic(dev, seed=1234, vit_image=None if clip else vcfg, vit_video=vcfg if clip else None, dtype=odt)
    gen = synth.make_generator(4321, dev)
    n_img = 8 if clip else 1
    pix = torch.randn((3, 8, size, size) if clip else (3, size, size), generator=gen, device=dev).to(odt)
    ids = torch.cat([torch.tensor([1], device=dev), torch.full((n_img,), -200, device=dev),
                     torch.randint(3, 32000, (511,), generator=gen, device=dev)]).unsqueeze(0)
    ids_host = ids.cpu()
    S = n_img * (size // 14) ** 2 + 512
    llama = model.get_model().llama
    model.set_precise(precise)
    model._ensure_kv((S + 63) // 64 + 4)

    def step():
        (_, _, _, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [pix], None, input_ids_host=ids_host)
        seq = SequenceState()
        ops.argmax(llama_forward(llama, model.kv, [seq], embeds[0], [embeds.shape[1]]))
        model.kv.release(seq.pages)
        return embeds.shape[1]

    for _ in range(3):
        assert step() == S
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"config": f"{kind} {size}px + 512 tokens", "operand": op, "precise_qk": precise, "S": S, "ms": dt * 1e3, "tokens_per_s": S / dt}))


if __name__ == "__main__":
    main()
