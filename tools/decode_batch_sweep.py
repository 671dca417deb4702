import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vitron_amd import _lib, synth
from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM
_lib.load(); dev = torch.device("cuda:0")
model = LlavaLlamaForCausalLM(LlavaConfig(**synth.VICUNA_7B, mm_hidden_size=1024, kv_prefix_reuse=False))
model.init_synthetic(dev, seed=1234, vit_image=None, vit_video=None)
llama = model.get_model().llama
for b in (4, 16, 32, 64):
    r = bench.decode_report(model, llama, dev, 32, batch=b, ctx=609)
    print(b, round(r["ms_per_step"], 3), "ms/step", round(r["tokens_per_s"]), "tok/s", {k: round(v, 3) for k, v in r["kernel_ms_per_step"].items()})
