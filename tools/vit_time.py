import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, synth
from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM
_lib.load(); dev = torch.device("cuda:0")
m = LlavaLlamaForCausalLM(LlavaConfig(**dict(synth.VICUNA_7B, num_hidden_layers=1), mm_hidden_size=1024))
m.init_synthetic(dev, seed=1, vit_image=dict(synth.VIT_L14, image_size=336), vit_video=dict(synth.VIT_L14, image_size=336, add_time_attn=True, num_frames=8))
clip = torch.randn((1, 3, 8, 336, 336), device=dev).bfloat16()
img = torch.randn((1, 3, 336, 336), device=dev).bfloat16()
for name, fn in (("video clip 8x336 (tower+projector)", lambda: m.encode_videos(clip)), ("image 336 (tower+projector)", lambda: m.encode_images(img))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t0) / 10 * 1e3, "ms")
