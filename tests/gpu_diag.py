"""Error-localisation report for the end-to-end path on the GPU box (not a test): per-stage / per-row rel-L2 of the
HIP path vs the emulating oracle and vs the reference-generated goldens. python tests/gpu_diag.py > gpurun_out/diag.txt; under tests/ because it calls the oracle"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vitron_oracle as O  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.test_gpu_model import CFGS, _states  # noqa: E402
from tests.util import f32, rel_l2  # noqa: E402
from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM  # noqa: E402

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
dev = torch.device("cuda:0")
st = _states()
cfg = LlavaConfig(**cases.LLM, mm_hidden_size=cases.MM_HIDDEN, mm_image_tower="g/LanguageBind_Image", mm_video_tower="g/LanguageBind_Video_merge")
m = LlavaLlamaForCausalLM(cfg)
m.get_image_tower().load_state(cases.VIT_IMAGE, st["image_tower"])
m.get_video_tower().load_state(cases.VIT_VIDEO, st["video_tower"])
sd = dict(st["llama"])
sd.update({"model.mm_projector." + k: v for k, v in st["projector"].items()})
sd.update({"model.region_extractor." + k: v for k, v in st["region"].items()})
m.load_state_dict(sd)
m.to(dev)
w = {k: f32(v) for k, v in st.items()}
g = np.load(os.path.join(G, "glue_llm.npz"))
gv = np.load(os.path.join(G, "vit.npz"))

from vitron_amd.engine import PackedVit  # noqa: E402
for name, c, shape in (("video", cases.VIT_VIDEO, (2, 3, 4, 56, 56)), ("image", cases.VIT_IMAGE, (3, 3, 56, 56))):
    vsd = st[name + "_tower"]
    x = cases.pixels(shape, cases.SEED_PIX)
    for nl in range(0, 4):
        sel = nl
        vit = PackedVit(vsd, c, dev, select_layer=sel)
        _, hid = vit.forward(x.to(dev).bfloat16(), return_hidden=True)
        emu = O.vit_forward(f32(vsd), c, x, nl, emulate_bf16=True)
        ref = torch.as_tensor(gv[f"{name}_hidden_{nl}"])
        print(f"vit {name} layers={nl}: vs emu {rel_l2(hid, emu):.2e}  vs ref-fp32 {rel_l2(hid, ref):.2e}  emu-vs-ref {rel_l2(emu, ref):.2e}")

for name, case in cases.glue_cases().items():
    m.config.tokenizer_model_max_length = case.get("max_length")
    m.config.tokenizer_padding_side = case.get("padding_side", "right")
    ids = case["input_ids"].to(dev)
    am = None if case["attention_mask"] is None else case["attention_mask"].to(dev)
    images = [im.to(dev).bfloat16() for im in case["images"]]
    (_, _, _, _, embeds, _) = m.prepare_inputs_labels_for_multimodal(ids, None, am, None, None, images, case["regions"])
    out = m(input_ids=ids, attention_mask=am, images=images, regions=case["regions"], use_cache=False)
    e_logits, e_embeds, e_mask, _ = O.multimodal_forward(w, CFGS, case["input_ids"], case["attention_mask"], case["images"], case["regions"],
                                                         case.get("max_length"), case.get("padding_side", "right"), True)
    ref_e, ref_l = torch.as_tensor(g[f"{name}_embeds"]), torch.as_tensor(g[f"{name}_logits"])
    valid = torch.as_tensor(g[f"{name}_mask"]).bool()
    lg = out.logits.cpu()
    print(f"{name}: embeds vs emu {rel_l2(embeds.float().cpu()[valid], e_embeds[valid]):.2e} vs ref {rel_l2(embeds.float().cpu()[valid], ref_e[valid]):.2e} | "
          f"logits vs emu {rel_l2(lg[valid], e_logits[valid]):.2e} vs ref {rel_l2(lg[valid], ref_l[valid]):.2e} emu-vs-ref {rel_l2(e_logits[valid], ref_l[valid]):.2e} | "
          f"argmax agree vs ref {(lg[valid].argmax(-1) == ref_l[valid].argmax(-1)).float().mean():.3f}")
    for b in range(lg.shape[0]):
        rows = [f"{rel_l2(lg[b, j], ref_l[b, j]):.1e}" for j in range(lg.shape[1]) if valid[b, j]]
        erow = [f"{rel_l2(embeds[b, j].float().cpu(), ref_e[b, j]):.1e}" for j in range(lg.shape[1]) if valid[b, j]]
        print("   logits/row:", " ".join(rows[:40]))
        print("   embeds/row:", " ".join(erow[:40]))
