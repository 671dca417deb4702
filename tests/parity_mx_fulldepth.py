"""What precise level 3 (16-bit hi + MX-FP4 lo halves on every decoder GEMM's A operand) leaves of the full-depth distance from the
reference (CPU, oracle only; a report script behind DESIGN.md 4 / profiles/r6_parity_mx_fulldepth.txt, not a test).

    python tests/parity_mx_fulldepth.py c2_224 [fp16|bf16]       # ~6 min on 8 cores, ~35 GB of host memory

Runs the oracle's 32-layer decoder on the spliced embeddings of a tests/golden/make_golden_fulldepth.py case in fp32, in the emulation of
the standard mode and in the emulation of level 3, each on exact input embeddings (the towers in precise level 2 are 1e-5 from the
reference) and on the standard mode's embeddings."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vitron_oracle as O
from tests.golden import make_golden_fulldepth as FD
from vitron_amd import synth

torch.set_num_threads(8)
name = sys.argv[1]
emu = {"fp16": "fp16", "bf16": True}[sys.argv[2] if len(sys.argv) > 2 else "fp16"]
cfg = dict(synth.VICUNA_7B)
lsd = {k: v.float() for k, v in FD.llama_weights().items()}
vcfg, vsd, psd, rsd = FD.case_weights(name)
w = {"image_tower": {k: v.float() for k, v in vsd.items()}, "video_tower": {k: v.float() for k, v in vsd.items()},
     "projector": {k: v.float() for k, v in psd.items()}, "region": {k: v.float() for k, v in rsd.items()}, "llama": lsd}
cfgs = {"image": vcfg, "video": vcfg, "llama": cfg}
pix, ids = FD.case_inputs(name)


def rel(a, b):
    return float((a - b).norm() / b.norm())


with torch.no_grad():
    e32, _, _ = O.multimodal_prepare(w, cfgs, ids, None, [pix], None)
    e16, _, _ = O.multimodal_prepare(w, cfgs, ids, None, [pix], None, emulate_bf16=emu)
    l32, _ = O.llama_forward(lsd, cfg, e32, None, None, None, False)

    def rep(tag, lg):
        print("%-72s logits %.3e last %.3e top1 %.4f" % (tag, rel(lg, l32), rel(lg[0, -1], l32[0, -1]),
                                                        float((lg.argmax(-1) == l32.argmax(-1)).float().mean())), flush=True)
    rep("standard mode, standard embeddings", O.llama_forward(lsd, cfg, e16, None, None, None, emu)[0])
    rep("level 3 decoder, exact embeddings", O.llama_forward(lsd, cfg, e32, None, None, None, emu, precise_qk=3)[0])
    rep("level 3 decoder, standard embeddings", O.llama_forward(lsd, cfg, e16, None, None, None, emu, precise_qk=3)[0])
