"""Which storage point of the fp16 build carries how much of the FULL-DEPTH distance from the reference (CPU, oracle only; a report
script behind DESIGN.md 4 / profiles/r5_parity_round_points_fulldepth.txt, not a test).

    python tests/parity_round_points_fulldepth.py c2_224        # ~10 min on 8 cores, ~35 GB of host memory

Runs the oracle's 32-layer decoder on the REAL spliced embeddings of a tests/golden/make_golden_fulldepth.py case (hash-stream weights,
23-layer tower + projector through the oracle) in fp32 and with ONE class of fp16 storage point switched on in all 32 layers at a time
(every "ONLY" line also carries the V^T page's fp16 image of v: 2.6e-4). Round 5 found with it that precise_qk -- which removes the five
q / k-path points that dominate a 1-2 layer chain on token-embedding-sized rows -- buys little at full depth behind real visual rows
(row norm 65 vs 1.15 for text rows): there the A operands of the big GEMMs (post-attention norm -> gate/up 9.2e-4, attention output ->
o_proj 6.4e-4, SwiGLU output -> down_proj 6.4e-4) and the tower's own error in the input embeddings (9.5e-4 at the logits) carry the
distance; the five q / k-path points together 6.4e-4."""
import math
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import vitron_oracle as O
from tests.golden import make_golden_fulldepth as FD
from vitron_amd import synth
torch.set_num_threads(8)
name = sys.argv[1]
cfg = dict(synth.VICUNA_7B); L=32
lsd = {k: v.float() for k,v in FD.llama_weights().items()}
vcfg, vsd, psd, rsd = FD.case_weights(name)
w = {"image_tower": {k:v.float() for k,v in vsd.items()}, "video_tower": {k:v.float() for k,v in vsd.items()}, "projector": {k:v.float() for k,v in psd.items()}, "region": {k:v.float() for k,v in rsd.items()}, "llama": lsd}
cfgs = {"image": vcfg, "video": vcfg, "llama": cfg}
pix, ids = FD.case_inputs(name)
def rel(a,b): return float((a-b).norm()/b.norm())
orig=O._r
NAMES=["norm1","q","k","v","q_rope","k_rope","attn_out","norm2","swiglu"]
def run(x, on_classes, pv=True, vt=True, final=True):
    cnt=[0]
    def r(t, emulate):
        i=cnt[0]; cnt[0]+=1
        if i >= 9*L: return orig(t,emulate) if final else t
        return orig(t,emulate) if NAMES[i%9] in on_classes else t
    O._r=r
    fr, fs = O.fp16_round, O.fp16_store
    if not pv: O.fp16_round = lambda t: t
    if not vt: O.fp16_store = lambda t: t
    try:
        lg,_,h = O.llama_forward(lsd,cfg,x,None,None,None,"fp16",return_hidden=True)
    finally:
        O._r=orig; O.fp16_round=fr; O.fp16_store=fs
    return lg,h
with torch.no_grad():
    e32, _, _ = O.multimodal_prepare(w, cfgs, ids, None, [pix], None)
    e16, _, _ = O.multimodal_prepare(w, cfgs, ids, None, [pix], None, emulate_bf16="fp16")
    l32,_,h32 = O.llama_forward(lsd,cfg,e32,None,None,None,False,return_hidden=True)
    def rep(tag, lg, h): print(tag, "hidden %.3e logits %.3e last %.3e" % (rel(h,h32), rel(lg,l32), rel(lg[0,-1], l32[0,-1])), flush=True)
    for c in NAMES:
        lg,h = run(e32, {c}, pv=False, vt=True, final=False); rep("ONLY "+c, lg,h)
    lg,h = run(e32, set(), pv=False, vt=True, final=True); rep("ONLY final norm", lg,h)
    lg,h = run(e32, {"norm1","q","k","q_rope","k_rope"}, pv=False, vt=True, final=False); rep("ONLY the five q/k-path points", lg,h)
    lg,h = run(e32, {"v","attn_out","norm2","swiglu"}, pv=True, vt=True, final=True); rep("everything BUT the five q/k-path points", lg,h)
