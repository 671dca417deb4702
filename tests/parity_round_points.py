"""Which storage point of a 7B-width decoder layer contributes how much of the fp16 build's distance from fp32 (CPU, oracle only).

    python tests/parity_round_points.py        # ~2 min on 8 cores

Runs the oracle's decoder layer (H = 4096 / I = 11008 / 32 heads, one layer, 256 rows, the benchmark's init) in fp32 and with the fp16
storage emulation, then with ONE storage point switched off / on at a time. Report script behind DESIGN.md 4 (not a test)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vitron_oracle as O
from tests.golden import cases
from vitron_amd import synth
torch.set_num_threads(8)
L = 1
cfg = dict(synth.VICUNA_7B, num_hidden_layers=L)
sd = synth.llama_state(cfg, synth.make_generator(cases.FW_SEED + L), **cases.FW_INIT)
sd = {k: v.float() for k, v in sd.items()}
S = 256
x = cases.fw_llama_embeds(S, cases.FW_SEED + S).unsqueeze(0).float()
def rel(a, b): return float((a - b).norm() / b.norm())
lg32, _, h32 = O.llama_forward(sd, cfg, x, None, None, None, False, return_hidden=True)
lg16, _, h16 = O.llama_forward(sd, cfg, x, None, None, None, "fp16", return_hidden=True)
print("all points fp16: hidden", rel(h16, h32), "logits", rel(lg16, lg32))
# disable one rounding call at a time: wrap _r with a counter
orig = O._r
names = {}
def run(skip):
    cnt = [0]
    def r(t, emulate):
        i = cnt[0]; cnt[0] += 1
        if i in skip: return t
        return orig(t, emulate)
    O._r = r
    try:
        lg, _, h = O.llama_forward(sd, cfg, x, None, None, None, "fp16", return_hidden=True)
    finally:
        O._r = orig
    return rel(h, h32), rel(lg, lg32), cnt[0]
base = run(set())
print("points per pass:", base[2])
for i in range(base[2]):
    h, l, _ = run({i})
    print(f"without rounding point {i}: hidden {h:.3e} logits {l:.3e}  (delta var hidden {base[0]**2 - h**2:+.2e})")
# only one point on
for i in range(base[2]):
    h, l, _ = run(set(range(base[2])) - {i})
    print(f"ONLY rounding point {i}: hidden {h:.3e} logits {l:.3e}")
