"""Shared helpers of the BASELINE-width parity tests (tests/test_oracle_fullwidth.py on the CPU, tests/test_gpu_parity_fullwidth.py
on the GPU): the weights / inputs of tests/golden/cases.py's FW_* cases regenerated from their seeds, and the comparison of a big
[rows, dim] tensor against its compact pin in tests/golden/fullwidth.npz (projections of every row on fixed directions + a few
whole rows + top-5 ids, written by tests/golden/make_golden.gen_fullwidth from the REFERENCE's own modules)."""
import os

import numpy as np
import torch

from tests.golden import cases
from vitron_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "fullwidth.npz")
GOLDEN_224 = os.path.join(os.path.dirname(__file__), "golden", "fullwidth_224.npz")     # the reference-native 224 px shapes (round 5)
ALL_LLAMA = dict(cases.FW_LLAMA, **cases.FW224_LLAMA)


def golden(which="336"):
    return np.load(GOLDEN_224 if str(which) == "224" else GOLDEN)


def golden_of(name):
    """The golden file that holds a decoder / tower case name."""
    return golden("224" if (name in cases.FW224_LLAMA or name.endswith("224")) else "336")


def llama_case(name):
    S, L = ALL_LLAMA[name]
    cfg = dict(synth.VICUNA_7B, num_hidden_layers=L)
    sd = synth.llama_state(cfg, synth.make_generator(cases.FW_SEED + L), **cases.FW_INIT)
    return cfg, sd, cases.fw_llama_embeds(S, cases.FW_SEED + S)


def vit_case(name):
    """name = 'video336' | 'image336' | 'video224' | 'image224'."""
    video, size = name.startswith("video"), int(name[-3:])
    cfg = dict(synth.VIT_L14, image_size=size, add_time_attn=video, num_frames=8 if video else 1, num_hidden_layers=cases.FW_VIT_LAYERS)
    sd = synth.vit_state(cfg, synth.make_generator(cases.FW_SEED + 7), **cases.FW_INIT)
    shape = {("video", 336): cases.FW_VIDEO_SHAPE, ("image", 336): cases.FW_IMAGE_SHAPE,
             ("video", 224): cases.FW224_VIDEO_SHAPE, ("image", 224): cases.FW224_IMAGE_SHAPE}[(name[:5], size)]
    return cfg, sd, cases.pixels(shape, cases.FW_SEED + 8)


def projector_case(which="336"):
    sd = synth.projector_state(1024, 4096, synth.make_generator(cases.FW_SEED + 9), **cases.FW_INIT)
    return sd, cases.features((cases.FW224_PROJ_ROWS if str(which) == "224" else cases.FW_PROJ_ROWS, 1024), cases.FW_SEED + 10)


def region_case(canvas, grid=24):
    """grid = 24: the 336 px tower's patch grid on a 224 / 336 canvas; grid = 16: the reference-native case (224 px tower, 224 canvas)."""
    sd = synth.region_state(1024, 4096, synth.make_generator(cases.FW_SEED + 11), **cases.FW_INIT)
    boxes = cases.FW224_BOXES if grid == 16 else (cases.FW_BOXES_224 if canvas == 224 else cases.FW_BOXES_336)
    return sd, cases.features((len(boxes), grid * grid, 1024), cases.FW_SEED + 12), boxes


def operand(op):
    """(torch dtype, oracle emulation flag, oracle rounding function) of an operand build name ('bf16' | 'fp16')."""
    from oracle import vitron_oracle as O
    return (torch.bfloat16, True, O.bf16_round) if op == "bf16" else (torch.float16, "fp16", O.fp16_store)


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def vs_pin(t, g, tag):
    """Distances of tensor t [rows, dim] from its pin: (rel-L2 over the projections of ALL rows, rel-L2 over the stored whole rows)."""
    t = torch.as_tensor(t).reshape(-1, t.shape[-1]).cpu()
    proj = t.double() @ cases.fw_directions(t.shape[-1])
    rows = g[f"{tag}_rowidx"].tolist()
    return rel(proj, g[f"{tag}_proj"]), rel(t[rows].float(), g[f"{tag}_rows"])


def topk_agreement(logits, g, tag):
    """(top-1 agreement, mean top-5 overlap) of `logits` [rows, V] with the reference's stored top-5 ids."""
    ref = torch.as_tensor(g[f"{tag}_top5"]).long()
    got = torch.as_tensor(logits).float().cpu().topk(5, dim=-1).indices
    top1 = float((got[:, 0] == ref[:, 0]).double().mean())
    overlap = float(torch.tensor([len(set(a.tolist()) & set(b.tolist())) / 5.0 for a, b in zip(got, ref)]).mean())
    return top1, overlap


_ORACLE_CACHE = {}


def oracle_llama(name, emulate, precise_qk=False):
    """(logits [S, V], final hidden [S, H]) of the oracle on llama_case(name): fp32 (emulate False), bf16-storage emulation (True) or
    fp16-storage emulation ("fp16"; precise_qk=True: with the storage points of the precise_qk prefill) -- computed once per test
    session: the prefill and the decode parity tests compare against the same passes (each ~1 min of host time at the 7B width)."""
    from oracle import vitron_oracle as O
    emulate = emulate if isinstance(emulate, str) else bool(emulate)
    key = (name, emulate, int(precise_qk))
    if key not in _ORACLE_CACHE:
        cfg, sd, x = llama_case(name)
        with torch.no_grad():
            lg, _, h = O.llama_forward({k: v.float() for k, v in sd.items()}, cfg, x.unsqueeze(0), emulate_bf16=emulate, return_hidden=True,
                                       precise_qk=precise_qk)
        _ORACLE_CACHE[key] = (lg[0], h[0])
    return _ORACLE_CACHE[key]


def vs_wide_pin(logits, g):
    """The wider pin of the full-depth logits (round 6; tests/golden/make_golden_fulldepth.py): rel-L2 over the projections of ALL rows on 32
    directions, and -- for the cases that hold them -- over the last 64 rows whole. (None, None) parts when the golden predates them."""
    t = torch.as_tensor(logits).reshape(-1, logits.shape[-1]).cpu()
    p32 = rel(t.double() @ cases.fw_directions(t.shape[-1], n=32, seed=cases.FW_SEED + 1), g["logits_proj32"]) if "logits_proj32" in g.files else None
    tail = rel(t[-64:].float(), g["logits_tail"]) if "logits_tail" in g.files else None
    return p32, tail
