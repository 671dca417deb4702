"""north_star's per-operator bar at the width the benchmark runs: every operator of a Vicuna-7B-shaped decoder layer (H = 4096, I = 11008,
32 heads, bench init, 1088 rows), run on the emulating oracle's OWN intermediates (teacher forcing, so each number is what ONE operator
adds on identical inputs), is within 1e-3 rel-L2 of the oracle's emulation of its bf16 / fp16 storage points -- flash attention included
since round 3 (softmax weights in fp16: 6.7e-4; round 2, bf16: 1.97e-3). Numbers: tests/parity_ops_fullwidth.py, profiles/r3_parity_ops_fullwidth.txt.
The whole layer, free running, is allowed 1.5 x its measured 3.3e-3 (rounding flips between operators)."""
import json

import pytest

pytestmark = pytest.mark.gpu

TOL_OP = 1e-3          # north_star
MEASURED_X_1_5 = {"rmsnorm": 1e-6, "qkv_gemm": 6.3e-5, "rope_q": 1e-5, "rope_k_pages": 1e-5, "flash_attn": 1.0e-3, "o_proj_resid": 1e-6,
                  "rmsnorm2": 3e-5, "swiglu_gemm": 7.8e-5, "down_proj_resid": 1.1e-6, "whole_layer_hidden": 5.0e-3}


# fp16-operand build (libvitron_hip_f16.so): one store leaves 2^-12 per element (2.1e-4 rel-L2) instead of bf16's 2^-9 (1.66e-3),
# so the operators are also compared with PLAIN fp32 here -- north_star's 1e-3 against an fp32 evaluation, not only against the
# emulation of the kernel's own storage points. Bounds = first measurement x 1.5 (profiles/r4_parity_ops_fullwidth_fp16.txt).
# measured: rmsnorm 2.7e-6, qkv 1.5e-5, rope 3.6e-6, flash attention 2.5e-4 (2.6e-4 from plain fp32), swiglu 1.9e-5, whole layer 6.7e-4
MEASURED_X_1_5_FP16 = {"rmsnorm": 4.1e-6, "qkv_gemm": 2.3e-5, "rope_q": 5.4e-6, "rope_k_pages": 5.3e-6, "flash_attn": 3.8e-4, "o_proj_resid": 1e-6,
                       "rmsnorm2": 7.7e-6, "swiglu_gemm": 2.9e-5, "down_proj_resid": 1.1e-6, "whole_layer_hidden": 1.0e-3}


@pytest.mark.parametrize("operand", ["bf16", "fp16"])
def test_every_decoder_operator_within_1e3_of_the_emulation_at_7b_width(operand):
    from tests import parity_ops_fullwidth as P
    rep = P.measure(1088, operand)
    print(f"[parity-ops-{operand}] " + json.dumps({k: {a: round(b, 7) for a, b in v.items()} for k, v in rep.items()}), flush=True)
    for name, r in rep.items():
        bound = (MEASURED_X_1_5 if operand == "bf16" else MEASURED_X_1_5_FP16)[name]
        if operand == "fp16" and "emu_vs_fp32" in r:
            assert r["emu_vs_fp32"] <= 1e-3, (name, r)        # ONE fp16 store is inside north_star's tolerance against plain fp32
        assert r["vs_emu"] <= bound, (name, r)
        if name != "whole_layer_hidden":
            assert r["vs_emu"] <= TOL_OP, (name, r)
    fa = rep["flash_attn"]
    assert fa["vs_fp32"] <= 1.25 * fa["emu_vs_fp32"] + 2e-4, fa          # no farther from fp32 than the emulation of its storage points
    if operand == "fp16":       # the whole LAYER, free running, is inside north_star's 1e-3 of the emulation, and attention of plain fp32
        assert rep["whole_layer_hidden"]["vs_emu"] <= TOL_OP and fa["vs_fp32"] <= 4e-4, (rep["whole_layer_hidden"], fa)
