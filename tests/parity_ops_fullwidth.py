"""Where does the HIP decoder layer leave the emulating oracle at the 7B width? One decoder layer (H = 4096, I = 11008, 32 heads,
bench init), S rows, every operator run ON THE ORACLE'S OWN INTERMEDIATES (teacher forcing), so each line is the deviation that
ONE operator adds on identical inputs: rel-L2 vs the emulation (same bf16 storage points) and vs fp32.

    python tests/parity_ops_fullwidth.py [S=1088] [bf16|fp16]        (report; tests/test_gpu_parity_ops.py asserts the same numbers)

Lives under tests/ because it calls the oracle (test infrastructure): nothing outside tests/, smoke() and bench.py's cpu_baseline does.
"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vitron_oracle as O  # noqa: E402
from tests import fullwidth_util as FW  # noqa: E402
from vitron_amd import _lib, ops, synth  # noqa: E402


def rel(a, b):
    return FW.rel(a, b)


def measure(S=1088, operand="bf16"):
    """{operator: {vs_emu, emu_vs_fp32, ..}} for one 7B-width decoder layer on S rows, in the bf16 or the fp16 operand build."""
    _lib.load(operand=operand)
    odt = _lib.torch_dtype(operand)
    rnd = O.bf16_round if operand == "bf16" else O.fp16_store          # one operand store of the build under test
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    cfg = dict(synth.VICUNA_7B, num_hidden_layers=1)
    sd = synth.llama_state(cfg, synth.make_generator(7), w_std=0.02)
    w = {k: v.float() for k, v in sd.items()}
    H, heads, hd, I = 4096, 32, 128, 11008
    g = torch.Generator().manual_seed(8)
    x0 = O.bf16_round(torch.randn((S, H), generator=g) * 0.02)          # bf16 values: exact in fp16 too
    p = "model.layers.0."
    eps = 1e-5
    rep = {}
    # ---- oracle intermediates (emulation, and the fp32 value of the same op on the same input) --------------------------------
    y32 = O.rmsnorm(x0, w[p + "input_layernorm.weight"], eps)
    y = rnd(y32)
    wqkv = torch.cat([w[p + "self_attn.q_proj.weight"], w[p + "self_attn.k_proj.weight"], w[p + "self_attn.v_proj.weight"]], 0)
    qkv32 = y @ wqkv.t()
    qkv = rnd(qkv32)
    cos, sin = O.rope_tables(hd, S)
    q = qkv[:, :H].view(S, heads, hd).transpose(0, 1)
    k = qkv[:, H:2 * H].view(S, heads, hd).transpose(0, 1)
    v = qkv[:, 2 * H:].view(S, heads, hd).transpose(0, 1)
    qr32, kr32 = O._rope(q, cos, sin), O._rope(k, cos, sin)
    qr, kr = rnd(qr32), rnd(kr32)
    mask = torch.triu(torch.full((S, S), float("-inf")), diagonal=1)
    s = qr @ kr.transpose(-1, -2) / math.sqrt(hd) + mask
    pr = torch.exp(s - s.amax(-1, keepdim=True))
    o_emu = ((O.fp16_round(pr) @ v) / pr.sum(-1, keepdim=True)).transpose(0, 1).reshape(S, H)
    o32 = (torch.softmax(s, -1) @ v).transpose(0, 1).reshape(S, H)
    o = rnd(o_emu)
    x1 = x0 + o @ w[p + "self_attn.o_proj.weight"].t()
    h32 = O.rmsnorm(x1, w[p + "post_attention_layernorm.weight"], eps)
    h = rnd(h32)
    gu = torch.nn.functional.silu(h @ w[p + "mlp.gate_proj.weight"].t()) * (h @ w[p + "mlp.up_proj.weight"].t())
    a = rnd(gu)
    x2 = x1 + a @ w[p + "mlp.down_proj.weight"].t()

    # ---- the same operators on the device, each fed the oracle's input ------------------------------------------------------------
    def d(t, dt=odt):
        return t.to(dev).to(dt).contiguous()

    rms1 = ops.rmsnorm(d(x0, torch.float32), d(w[p + "input_layernorm.weight"], torch.float32), eps, dtype=odt)
    rep["rmsnorm"] = dict(vs_emu=rel(rms1.float().cpu(), y), emu_vs_fp32=rel(y, y32))
    g_qkv = ops.gemm(d(y), d(wqkv), None, ops.EPI_BF16)
    rep["qkv_gemm"] = dict(vs_emu=rel(g_qkv.float().cpu(), qkv), emu_vs_fp32=rel(qkv, qkv32))
    npages = (S + 63) // 64
    kt = torch.zeros(npages * heads * 64 * hd, dtype=odt, device=dev)
    vt = torch.zeros(npages * heads * 64 * hd, dtype=torch.float16, device=dev)       # V^T pages hold fp16
    table = torch.arange(npages, dtype=torch.int32, device=dev)
    desc = torch.tensor([[0, S, S, 0]], dtype=torch.int32, device=dev)
    pos = torch.arange(S, dtype=torch.int32, device=dev)
    xq = d(qkv)
    ops.kv_tiles(xq, 0, H, 2 * H, kt, vt, table, desc, npages, heads, hd, cos.to(dev).contiguous(), sin.to(dev).contiguous(), pos)
    q_dev = xq[:, :H].float().cpu().view(S, heads, hd).transpose(0, 1)
    kp = kt.view(npages, heads, 64, hd).permute(1, 0, 2, 3).reshape(heads, npages * 64, hd)[:, :S].float().cpu()
    rep["rope_q"] = dict(vs_emu=rel(q_dev, qr), emu_vs_fp32=rel(qr, qr32))
    rep["rope_k_pages"] = dict(vs_emu=rel(kp, kr), emu_vs_fp32=rel(kr, kr32))
    # attention on the ORACLE's rotated q / k / v (pages rebuilt from them so that the inputs are identical)
    kt2 = torch.zeros_like(kt).view(npages, heads, 64, hd)
    vt2 = torch.zeros_like(vt).view(npages, heads, hd, 64)
    kr_p = torch.zeros((heads, npages * 64, hd))
    kr_p[:, :S] = kr
    v_p = torch.zeros((heads, npages * 64, hd))
    v_p[:, :S] = v
    kt2.copy_(d(kr_p.view(heads, npages, 64, hd).permute(1, 0, 2, 3)))
    vt2.copy_(d(v_p.view(heads, npages, 64, hd).permute(1, 0, 3, 2), torch.float16))       # exact: v is bf16-rounded, |v| << 65504
    qin = torch.zeros((S, 3 * H))
    qin[:, :H] = qr.transpose(0, 1).reshape(S, H)
    att = ops.flash_attn(d(qin), kt2.view(-1), vt2.view(-1), table, desc, S, heads, hd, True, 1.0 / math.sqrt(hd))
    rep["flash_attn"] = dict(vs_emu=rel(att.float().cpu(), o), vs_fp32=rel(att.float().cpu(), o32), emu_vs_fp32=rel(o, o32),
                             vs_emu_unrounded=rel(att.float().cpu(), o_emu))
    x1_dev = ops.gemm(d(o), d(w[p + "self_attn.o_proj.weight"]), None, ops.EPI_F32_RESID, out=d(x0, torch.float32).clone())
    rep["o_proj_resid"] = dict(vs_emu=rel(x1_dev.cpu(), x1))
    rms2 = ops.rmsnorm(d(x1, torch.float32), d(w[p + "post_attention_layernorm.weight"], torch.float32), eps, dtype=odt)
    rep["rmsnorm2"] = dict(vs_emu=rel(rms2.float().cpu(), h), emu_vs_fp32=rel(h, h32))
    from vitron_amd.engine import interleave_gate_up
    wgu = interleave_gate_up(d(w[p + "mlp.gate_proj.weight"]), d(w[p + "mlp.up_proj.weight"])).contiguous()
    a_dev = ops.gemm(d(h), wgu, None, ops.EPI_SWIGLU_BF16)
    rep["swiglu_gemm"] = dict(vs_emu=rel(a_dev.float().cpu(), a), emu_vs_fp32=rel(a, gu))
    x2_dev = ops.gemm_resid_splitk(d(a), d(w[p + "mlp.down_proj.weight"]), d(x1, torch.float32).clone(), None, 0,
                                   torch.empty(8 * S * H, device=dev))
    rep["down_proj_resid"] = dict(vs_emu=rel(x2_dev.cpu(), x2))
    # ---- the whole layer, free running ------------------------------------------------------------------------------------------------
    from vitron_amd.engine import PackedLlama, PagedKVCache, SequenceState, llama_forward
    llama = PackedLlama(sd, cfg, dev, dtype=odt)
    kvc = PagedKVCache(llama, npages + 1)
    _, hid = llama_forward(llama, kvc, [SequenceState()], d(x0), [S], logit_rows=[S - 1], return_hidden=True)
    rep["whole_layer_hidden"] = dict(vs_emu=rel(hid.cpu(), x2))
    return rep


def main():
    rep = measure(int(sys.argv[1]) if len(sys.argv) > 1 else 1088, sys.argv[2] if len(sys.argv) > 2 else "bf16")
    for k_, v_ in rep.items():
        print(f"{k_:20s} " + json.dumps({a_: round(b_, 7) for a_, b_ in v_.items()}), flush=True)


if __name__ == "__main__":
    main()
