"""A/B of the decoder prefill with the QKV epilogue fusion (vt_llama_model.qkv_fuse) off / on (measurement helper).

    python tools/qkv_fuse_ab.py [S=5120] [layers=8] [modes=0,1]

Vicuna-7B-shaped decoder with `layers` distinct layers (weights cold from HBM as in the benchmark step), one packed prefill of S rows per
pass, the modes alternated pass by pass on one box: 0 = QKV GEMM + vt_kv_tiles (rotary, K / V^T page writes), 1 = the fused epilogue
on the ping-pong kernel (vt_llama_model.qkv_fuse = 1). Prints ms per pass and per layer, and checks that the logits of every mode are
bit-identical to mode 0's. (Round 3 also measured a fused epilogue on the four-wave kernel with this tool: DESIGN.md 3.1.)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, synth  # noqa: E402
from vitron_amd.engine import PackedLlama, PagedKVCache, SequenceState, llama_forward  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    modes = [int(m) for m in (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(",")]
    _lib.load()
    dev = torch.device("cuda:0")
    cfg = dict(synth.VICUNA_7B, num_hidden_layers=L)
    llama = PackedLlama(synth.llama_state(cfg, synth.make_generator(5, dev), dev), cfg, dev)
    kv = PagedKVCache(llama, (S + 63) // 64 + 2)
    emb = (torch.randn((S, 4096), device=dev) * 0.02).bfloat16()

    def once(mode):
        llama.set_qkv_fuse(mode)
        seq = SequenceState()
        out = llama_forward(llama, kv, [seq], emb, [S])
        kv.release(seq.pages)
        return out

    ref = {m: once(m).clone() for m in modes}
    torch.cuda.synchronize()
    ms = {m: [] for m in modes}
    for _ in range(6):
        for m in modes:
            once(m)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                once(m)
            e1.record()
            torch.cuda.synchronize()
            ms[m].append(e0.elapsed_time(e1) / 3)
    llama.set_qkv_fuse(0)
    med = {m: sorted(v)[len(v) // 2] for m, v in ms.items()}
    print(json.dumps({"S": S, "layers": L, "ms_per_pass": {str(m): round(v, 3) for m, v in med.items()},
                      "us_per_layer_vs_mode0": {str(m): round((med[m] - med[modes[0]]) / L * 1e3, 1) for m in modes},
                      "logits_bit_equal_to_mode0": {str(m): bool(torch.equal(ref[m], ref[modes[0]])) for m in modes}}), flush=True)


if __name__ == "__main__":
    main()
