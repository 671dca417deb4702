"""A/B of tile configurations of the MFMA GEMM on given shapes (measurement helper, not part of the product path).

    python tools/cfg_ab.py "1088,12288,4096,BF16;1088,22016,4096,SWIGLU_BF16" 0,13,16 [iters]

Per shape: us per launch for every cfg id (0 = the dispatcher's choice), weights rotated over enough copies (>= 4, >= 600 MB in total)
that no launch streams its weights from the 256 MB Infinity Cache, interleaved in rounds (cfg a, cfg b, .. repeated) so that clock drift hits all of them alike; the
bf16 / fp32 outputs of every cfg are compared bit for bit with the first one's (same fragments, same accumulation order)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402


def main():
    shapes = [s.split(",") for s in sys.argv[1].split(";") if s]
    cfgs = [int(c) for c in sys.argv[2].split(",")]
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    _lib.load()
    dev = torch.device("cuda:0")
    for M, N, K, epi_name in shapes:
        M, N, K = int(M), int(N), int(K)
        epi = getattr(ops, "EPI_" + epi_name)
        a = torch.randn((M, K), device=dev).bfloat16()
        nw = max(4, -(-600_000_000 // (N * K * 2)))
        ws = [(torch.randn((N, K), device=dev) * 0.02).bfloat16() for _ in range(nw)]
        resid = torch.randn((M, N), device=dev) if epi_name == "F32_RESID" else None
        outs, us = {}, {c: 0.0 for c in cfgs}
        for c in cfgs:
            o = ops.gemm(a, ws[0], None, epi, out=resid.clone() if resid is not None else None, cfg=c)
            outs[c] = o.clone()
        torch.cuda.synchronize()
        out = resid.clone() if resid is not None else None
        rounds = 4
        for _ in range(rounds):
            for c in cfgs:
                for i in range(3):
                    ops.gemm(a, ws[i % nw], None, epi, out=out, cfg=c)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(iters):
                    ops.gemm(a, ws[i % nw], None, epi, out=out, cfg=c)
                e1.record()
                torch.cuda.synchronize()
                us[c] += e0.elapsed_time(e1) / iters * 1e3 / rounds
        row = {"shape": [M, N, K], "epi": epi_name, "plan": list(ops.gemm_plan(M, N, K, epi)),
               "us": {str(c): round(us[c], 1) for c in cfgs},
               "tflops": {str(c): round(2.0 * M * N * K / us[c] / 1e6, 1) for c in cfgs},
               "bit_equal_to_first": {str(c): bool(torch.equal(outs[c], outs[cfgs[0]])) for c in cfgs[1:]}}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
