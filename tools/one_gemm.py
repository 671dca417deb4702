"""The headline step's dominant launch alone -- gate/up of the C3 prefill, 5120 x 22016 x 4096 with the SwiGLU epilogue, as the dispatcher runs it
(gemm_w4_kernel<6, 8>) -- eight times over four weight copies, for counter passes:
    rocprofv3 --pmc FETCH_SIZE -f csv -d out -- python tools/one_gemm.py [bf16|fp16]
bench.py runs exactly this under rocprofv3 (two passes: FETCH_SIZE, WRITE_SIZE) to fill roofline.traffic live. Measurement helper."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import ops  # noqa: E402

dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn((5120, 4096), device=dev, generator=g).to(dt)
ws = [(torch.randn((22016, 4096), device=dev, generator=g) * 0.02).to(dt) for _ in range(4)]
for i in range(8):
    ops.gemm(a, ws[i % 4], None, ops.EPI_SWIGLU_BF16)
torch.cuda.synchronize()
