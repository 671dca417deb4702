"""Times vt_argmax and vt_sample_top_p on decode-sized logits ([4, 32000] fp32)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402

_lib.load()
dev = torch.device("cuda:0")
for rows in (1, 4, 16):
    lg = torch.randn((rows, 32000), device=dev) * 3
    for name, fn in (("argmax", lambda: ops.argmax(lg)), ("top_p 0.7", lambda: ops.sample_top_p(lg, 0.2, 0.7, 1, 2)),
                     ("top_p 1.0", lambda: ops.sample_top_p(lg, 1.0, 1.0, 1, 2))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fn()
        b.record()
        torch.cuda.synchronize()
        print(f"rows={rows:2d} {name:10s} {a.elapsed_time(b) / 50 * 1e3:8.1f} us")
