"""Condense gemm_ab JSON lines (stdin) to one line per shape: cfg -> (TFLOP/s, max |diff| vs the first cfg)."""
import json
import sys

for l in sys.stdin:
    l = l.strip()
    if l.startswith('{"shape"'):
        d = json.loads(l)
        print("  ", d["shape"], d["epi"], {k: (v["tflops"], v["maxdiff_vs_first"]) if v else None for k, v in d["results"].items()})
    elif l and not l.startswith("[gpurun]") and "amdgpu.ids" not in l:
        print(l)
