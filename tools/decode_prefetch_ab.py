"""Would a weight prefetcher running one GEMM ahead help the decode step? (measurement helper)

    python tools/decode_prefetch_ab.py [layers=32] [rows=4]

A decode step is ~130 launches of 7-32 us that each stream their weights from HBM once (PMC: fetched bytes = weight bytes); every launch
pays its own ramp-up and drain, so the step sees 5.0 of the 6.2 TB/s a long copy reaches. Idea under test: a second stream reads the NEXT
GEMM's weights (a plain streaming-read kernel, result discarded) while the current GEMM runs, so that the next GEMM finds them in the
256 MB Infinity Cache and HBM never idles between launches. Arms, alternated: (a) the four weight-streaming GEMMs of a decoder layer
(qkv, o_proj, gate/up SwiGLU, down_proj at M = rows) over `layers` distinct weight sets, back to back on one stream; (b) the same with the
prefetch stream one GEMM ahead (event-ordered: the read of GEMM k+1's weights starts when GEMM k is enqueued)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitron_amd import _lib, ops  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    _lib.load()
    dev = torch.device("cuda:0")
    H, I = 4096, 11008
    shapes = [("qkv", 3 * H, H, ops.EPI_BF16), ("o", H, H, ops.EPI_F32_RESID), ("gu", 2 * I, H, ops.EPI_SWIGLU_BF16), ("down", H, I, ops.EPI_F32_RESID)]
    ws = [[(torch.randn((N, K), device=dev) * 0.02).bfloat16() for (_, N, K, _) in shapes] for _ in range(L)]
    xs = {K: torch.randn((M, K), device=dev).bfloat16() for K in (H, I)}
    resid = torch.zeros((M, H), device=dev)
    seq = [(ws[l][i], shapes[i]) for l in range(L) for i in range(4)]
    side = torch.cuda.Stream(dev)
    sink = torch.zeros((1,), device=dev, dtype=torch.int32)

    def touch(w):          # a streaming read of the whole matrix (one pass, nothing written but 4 bytes)
        sink.add_(w.view(torch.int32).sum(dtype=torch.int32))

    def run(prefetch):
        main_s = torch.cuda.current_stream(dev)
        for j, (w, (_, N, K, epi)) in enumerate(seq):
            if prefetch and j + 1 < len(seq):
                ev = torch.cuda.Event()
                ev.record(main_s)                      # GEMM j is about to be enqueued: the read of GEMM j+1's weights may start
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    touch(seq[j + 1][0])
            ops.gemm(xs[K], w, None, epi, out=resid if epi == ops.EPI_F32_RESID else None)
        if prefetch:
            main_s.wait_stream(side)

    res = {}
    for arm in (False, True):
        run(arm)
    torch.cuda.synchronize()
    # both arms are captured into hipGraphs so that the comparison is device-side only (the host loop with one event per GEMM would
    # otherwise pace the prefetch arm)
    graphs = {}
    cap = torch.cuda.Stream(dev)
    for arm in (False, True):
        g = torch.cuda.CUDAGraph()
        cap.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cap):
            with torch.cuda.graph(g, stream=cap):
                run(arm)
        torch.cuda.current_stream(dev).wait_stream(cap)
        graphs[arm] = g
    torch.cuda.synchronize()
    t = {False: [], True: []}
    for _ in range(7):
        for arm in (False, True):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graphs[arm].replay()
            e1.record()
            torch.cuda.synchronize()
            t[arm].append(e0.elapsed_time(e1))
    wbytes = L * (4 * H * H + 3 * H * I) * 2
    for arm in (False, True):
        ms = sorted(t[arm])[len(t[arm]) // 2]
        res["with_prefetch" if arm else "plain"] = {"ms": round(ms, 3), "us_per_layer": round(ms / L * 1e3, 1), "weight_GBps": round(wbytes / ms / 1e6, 1)}
    # the streaming read alone (what the prefetcher costs when nothing else runs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for w, _ in seq:
        touch(w)
    e1.record()
    torch.cuda.synchronize()
    res["touch_only"] = {"ms": round(e0.elapsed_time(e1), 3), "GBps": round(wbytes / e0.elapsed_time(e1) / 1e6, 1)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
