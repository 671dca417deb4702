"""bench.py -- end-to-end multimodal prefill throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic clip: pixels already in HBM -> LanguageBind video tower
(ViT-L/14 @336, temporal attention over 8 frames, the 23 layers hidden_states[-2] needs) -> mm_projector -> splice
with the 512-token prompt (S = 8*576 + 512 = 5120) -> 32-layer Vicuna-7B-shaped decoder prefill on a paged KV cache ->
last-position logits -> greedy first token. Everything runs through libvitron_hip.so.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --gpus N ...        (plain python, N > 1: re-launches itself under torch.distributed.run on a free port; on a box
                                       with fewer than N GPUs it prints one JSON error line and exits 2)

N > 1: one process per GPU, one clip per rank (weak scaling). Clips are encoded on the rank that owns them, the
visual tokens are exchanged with ONE RCCL all-gather over xGMI (BASELINE config 4) that overlaps the rank's own
prefill (started asynchronously, waited for before the step ends), and every rank prefills its own sequence. value = (tokens of all ranks) / (max over ranks of the timed region).

Prints ONE JSON line (rank 0) with the driver's contract + "roofline" (dominant kernel class = the MFMA tile GEMM,
timed live with HIP events on the kernel's stream) + "cpu_baseline" (the CPU oracle on the whole workload at full depth) and,
at N = 1, the secondary objects config.c2 (BASELINE configs[1]), config.c4 (configs[3], strong scaling) and decode (configs[4]).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def algorithmic_flops(S, n_vis_tokens, frames, N_vit, image_tokens, temporal=True):
    """SURVEY.md 8(d) accounting: 2 FLOP/MAC, ViT 23 layers, causal attention S(S+1)/2, lm_head last row only.
    temporal=False: the image tower (no temporal attention block)."""
    D, I_v, H, I, L, V = 1024, 4096, 4096, 11008, 32, 32000
    rows = frames * N_vit
    vit_lin = rows * 23 * (24 * D * D + (8 * D * D if temporal else 0))   # spatial qkv/o + mlp (24 D^2) + temporal qkv/o (8 D^2)
    vit_att = rows * 23 * (4 * N_vit * D + (4 * frames * D if temporal else 0))
    patch = frames * image_tokens * 2 * 588 * D
    proj = n_vis_tokens * 2 * (D * H + H * H)
    llm_lin = S * L * 2 * (4 * H * H + 3 * H * I)
    llm_att = L * 4 * H * S * (S + 1) / 2
    head = 2 * H * V
    return dict(vit=vit_lin + vit_att + patch, projector=proj, llm_linear=llm_lin, llm_attention=llm_att, lm_head=head,
                total=vit_lin + vit_att + patch + proj + llm_lin + llm_att + head)


def pmc_traffic_per_launch():
    """HBM-side bytes per launch of the MFMA tile GEMM class from the committed rocprofv3 PMC passes of THIS round's build
    (profiles/r<N>_pmc_traffic.json, newest round first: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command, summarised by
    tools/pmc_summarize.py; the file records the commit it was measured on). Units are KiB; FETCH_SIZE is doubled as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B); the counters sit on the L2's
    fabric side, so Infinity-Cache hits are included. PMC counters cannot be collected inside this process: the value is a
    committed measurement, labelled as such on the JSON line; (None, None, None) when no file is there."""
    path = next((p_ for p_ in (os.path.join(ROOT, "profiles", f"r{r}_pmc_traffic.json") for r in range(9, 1, -1)) if os.path.exists(p_)), None)
    if path is None:
        return None, None, None
    with open(path) as f:
        d = json.load(f)
    tot, n = 0.0, 0
    for k, v in d.items():
        if k.startswith(("gemm_bt_kernel", "gemm_p8_kernel", "gemm_w4_kernel", "gemm_w4r_kernel", "gemm_rp_kernel")) \
                and isinstance(v, dict) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            l = v["FETCH_SIZE"]["launches"]
            tot += l * (2.0 * v["FETCH_SIZE"]["mean_per_launch"] + v["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
            n += l
    return (tot / n if n else None), d.get("_measured_on", "unknown build"), os.path.relpath(path, ROOT)


def live_pmc_traffic(dtype):
    """HBM-side bytes of ONE launch of the step's dominant kernel -- gate/up, 5120 x 22016 x 4096, gemm_w4_kernel<6, 8> -- collected IN THIS RUN
    (VERDICT r5 #9): two child processes of this command, `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, counters
    only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), each over tools/one_gemm.py. KiB units, FETCH_SIZE doubled (gfx950: 128-byte
    requests tallied at 64). Returns (bytes per launch, note) or (None, reason) -- the caller then falls back to the committed file and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as d:
                r = subprocess.run([exe, "--pmc", counter, "-f", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "tools", "one_gemm.py"), dtype],
                                   cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, capture_output=True, text=True, timeout=180)
                tot, n = 0.0, 0
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if "gemm_w4_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                            tot += float(row["Counter_Value"])
                            n += 1
                if n == 0:
                    return None, f"rocprofv3 --pmc {counter} returned no gemm_w4_kernel rows (rc {r.returncode}): {r.stderr[-200:]!r}"
                vals[counter] = tot / n
    except Exception as e:  # noqa: BLE001  (a profiler that cannot run must not take the benchmark line with it)
        return None, f"{type(e).__name__}: {e}"
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, \
        "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child processes of this run) over tools/one_gemm.py, mean of 8 launches"


def pmc_traffic_decode_per_launch():
    """The same for the weight-streaming GEMM of the decode flow (profiles/r<N>_pmc_traffic_decode.json: separate --pmc FETCH_SIZE /
    WRITE_SIZE passes over tools/decode_bench.py 64): mean HBM-side bytes per gemm_skinny_dma_kernel launch."""
    path = next((p_ for p_ in (os.path.join(ROOT, "profiles", f"r{r}_pmc_traffic_decode.json") for r in range(9, 0, -1)) if os.path.exists(p_)), None)
    if path is None:
        return None, None
    with open(path) as f:
        d = json.load(f)
    tot, n = 0.0, 0
    for k, v in d.items():
        if k.startswith("gemm_skinny") and isinstance(v, dict) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            l = v["FETCH_SIZE"]["launches"]
            tot += l * (2.0 * v["FETCH_SIZE"]["mean_per_launch"] + v["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
            n += l
    return (tot / n if n else None), os.path.relpath(path, ROOT)


def committed_c4_n1():
    """tokens/s of config.c4 at N = 1 from the newest committed bench line (profiles/r<N>_bench.json): the denominator of the
    strong-scaling ratio SURVEY.md 8(e) asks for, so that a multi-GPU line carries it next to the weak-scaling headline."""
    for r in range(9, 1, -1):   # newest round first
        path = os.path.join(ROOT, "profiles", f"r{r}_bench.json")
        if os.path.exists(path):
            try:
                with open(path) as f:
                    d = json.load(f)
                c4 = d.get("config", {}).get("c4", {})
                if d.get("n_gpus") == 1 and c4.get("tokens_per_s"):
                    return {"tokens_per_s": c4["tokens_per_s"], "source": os.path.relpath(path, ROOT)}
            except (OSError, ValueError):
                pass
    return None


def cpu_reference_measurement():
    """The measured CPU baseline: the reference's own modules at full depth on this workload, run ONCE in the build container
    (tests/golden/make_golden_fulldepth.py -> profiles/r2_cpu_reference.json). None when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r2_cpu_reference.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def _alias_layers(sd_one, prefix_fmt, n_layers):
    """State dict whose `n_layers` layers all point at layer 0's tensors (same shapes, same FLOPs, 1/n of the host memory)."""
    out = {}
    for k, v in sd_one.items():
        if k.startswith(prefix_fmt.format(0)):
            for l in range(n_layers):
                out[prefix_fmt.format(l) + k[len(prefix_fmt.format(0)):]] = v
        else:
            out[k] = v
    return out


def cpu_baseline(image_size, frames, text_len, seed, mode="full", budget_s=420.0):
    """The CPU oracle (a port of the reference's algorithm, fp32) timed on this box's host cores.

    mode "full" (default): the WHOLE workload at FULL depth, as the reference runs it -- 24-layer video tower on the 8-frame clip
    (the reference computes layer 24 and discards it), projector on all visual tokens, 32 decoder layers on all S positions
    with eager S x S attention, final norm + lm_head on every position -- once. The 32 decoder layers (24 tower layers) reuse ONE
    layer's random weights: identical shapes, FLOPs and memory traffic per layer (0.8 GB of fp32 weights per layer, far beyond
    any cache), 3 GB of host memory instead of 29. The thread count is the fastest of a short calibration (PyTorch's CPU GEMMs stop
    scaling long before 256 hardware threads at these shapes). If the calibration predicts more than `budget_s` seconds, or in
    mode "sample", a bounded sample is timed and extrapolated instead (and says so)."""
    import torch

    from oracle import vitron_oracle as O
    from vitron_amd import synth

    t_start = time.perf_counter()
    ncpu = os.cpu_count() or 1
    gen = synth.make_generator(seed)
    G = image_size // 14
    n_vis = frames * G * G
    S = n_vis + text_len
    vcfg1 = dict(synth.VIT_L14, image_size=image_size, add_time_attn=True, num_frames=frames, num_hidden_layers=1)
    vsd1 = {k: v.float() for k, v in synth.vit_state(vcfg1, gen).items()}
    psd = {k: v.float() for k, v in synth.projector_state(1024, 4096, gen).items()}
    lcfg1 = dict(synth.VICUNA_7B, num_hidden_layers=1)
    lsd1 = {k: v.float() for k, v in synth.llama_state(lcfg1, gen).items()}
    clip = torch.randn((1, 3, frames, image_size, image_size), generator=gen).to(torch.bfloat16).float()
    emb_s = torch.randn((1, 1024, 4096), generator=gen) * 0.02
    # ---- calibration: one decoder layer on 1024 rows at a few thread counts
    cand = sorted({c for c in (16, 32, 64, 96, 128) if c <= ncpu} | {min(ncpu, 8)})
    cal = {}
    with torch.no_grad():
        for c in cand:
            torch.set_num_threads(c)
            O.llama_forward(lsd1, lcfg1, emb_s[:, :256], num_layers=1)          # warm the thread pool
            t0 = time.perf_counter()
            O.llama_forward(lsd1, lcfg1, emb_s, num_layers=1)
            cal[c] = time.perf_counter() - t0
    cores = min(cal, key=cal.get)
    torch.set_num_threads(cores)
    predicted = cal[cores] * (S / 1024.0) * 32 * 1.3                             # + towers, + the quadratic attention term
    if mode == "full" and predicted > budget_s:
        mode = f"sample (full depth predicted {predicted:.0f} s > {budget_s:.0f} s budget)"
    if mode == "full":
        vcfg = dict(vcfg1, num_hidden_layers=24)
        vsd = _alias_layers(vsd1, "encoder.layers.{}.", 24)
        lcfg = dict(lcfg1, num_hidden_layers=32)
        lsd = _alias_layers(lsd1, "model.layers.{}.", 32)
        with torch.no_grad():
            t0 = time.perf_counter()
            h = O.vit_forward(vsd, vcfg, clip, 24)                               # all 24 layers, like CLIPEncoder.forward
            feats = h[:, 1:]                                                     # feature_select drops CLS (which hidden state feeds on does not change the time)
            t1 = time.perf_counter()
            vis = O.projector_forward(psd, feats.reshape(-1, 1024))
            t2 = time.perf_counter()
            emb = torch.cat([torch.randn((text_len, 4096), generator=gen) * 0.02, vis], 0).unsqueeze(0)
            assert emb.shape[1] == S
            logits, _ = O.llama_forward(lsd, lcfg, emb)                          # 32 layers, logits of all S positions
            t3 = time.perf_counter()
        total = t3 - t0
        return {"value": S / total, "unit": "tokens/s", "cores": cores, "kind": "port",
                "sample": (f"PORT (the oracle, not the reference's modules) with ALIASED layer weights -- the WHOLE workload at FULL depth, once: oracle fp32 on {cores} host threads (of {ncpu}; fastest of the calibration "
                           f"{ {c: round(v, 2) for c, v in cal.items()} } s per layer on 1024 rows): 24-layer video tower on the {frames}-frame {image_size}px clip, "
                           f"projector on {n_vis} visual tokens, 32 decoder layers + final norm + lm_head on all {S} positions (eager S x S attention, as the "
                           "reference runs it); the layers of a stack reuse one layer's random weights (same shapes / FLOPs / bytes per layer, "
                           "3 GB of host memory instead of 29: friendlier to the host's caches than 32 distinct layers would be -- a stand-in, "
                           "see reference_measured for the reference's own modules on another box)"),
                "seconds_total": total, "seconds_tower": t1 - t0, "seconds_projector": t2 - t1, "seconds_decoder": t3 - t2,
                "measured_seconds": time.perf_counter() - t_start}
    with torch.no_grad():
        t0 = time.perf_counter()
        O.vit_forward(vsd1, vcfg1, clip, 0)
        t1 = time.perf_counter()
        O.vit_forward(vsd1, vcfg1, clip, 1)
        t2 = time.perf_counter()
        t_vit = (t1 - t0) + 24 * max((t2 - t1) - (t1 - t0), 0.0)
        feats = torch.randn((n_vis, 1024), generator=gen)
        t3 = time.perf_counter()
        O.projector_forward(psd, feats)
        t_proj = time.perf_counter() - t3
        Ss = 1024
        t4 = time.perf_counter()
        O.llama_forward(lsd1, lcfg1, emb_s, num_layers=0)        # final norm + lm_head on every position (as the reference does)
        t5 = time.perf_counter()
        O.llama_forward(lsd1, lcfg1, emb_s, num_layers=1)
        t6 = time.perf_counter()
        t_head = (t5 - t4) * (S / Ss)
        t_layer = max((t6 - t5) - (t5 - t4), 0.0) * (S / Ss)
        t_llm = t_head + 32 * t_layer
    total = t_vit + t_proj + t_llm
    return {"value": S / total, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": (f"{mode}: oracle fp32 on {cores} host threads (of {ncpu}): ViT embeddings + 1 of 24 layers on the full {frames}-frame {image_size}px clip, "
                       f"projector on all {n_vis} visual tokens, final-norm+lm_head and 1 of 32 decoder layers on the first {Ss} of {S} "
                       "positions; extrapolated linearly in depth and sequence length (attention's quadratic term is under-counted, "
                       "which flatters the CPU)"),
            "est_seconds_per_step": total, "measured_seconds": time.perf_counter() - t_start}


def cpu_baseline_reference(dev, image_sizes, frames, text_len, seed, budget_s=330.0):
    """kind "reference" (SURVEY.md 8(d) "CPU baseline timed beside it"): the REFERENCE's OWN modules on THIS box's host
    cores -- LanguageBind video tower (24 layers, as the reference runs them) + mlp2x_gelu projector + prepare_inputs_labels_for_multimodal
    + LlavaLlamaForCausalLM.forward (32 layers, eager S x S attention, logits of all positions), fp32, full depth, 32 DISTINCT decoder
    layers (27 GB of fp32 weights: no cache flattery), run once per image size. The modules come from oracle/_ref/vitron_ref.zip, staged
    from /root/reference by __graft_entry__.build() (oracle/stage_ref.py), imported through oracle/ref_shim.py. Returns None when
    neither the archive nor the reference tree is there (the caller falls back to the port)."""
    import torch

    from oracle import ref_model, ref_shim
    from vitron_amd import synth

    if not ref_shim.available():
        return None
    t_start = time.perf_counter()
    ncpu = os.cpu_count() or 1
    ns = ref_shim.install()
    # weights: counter-based streams evaluated on the GPU (seconds), moved to the host tensor by tensor
    model = ref_model.build_decoder(ns, synth.VICUNA_7B, synth.llama_state(synth.VICUNA_7B, synth.HashGenerator(seed + 10), dev))
    psd = {k: v.cpu() for k, v in synth.projector_state(1024, 4096, synth.HashGenerator(seed + 12), dev).items()}
    t_build = time.perf_counter() - t_start
    # thread count: fastest of a short calibration on the MLP GEMM shape (PyTorch's CPU GEMMs stop scaling long before 256 threads)
    a, w = torch.randn((1024, 4096)), torch.randn((11008, 4096))
    cal = {}
    for c in sorted({c for c in (16, 32, 64, 96, 128) if c <= ncpu} | {min(ncpu, 8)}):
        torch.set_num_threads(c)
        torch.matmul(a, w.t())
        t0 = time.perf_counter()
        for _ in range(3):
            torch.matmul(a, w.t())
        cal[c] = (time.perf_counter() - t0) / 3
    cores = min(cal, key=cal.get)
    torch.set_num_threads(cores)
    res = {}
    for size in sorted(image_sizes):                       # smallest first: its time decides whether the larger one fits the budget
        G = size // 14
        S = frames * G * G + text_len
        if res:
            (s0, r0), = list(res.items())[-1:]
            predicted = r0["seconds_total"] * (S / r0["S"]) * (1.0 + 0.15 * S / 5120)
            if time.perf_counter() - t_start + predicted > budget_s:
                res[size] = {"S": S, "skipped": f"predicted {predicted:.0f} s on top of {time.perf_counter() - t_start:.0f} s spent exceeds the {budget_s:.0f} s budget of this leg"}
                continue
        vcfg = dict(synth.VIT_L14, image_size=size, add_time_attn=True, num_frames=frames)
        ref_model.attach(ns, model, vcfg, {k: v.cpu() for k, v in synth.vit_state(vcfg, synth.HashGenerator(seed + 11), dev).items()}, psd)
        g = torch.Generator().manual_seed(seed + size)
        clip = torch.randn((3, frames, size, size), generator=g).to(torch.bfloat16).float()
        ids = torch.cat([torch.tensor([1]), torch.full((frames,), -200), torch.randint(3, 32000, (text_len - 1,), generator=g)]).unsqueeze(0)
        logits, embeds, t_front, t_dec = ref_model.prefill(model, ids, clip)
        assert embeds.shape[1] == S and logits.shape[1] == S
        res[size] = {"S": S, "tokens_per_s": S / (t_front + t_dec), "seconds_total": t_front + t_dec,
                     "seconds_towers_projector_splice": t_front, "seconds_decoder": t_dec}
        del logits, embeds
    del model
    return {"cores": cores, "host_threads_available": ncpu, "calibration_s_per_mlp_gemm": {c: round(v, 4) for c, v in cal.items()},
            "modules_from": ref_shim.source(), "seconds_model_build": t_build, "by_image_size": res,
            "measured_seconds": time.perf_counter() - t_start}


def decode_report_synthetic(model, llama, dev, steps, batch=4, ctx=609):
    """Sub-field of the decode report (round 2's number, kept for continuity): BASELINE configs[4]-shaped decode -- `batch`
    sequences with a `ctx`-token context (576 visual + region + prompt tokens in the real flow; synthetic embedding rows
    here), then `steps` greedy steps: token embedding -> 32 decoder layers (weight-streaming GEMMs + fused decode attention
    on the paged KV) -> lm_head -> argmax, through the same vt_llama_forward. HBM-bound: bytes per step = all decoder
    weights + lm_head once + the K/V tiles of every sequence."""
    import torch

    from vitron_amd import _lib, ops, synth
    from vitron_amd.engine import DecodeState, SequenceState, llama_forward

    gen = synth.make_generator(777, dev)
    H, L, I, V = llama.H, llama.L, llama.I, llama.V
    model._ensure_kv(batch * ((ctx + steps + 8 + 63) // 64 + 1))
    seqs = [SequenceState() for _ in range(batch)]
    emb = (torch.randn((batch * ctx, H), generator=gen, device=dev) * 0.02).to(llama.dtype)
    logits = llama_forward(llama, model.kv, seqs, emb, [ctx] * batch)
    state = DecodeState(llama, model.kv, seqs, steps + 8)   # device-resident step state: what generate() runs
    tok = ops.argmax(logits)

    def one(tok):
        state.feed(tok)
        return ops.argmax(state.forward())

    for _ in range(4):
        tok = one(tok)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tok = one(tok)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    _lib.profile_begin()      # separate pass: per-launch events perturb the wall clock
    for _ in range(4):
        tok = one(tok)
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    for s_ in seqs:
        model.kv.release(s_.pages)
    wbytes = (L * (4 * H * H + 3 * H * I) + V * H) * 2
    kv_bytes = batch * (ctx + 4 + steps // 2) * L * 2 * H * 2
    gs = dict(prof["gemm_skinny"])
    ev_ms = event_pair_overhead_ms()
    gs["ms"] = max(gs["ms"] - gs["launches"] * ev_ms, 1e-9)        # minus the empty-event-pair cost per launch (see c5_report)
    return {"workload": f"batch {batch} greedy decode, context {ctx}+, Vicuna-7B-shaped decoder, paged KV (64-token pages), synthetic context rows",
            "steps": steps, "ms_per_step": dt * 1e3, "tokens_per_s": batch / dt,
            "roofline": {"bound": "hbm", "kernel": "gemm_skinny_dma_kernel (weight-streaming GEMM, M <= 16)",
                         "achieved": gs["work"] / (gs["ms"] * 1e-3) / 1e9 if gs["ms"] > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (gs["work"] / (gs["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if gs["ms"] > 0 else 0.0,
                         "avg_launch_ms": gs["ms"] / max(gs["launches"], 1), "launches_per_step": gs["launches"] / 4,
                         "algorithmic_mbytes_per_launch": gs["work"] / max(gs["launches"], 1) / 1e6},
            "whole_step_gbytes": (wbytes + kv_bytes) / 1e9, "whole_step_GBps": (wbytes + kv_bytes) / dt / 1e9,
            "whole_step_frac_of_hbm_peak": (wbytes + kv_bytes) / dt / 1e9 / HBM_PEAK_GBS,
            "kernel_ms_per_step": {k: max(v["ms"] - v["launches"] * ev_ms, 0.0) / 4 for k, v in prof.items() if v["launches"]}}


def c5_report(model, llama, dev, steps, seed):
    """BASELINE configs[4] as SURVEY.md 8(d) defines C5: 4 x (336 px image + box) + a short prompt through image tower ->
    region_extractor -> projector -> splice -> packed prefill, then `steps` greedy decode steps at batch 4 on the paged KV cache
    (64-token pages) -- through model.generate(), the reference's own entry point (app.py:562-571): device-resident decode state,
    weight-streaming GEMMs with RMSNorm folded in, fused rotary / append / attention kernel, on-device arg-max. Outside the
    headline's timed region. HBM-bound: bytes per step = all decoder weights + lm_head once + the K / V^T pages of every sequence."""
    import torch

    from vitron_amd import _lib, synth

    B = 4
    gen = synth.make_generator(seed + 55, dev)
    boxes = [[0, 0, 224, 224], [0, 58.9, 117.9, 117.9], [100, 20, 180, 200], [7, 7, 8, 8]]      # SURVEY.md 8(d), the reference's 224 canvas
    rnd = lambda k: torch.randint(3, 32000, (k,), generator=gen, device=dev).tolist()            # noqa: E731
    prompt = [1, -200] + rnd(6) + [-300, 1] + rnd(24)                                           # app.py:525-534 layout
    ids = torch.tensor([prompt] * B, device=dev)
    images = [torch.randn((3, 336, 336), generator=gen, device=dev).to(llama.dtype) for _ in range(B)]
    ctx = 576 + len(prompt) - 1
    reuse = getattr(model.config, "kv_prefix_reuse", True)
    model.config.kv_prefix_reuse = False
    try:
        run = lambda n: model.generate(ids, images=images, regions=boxes, do_sample=False, max_new_tokens=n, eos_token_id=-1)   # noqa: E731
        run(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        run(1)
        torch.cuda.synchronize()
        tp = time.perf_counter() - t1
        psteps = min(steps, 65)
        _lib.profile_begin()            # separate pass: per-launch events perturb the wall clock
        run(psteps)
        torch.cuda.synchronize()
        prof = _lib.profile_end()
    finally:
        model.config.kv_prefix_reuse = reuse
    assert out.shape == (B, len(prompt) + steps)
    dec = (dt - tp) / max(steps - 1, 1)
    H, L, I, V = llama.H, llama.L, llama.I, llama.V
    wbytes = (L * (4 * H * H + 3 * H * I) + V * H) * 2
    kv_bytes = B * (ctx + steps // 2) * L * 2 * H * 2
    gs = dict(prof["gemm_skinny"])
    # An event pair around an 8-32 us launch measures the launch PLUS the pair's own cost (the pairs summed to more than the
    # step itself). The empty-pair time is measured on this stream and subtracted per launch; the raw sum stays on the line.
    ev_ms = event_pair_overhead_ms()
    gs_raw_ms = gs["ms"]
    gs["ms"] = max(gs["ms"] - gs["launches"] * ev_ms, 1e-9)
    n_dec = psteps - 1                                       # decode passes inside the profiled generate (the prefill runs tile GEMMs)
    return {"workload": (f"BASELINE configs[4]: {B} x (336 px image + box) through image tower (ViT-L/14, 23 layers) + region_extractor + projector, "
                         f"splice ({ctx} context rows each), packed prefill, then {steps} greedy decode steps at batch {B} through generate(): "
                         "Vicuna-7B-shaped decoder, paged KV (64-token pages)"),
            "steps": steps, "batch": B, "context_rows": ctx, "prefill_ms": tp * 1e3, "ms_per_step": dec * 1e3, "tokens_per_s": B / dec,
            "roofline": {"bound": "hbm", "kernel": "gemm_skinny_dma_kernel (weight-streaming GEMM, M <= 16)",
                         "achieved": gs["work"] / (gs["ms"] * 1e-3) / 1e9 if gs["ms"] > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (gs["work"] / (gs["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if gs["ms"] > 0 else 0.0,
                         "avg_launch_ms": gs["ms"] / max(gs["launches"], 1), "launches_per_step": gs["launches"] / max(n_dec, 1),
                         "algorithmic_mbytes_per_launch": gs["work"] / max(gs["launches"], 1) / 1e6,
                         "event_pair_overhead_ms": ev_ms, "avg_launch_ms_raw_event_pair": gs_raw_ms / max(gs["launches"], 1),
                         "achieved_raw_event_pairs": gs["work"] / (gs_raw_ms * 1e-3) / 1e9 if gs_raw_ms > 0 else 0.0,
                         "traffic": pmc_traffic_decode_per_launch()[0],
                         "traffic_note": f"mean bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, KiB) from the committed rocprofv3 --pmc passes {pmc_traffic_decode_per_launch()[1]}; not collected live",
                         "note": (f"HIP-event pairs around the launches of a separate {psteps}-token generate() (incl. the region path's three weight-streaming "
                                  "GEMMs), minus the measured cost of an empty event pair per launch; the rocprofv3 kernel durations of the same flow "
                                  "(profiles/r<N>_decode_rocprofv3_kernel_stats.csv) are the cross-check")},
            "whole_step_gbytes": (wbytes + kv_bytes) / 1e9, "whole_step_GBps": (wbytes + kv_bytes) / dec / 1e9,
            "whole_step_frac_of_hbm_peak": (wbytes + kv_bytes) / dec / 1e9 / HBM_PEAK_GBS,
            "kernel_ms_per_decode_step": {k: max(v["ms"] - v["launches"] * ev_ms, 0.0) / max(n_dec, 1) for k, v in prof.items()
                                          if k in ("gemm_skinny", "attn_decode") and v["launches"]}}


def c2_report(model, llama, dev, seed, reps=10):
    """BASELINE configs[1]: ONE 336 x 336 image + 512-token prompt (576 + 512 = 1088 rows): image tower (ViT-L/14, 23 layers) ->
    projector -> splice -> 32-layer decoder prefill -> last-position logits -> greedy token. Outside the headline's timed region."""
    import torch

    from vitron_amd import _lib, ops, synth
    from vitron_amd.engine import SequenceState, llama_forward

    gen = synth.make_generator(seed + 77, dev)
    image = torch.randn((3, 336, 336), generator=gen, device=dev).to(llama.dtype)
    ids = torch.cat([torch.tensor([1, -200], device=dev), torch.randint(3, 32000, (511,), generator=gen, device=dev)]).unsqueeze(0)
    ids_host = ids.cpu()
    S = 576 + 512
    model._ensure_kv((S + 63) // 64 + 4)

    def step():
        (_, _, _, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [image], None, input_ids_host=ids_host)
        seq = SequenceState()
        tok = ops.argmax(llama_forward(llama, model.kv, [seq], embeds[0], [embeds.shape[1]]))
        model.kv.release(seq.pages)
        return embeds.shape[1]

    for _ in range(3):
        assert step() == S
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(reps):
        step()
        marks[k + 1].record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ev = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(reps))
    _lib.profile_begin()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    fl = algorithmic_flops(S, 576, 1, 577, 576, temporal=False)
    gt = prof["gemm_tile"]
    return {"workload": "BASELINE configs[1]: one 336x336 image (576 visual tokens) + 512-token prompt -> S=1088; LanguageBind image ViT-L/14 (23 of 24 layers) + "
                        "mlp2x_gelu projector + Vicuna-7B-shaped decoder prefill (32 layers, paged KV), last-position logits + greedy token",
            "S": S, "reps": reps, "ms_per_step": dt * 1e3, "ms_per_step_hipevent_median": ev[len(ev) // 2], "tokens_per_s": S / dt,
            "algorithmic_tflop_per_step": fl["total"] / 1e12, "end_to_end_frac_of_mfma_peak": fl["total"] / 1e12 / dt / MFMA_BF16_PEAK_TFLOPS,
            "gemm_class_tflops": gt["work"] / (gt["ms"] * 1e-3) / 1e12 if gt["ms"] > 0 else 0.0,
            "kernel_ms_per_step": {k: v["ms"] / 3 for k, v in prof.items() if v["launches"]}}


def prefill_report_224(model, llama, dev, seed, kind, reps, frames=8, text_len=512):
    """The REFERENCE-NATIVE image size (the reference's processors are hard-wired to 224 x 224 --
    image/processing_image.py:20-21, video/processing_video.py:50-51, region_extractor/layer.py:60 -- so N = 257 tokens per frame and
    G = 16 are the only shapes a real Vitron checkpoint runs). kind "clip": C3-224 = one 8-frame 224 px clip + 512 tokens, S = 2560,
    36.60 TFLOP; kind "image": C2-224 = one 224 px image + 512 tokens, S = 768, 10.27 TFLOP. Same full-work step as the headline
    (tower, projector, splice, 32 layers, last-position logits, arg-max; nothing cached), with a temporary 224 px tower of the same
    random init. Outside the headline's timed region."""
    import torch
    from types import SimpleNamespace

    from vitron_amd import _lib, ops, synth
    from vitron_amd.engine import SequenceState, llama_forward
    from vitron_amd.model.multimodal_encoder.builder import build_image_tower, build_video_tower

    clip_kind = kind == "clip"
    gen = synth.make_generator(seed + (224 if clip_kind else 225), dev)
    vcfg = dict(synth.VIT_L14, image_size=224, add_time_attn=clip_kind, num_frames=frames if clip_kind else 1)
    sel = model.config.mm_vision_select_layer
    if clip_kind:
        tower = build_video_tower(SimpleNamespace(mm_video_tower="synthetic224/LanguageBind_Video_merge", mm_vision_select_layer=sel), delay_load=True)
        slot = "video_tower"
    else:
        tower = build_image_tower(SimpleNamespace(mm_image_tower="synthetic224/LanguageBind_Image", mm_vision_select_layer=sel), delay_load=True)
        slot = "image_tower"
    tower._dtype = llama.dtype
    tower.init_synthetic(vcfg, gen, dev)
    n_img = frames if clip_kind else 1
    pix = torch.randn((3, frames, 224, 224) if clip_kind else (3, 224, 224), generator=gen, device=dev).to(llama.dtype)
    ids = torch.cat([torch.tensor([1], device=dev), torch.full((n_img,), -200, device=dev),
                     torch.randint(3, 32000, (text_len - 1,), generator=gen, device=dev)]).unsqueeze(0)
    ids_host = ids.cpu()
    S = n_img * 256 + text_len
    model._ensure_kv((S + 63) // 64 + 4)
    inner = model.get_model()
    keep = getattr(inner, slot, None)
    setattr(inner, slot, tower)
    try:
        def step():
            (_, _, _, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [pix], None, input_ids_host=ids_host)
            seq = SequenceState()
            ops.argmax(llama_forward(llama, model.kv, [seq], embeds[0], [embeds.shape[1]]))
            model.kv.release(seq.pages)
            return embeds.shape[1]

        for _ in range(3):
            assert step() == S
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks[0].record()
        for k in range(reps):
            step()
            marks[k + 1].record()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        ev = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(reps))
        _lib.profile_begin()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        prof = _lib.profile_end()
    finally:
        setattr(inner, slot, keep)
    fl = algorithmic_flops(S, n_img * 256, n_img, 257, 256, temporal=clip_kind)
    gt = prof["gemm_tile"]
    return {"workload": (f"{'BASELINE configs[2]' if clip_kind else 'BASELINE configs[1]'} at the reference-native 224 px: "
                         f"{'one 8-frame 224x224 clip (2048 visual tokens)' if clip_kind else 'one 224x224 image (256 visual tokens)'} + {text_len}-token prompt -> S={S}; "
                         f"LanguageBind {'video' if clip_kind else 'image'} ViT-L/14 (23 of 24 layers, N = 257) + mlp2x_gelu projector + Vicuna-7B-shaped decoder "
                         "prefill (32 layers, paged KV), last-position logits + greedy token"),
            "S": S, "reps": reps, "ms_per_step": dt * 1e3, "ms_per_step_hipevent_median": ev[len(ev) // 2], "tokens_per_s": S / dt,
            "algorithmic_tflop_per_step": fl["total"] / 1e12, "end_to_end_frac_of_mfma_peak": fl["total"] / 1e12 / dt / MFMA_BF16_PEAK_TFLOPS,
            "gemm_class_tflops": gt["work"] / (gt["ms"] * 1e-3) / 1e12 if gt["ms"] > 0 else 0.0,
            "kernel_ms_per_step": {k: v["ms"] / 3 for k, v in prof.items() if v["launches"]}}


def empirical_peaks(dev):
    """the vendor peaks next to what this box delivers on a library GEMM and a plain copy (measurement aids, not part
    of the product path): hipBLASLt bf16 8192^3 through torch.matmul (sustained ~0.5 s window, random operands) and a 1 GiB
    device-to-device copy (read + write bytes; the faster of hipMemcpy and an elementwise copy kernel)."""
    import torch
    res = {}
    try:
        n = 8192
        a = torch.randn((n, n), device=dev).to(torch.bfloat16)
        b = torch.randn((n, n), device=dev).to(torch.bfloat16)
        for _ in range(3):
            torch.matmul(a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = 0
        while True:
            for _ in range(20):
                torch.matmul(a, b)
            torch.cuda.synchronize()
            it += 20
            if time.perf_counter() - t0 > 0.5:
                break
        res["hipblaslt_bf16_gemm_8192_tflops"] = 2.0 * n ** 3 * it / (time.perf_counter() - t0) / 1e12
        del a, b
        src = torch.ones(1 << 28, dtype=torch.float32, device=dev)          # 1 GiB
        dst = torch.empty_like(src)
        best = 0.0
        for name, fn in (("memcpy", lambda: dst.copy_(src)), ("kernel", lambda: torch.mul(src, 1.0, out=dst))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            rate = 2.0 * (1 << 30) * 20 / (time.perf_counter() - t0) / 1e9          # bytes read + bytes written
            res[f"d2d_copy_{name}_GBps"] = rate
            best = max(best, rate)
        res["d2d_copy_GBps"] = best
        del src, dst
        # the matrix pipe ALONE under this box's power limit (vt_probe_mfma: one wave per SIMD, operands in registers, nothing but
        # v_mfma_f32_16x16x32 on the GEMM's 128 x 128 wave tile), on operand values distributed like the workload's (activations ~ N(0,1),
        # weights ~ N(0, 0.02^2)) and on zeros: what "MFMA peak" means on this part once the operands are not constant
        from vitron_amd import _lib
        lib = _lib.load(operand="bf16")
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        outp = torch.empty(ncu * 256, dtype=torch.float32, device=dev)
        iters = 4000
        flop = 2.0 * 128 * 128 * 64 * iters * 4 * ncu
        for name, fa, fb in (("workload_like_operands", lambda: torch.randn(65536 * 8, device=dev), lambda: torch.randn(65536 * 8, device=dev) * 0.02),
                             ("zero_operands", lambda: torch.zeros(65536 * 8, device=dev), lambda: torch.zeros(65536 * 8, device=dev))):
            a, b = fa().to(torch.bfloat16), fb().to(torch.bfloat16)
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                _lib.check(lib.vt_probe_mfma(a.data_ptr(), b.data_ptr(), outp.data_ptr(), iters, st), "vt_probe_mfma", lib)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 0.5:
                for _ in range(10):
                    lib.vt_probe_mfma(a.data_ptr(), b.data_ptr(), outp.data_ptr(), iters, st)
                torch.cuda.synchronize()
                n += 10
            res[f"mfma_only_bf16_tflops_{name}"] = flop * n / (time.perf_counter() - t0) / 1e12
        # what a kernel that does NOTHING BUT read sustains from HBM (vt_probe_read: 1 GiB, larger than the Infinity Cache; plain loads and the
        # read-once policy of the weight-streaming GEMM): the yardstick of the decode step, whose bytes are weights and KV pages read once
        buf = torch.ones(1 << 28, dtype=torch.float32, device=dev)
        flag = torch.zeros(4, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for nt, key in ((0, "hbm_read_kernel_GBps"), (1, "hbm_read_kernel_nt_GBps")):
            for _ in range(3):
                _lib.check(lib.vt_probe_read(buf.data_ptr(), buf.numel() * 4, nt, flag.data_ptr(), st), "vt_probe_read", lib)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                lib.vt_probe_read(buf.data_ptr(), buf.numel() * 4, nt, flag.data_ptr(), st)
            torch.cuda.synchronize()
            res[key] = (1 << 30) * 20 / (time.perf_counter() - t0) / 1e9
        del buf
    except Exception as e:  # noqa: BLE001 -- a measurement aid must not take the benchmark down
        res["error"] = f"{type(e).__name__}: {e}"
    return res



def c4_report(model, llama, dev, world, rank, use_dist, dist, frames, image_size, text_len, steps=2):
    """BASELINE configs[3] as SURVEY.md 8(d)/(e) define it: a FIXED global batch of 8 clips, clip-sharded over the N ranks
    (8/N clips per GPU; all 8 through the one GPU at N = 1), each rank encodes its clips, the projected visual tokens of all
    clips are all-gathered (asynchronously: a rank's own prefill only needs its own tokens), and every rank prefills its own
    8/N sequences in ONE packed decoder pass. tokens/s = 8 * S / max-over-ranks time: comparable across N (strong scaling).
    Runs outside the headline's timed region."""
    import torch

    from vitron_amd import ops, synth
    from vitron_amd.engine import SequenceState, llama_forward
    from vitron_amd.parallel import start_all_gather_visual_tokens

    GLOBAL = 8
    if GLOBAL % world:
        return {"skipped": f"8 clips do not shard evenly over {world} ranks"}
    n_local = GLOBAL // world
    G = image_size // 14
    S = frames * G * G + text_len
    gen = synth.make_generator(9000 + rank, dev)
    clips = [torch.randn((3, frames, image_size, image_size), generator=gen, device=dev).to(llama.dtype) for _ in range(n_local)]
    text = torch.randint(3, 32000, (n_local, text_len - 1), generator=gen, device=dev)
    ids = torch.cat([torch.ones((n_local, 1), dtype=torch.long, device=dev), torch.full((n_local, frames), -200, device=dev), text], 1)
    ids_host = ids.cpu()
    model._ensure_kv(n_local * ((S + 63) // 64 + 1) + 4)
    pending = []
    orig = model.encode_videos
    if use_dist:
        def enc(videos):
            f = orig(videos)
            pending.append(start_all_gather_visual_tokens(f))      # [n_local, T, P, H] -> all GLOBAL clips, overlapping the prefill
            return f
        model.encode_videos = enc
    try:
        def step():
            (_, _, _, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, clips, None,
                                                                                 input_ids_host=ids_host)
            seqs = [SequenceState() for _ in range(n_local)]
            logits = llama_forward(llama, model.kv, seqs, embeds.reshape(-1, embeds.shape[-1]), [embeds.shape[1]] * n_local)
            tok = ops.argmax(logits)
            for q in seqs:
                model.kv.release(q.pages)
            for g in pending:
                assert g.wait().shape[0] == GLOBAL
            pending.clear()
            return tok

        def fence():
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
                torch.cuda.synchronize()

        step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
    finally:
        model.encode_videos = orig
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return {"workload": f"BASELINE configs[3]: fixed global batch of {GLOBAL} 8-frame {image_size}px clips + {text_len}-token prompts, "
                        f"clip-sharded over {world} GPU(s), all-gather of the visual tokens, packed data-parallel prefill",
            "global_clips": GLOBAL, "clips_per_gpu": n_local, "tokens_per_clip": S, "steps": steps,
            "ms_per_step": dt / steps * 1e3, "tokens_per_s": GLOBAL * S * steps / dt, "scaling": "strong",
            "gather": "async RCCL all-gather overlapped with the rank's own prefill" if use_dist else "none (one GPU holds every clip)"}


def c4_uneven_report(model, llama, dev, world, rank, dist, frames, image_size, steps=2):
    """The case that NEEDS the exchange: the same 8 clips with prompts of UNEVEN length (128 .. 896 tokens, mean
    512: the same total as configs[3]). Clips are encoded where shard_range puts them; sequences are prefilled where
    vitron_amd.parallel.plan_prefill_placement puts them (balanced by prefill cost), so a rank prefills sequences whose visual tokens reached it
    over the all-gather -- the gathered tensor feeds the splice of the sequences this rank owns. N > 1 only; wrapped by the caller."""
    import torch

    from vitron_amd import ops, synth
    from vitron_amd.engine import SequenceState, llama_forward
    from vitron_amd.parallel import (all_gather_visual_tokens, plan_prefill_placement, sequences_of_rank, shard_range, visual_tokens_for_rank)

    GLOBAL = 8
    text_lens = [128, 896, 256, 768, 384, 640, 512, 512]
    G = image_size // 14
    nvis = frames * G * G
    place = plan_prefill_placement([nvis + t for t in text_lens], world)
    mine = sequences_of_rank(place, rank)
    es, ee = shard_range(GLOBAL, world, rank)
    gen = synth.make_generator(9100, dev)                                   # ONE stream on every rank: every rank knows every prompt
    clips_all = [torch.randn((3, frames, image_size, image_size), generator=gen, device=dev).to(llama.dtype) for _ in range(GLOBAL)]
    ids_all = [torch.cat([torch.ones((1,), dtype=torch.long, device=dev), torch.full((frames,), -200, device=dev),
                          torch.randint(3, 32000, (t - 1,), generator=gen, device=dev)]).unsqueeze(0) for t in text_lens]
    ids_host = [i.cpu() for i in ids_all]
    lens = [nvis + text_lens[i] for i in mine]
    model._ensure_kv(sum((l + 63) // 64 + 1 for l in lens) + 4)
    orig = model.encode_videos
    foreign = [i for i in mine if not es <= i < ee]

    def step():
        local = orig(torch.stack(clips_all[es:ee])) if ee > es else None      # this rank's clips: [n_local, T, P, H]
        allv = all_gather_visual_tokens(local, GLOBAL)                          # blocking: the prefill below consumes it
        feats = visual_tokens_for_rank(allv, place, rank)
        rows = []
        try:
            for k, i in enumerate(mine):
                model.encode_videos = lambda videos, f=feats[k:k + 1]: f        # the splice takes the GATHERED tokens of sequence i
                (_, _, _, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids_all[i], None, None, None, None, [clips_all[i]], None,
                                                                                     input_ids_host=ids_host[i])
                rows.append(embeds[0])
        finally:
            model.encode_videos = orig
        if not rows:
            return None
        seqs = [SequenceState() for _ in mine]
        tok = ops.argmax(llama_forward(llama, model.kv, seqs, torch.cat(rows, 0), lens))
        for q in seqs:
            model.kv.release(q.pages)
        return tok

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    total = sum(nvis + t for t in text_lens)
    return {"what": "8 clips, prompts of 128 .. 896 tokens: clips encoded by shard_range, sequences prefilled by cost-balanced placement -- the gathered "
                    "visual tokens are what the prefilling rank splices (blocking all-gather: it is on the critical path here)",
            "placement": place, "sequences_prefilled_away_from_their_encoder_on_rank0": foreign if rank == 0 else None,
            "steps": steps, "ms_per_step": float(dt.item()) / steps * 1e3, "tokens_per_s": total * steps / float(dt.item())}


def gather_report(dev, world, dist, frames, image_size, iters=10):
    """The exchange step alone (blocking, nothing to hide behind): RCCL's all-gather collective vs the direct full-mesh
    point-to-point exchange (vitron_amd.parallel.all_gather_direct_p2p), on the per-rank message of BASELINE configs[3]
    (one clip's projected visual tokens, 37.75 MB at 336 px)."""
    import torch

    from vitron_amd.parallel import all_gather_direct_p2p, start_all_gather_visual_tokens

    G = image_size // 14
    local = torch.randn((1, frames, G * G, 4096), device=dev).to(torch.bfloat16)     # 2 bytes per element in either operand format
    res = {"message_mbytes_per_rank": local.numel() * 2 / 1e6}
    for name, fn in (("rccl_all_gather_ms", lambda: start_all_gather_visual_tokens(local).wait()),
                     ("direct_p2p_ms", lambda: all_gather_direct_p2p(local))):
        try:
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                out = fn()
            torch.cuda.synchronize()
            dt = torch.tensor([(time.perf_counter() - t0) / iters * 1e3], dtype=torch.float64, device=dev)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            res[name] = float(dt.item())
            assert out.shape[0] == world
        except Exception as e:  # noqa: BLE001 -- a transport the installed RCCL refuses must not take the benchmark down
            res[name] = f"failed: {type(e).__name__}: {e}"
    return res


def fp16_ab_report(args, dev, model_bf16, step_bf16, ids, ids_host, clip_bf16, vit_image=None):
    """The headline workload on the fp16-operand build (libvitron_hip_f16.so: the reference's own inference dtype), same box, same
    seed, same inputs, arms alternated (bf16, fp16, bf16, fp16; K steps each): does the step time move, and how far are the two
    builds' last-position logits apart. Outside the timed region of the headline; BASELINE's dtype stays bf16."""
    import torch

    from vitron_amd import ops, synth
    from vitron_amd.engine import SequenceState, llama_forward, pair_lo
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM

    G = args.image_size // 14
    S = args.frames * G * G + args.text_len
    m16 = LlavaLlamaForCausalLM(LlavaConfig(**synth.VICUNA_7B, mm_hidden_size=1024))
    # (the same tower list as the headline model: the weights come from ONE seeded stream, so an extra tower shifts everything behind it)
    m16.init_synthetic(dev, seed=args.seed, vit_image=vit_image,
                       vit_video=dict(synth.VIT_L14, image_size=args.image_size, add_time_attn=True, num_frames=args.frames), dtype=torch.float16)
    l16 = m16.get_model().llama
    m16._ensure_kv((S + 63) // 64 + 4)
    clip16 = clip_bf16.to(torch.float16)              # bf16 values are exact in fp16: identical pixels

    def step16():
        (_, _, _, _, embeds, _) = m16.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [clip16], None, input_ids_host=ids_host)
        seq = SequenceState()
        lo = pair_lo(embeds)                      # precise level 2: the spliced embeddings are an operand pair
        logits = llama_forward(l16, m16.kv, [seq], embeds[0], [embeds.shape[1]], embeds_lo=None if lo is None else lo[0])
        tok = ops.argmax(logits)
        m16.kv.release(seq.pages)
        return tok, logits

    def timed(fn, k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3, out

    for _ in range(2):
        step16()
    K = args.fp16_ab_steps
    arms = {"bf16": [], "fp16": []}
    lb = l16_ = None
    for _ in range(2):
        ms, out = timed(step_bf16, K)
        arms["bf16"].append(ms)
        lb = out[2]
        ms, out = timed(step16, K)
        arms["fp16"].append(ms)
        l16_ = out[1]
    d = float((l16_.double() - lb.double()).norm() / lb.double().norm())
    # the fp16 build's precise modes on the same workload (round 5): precise_qk = q / k path as operand pairs; precise2 = the verification
    # mode (every GEMM A operand a pair: full-depth logits 4.5e-4 from the reference's fp32, tests/test_gpu_parity_fulldepth.py)
    f16_mean = sum(arms["fp16"]) / len(arms["fp16"])
    precise = {}
    for level, tag in ((1, "precise_qk"), (3, "precise3"), (2, "precise2")):
        m16.set_precise(level)
        try:
            step16()
            ms, out_p = timed(step16, K if level == 3 else max(2, K // 2))
        finally:
            m16.set_precise(0)
        precise[tag] = {"ms_per_step": ms, "over_fp16": ms / f16_mean,
                        "last_position_logits_rel_l2_vs_fp16_standard": float((out_p[1].double() - l16_.double()).norm() / l16_.double().norm())}
        lp16 = out_p[1]                                # (the last one: level 2)
    for tag in ("precise_qk", "precise3"):
        precise[tag]["last_position_logits_rel_l2_vs_fp16_precise2"] = None
    # (level 3's distance from level 2 on this workload: the two modes that meet 1e-3 against the reference must agree far inside it)
    m16.set_precise(3)
    try:
        l3 = step16()[1]
        precise["precise3"]["last_position_logits_rel_l2_vs_fp16_precise2"] = float((l3.double() - lp16.double()).norm() / lp16.double().norm())
    finally:
        m16.set_precise(0)
    # the bf16 (benchmark) build in the verification mode: two libraries compiled from the same sources for different operand formats,
    # 1.2e-2 apart in the standard mode, must land on the same logits (each is ~4e-4 from the reference's fp32 at full depth)
    bf16_mean = sum(arms["bf16"]) / len(arms["bf16"])
    model_bf16.set_precise(2)
    try:
        step_bf16()
        ms, out_b = timed(step_bf16, max(2, K // 2))
    finally:
        model_bf16.set_precise(0)
    bf16_p2 = {"ms_per_step": ms, "over_bf16": ms / bf16_mean,
               "last_position_logits_rel_l2_vs_fp16_precise2": float((out_b[2].double() - lp16.double()).norm() / lp16.double().norm()),
               "last_position_logits_rel_l2_vs_bf16_standard": float((out_b[2].double() - lb.double()).norm() / lb.double().norm())}
    rep = {"what": "same workload, same seed, same box: bf16 build vs fp16-operand build, arms alternated, wall clock per step",
           "steps_per_arm": K, "ms_per_step_bf16": arms["bf16"], "ms_per_step_fp16": arms["fp16"],
           "fp16_over_bf16": f16_mean / (sum(arms["bf16"]) / len(arms["bf16"])),
           "last_position_logits_rel_l2_fp16_vs_bf16": d, "greedy_token_equal": bool(int(out[0][0]) == int(ops.argmax(lb)[0])),
           "fp16_precise_modes": precise, "bf16_precise2": bf16_p2,
           "note": "parity of each build / mode against the reference: the top-level `parity` object of this line (same run) and "
                   "profiles/r6_parity_fulldepth*.json (tests/test_gpu_parity_fulldepth.py)"}
    del m16, l16
    torch.cuda.empty_cache()
    return rep


def parity_report(dev, modes):
    """Distance of a mode's full-depth logits from the REFERENCE on BASELINE configs[2] (tests/golden/fulldepth_c3.npz: the reference's own
    modules on the hash-stream weights / inputs of tests/golden/make_golden_fulldepth.py), measured in THIS run on THIS box:
    one prefill with all 5120 logit rows per mode, compared through the golden's pins -- projections of every row on 4 + 32 fixed
    directions, the last 64 rows whole, the last position, top-1 ids of every row. modes: [(tag, operand, precise level)]. The golden is test
    data (inputs + expected outputs); nothing of oracle/ is imported here."""
    import numpy as np
    import torch

    from tests import fullwidth_util as FW
    from tests.golden import cases
    from tests.golden import make_golden_fulldepth as FD
    from vitron_amd import _lib, synth
    from vitron_amd.engine import SequenceState, llama_forward, pair_lo
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "fulldepth_c3.npz"))
    out = {"case": "c3", "golden": "tests/golden/fulldepth_c3.npz (reference modules, fp32, CPU; tests/golden/make_golden_fulldepth.py)",
           "what": "rel-L2 of this run's full-depth logits against the reference's: projections of all S rows on fixed random directions, the last 64 "
                   "rows whole, the last position; top-1 agreement over all rows", "modes": {}}
    pix, ids = FD.case_inputs("c3")
    S = int(g["S"])
    by_op = {}
    for tag, op, level in modes:
        by_op.setdefault(op, []).append((tag, level))
    for op, items in by_op.items():
        _lib.load(operand=op)
        odt = _lib.torch_dtype(op)
        model = LlavaLlamaForCausalLM(LlavaConfig(**synth.VICUNA_7B, mm_hidden_size=1024, mm_video_tower="fulldepth/LanguageBind_Video_merge",
                                                  kv_prefix_reuse=False))
        vcfg, vsd, psd, rsd = FD.case_weights("c3", dev)
        model.get_video_tower().load_state(vcfg, vsd)
        sd = dict(FD.llama_weights(dev))
        sd.update({"model.mm_projector." + k: v for k, v in psd.items()})
        sd.update({"model.region_extractor." + k: v for k, v in rsd.items()})
        model.load_state_dict(sd)
        model.to(dev, dtype=odt)
        del vsd, psd, rsd, sd
        llama = model.get_model().llama
        model._ensure_kv((S + 63) // 64 + 2)
        for tag, level in items:
            model.set_precise(level)
            try:
                (_, _, _, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids.to(dev), None, None, None, None, [pix.to(dev).to(odt)], None,
                                                                                     input_ids_host=ids)
                lo = pair_lo(embeds)
                seq = SequenceState()
                logits = llama_forward(llama, model.kv, [seq], embeds[0], [S], logit_rows=list(range(S)), embeds_lo=None if lo is None else lo[0])
                model.kv.release(seq.pages)
            finally:
                model.set_precise(0)
            logits = logits.float().cpu()
            proj4, rows4 = FW.vs_pin(logits, g, "logits")
            rep = {"operand": op, "precise_level": level, "rel_l2_proj": proj4, "rel_l2_4_whole_rows": rows4,
                   "last_rel_l2": FW.rel(logits[-1], g["last_logits"]), "top1": FW.topk_agreement(logits, g, "logits")[0]}
            if "logits_proj32" in g.files:
                rep["rel_l2_proj32"] = FW.rel(logits.double() @ cases.fw_directions(logits.shape[-1], n=32, seed=cases.FW_SEED + 1), g["logits_proj32"])
            if "logits_tail" in g.files:
                rep["rel_l2_last_64_rows"] = FW.rel(logits[-64:], g["logits_tail"])
            out["modes"][tag] = rep
            del logits
        del model, llama
        torch.cuda.empty_cache()
    return out


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment (the way the driver starts `--gpus 1`): start one
    process per GPU through torch.distributed.run on 127.0.0.1 and a free port, with the same arguments, and return its exit code.
    A box with fewer than N devices gets ONE JSON error line and exit code 2 -- deterministic, never a traceback.
    VT_BENCH_STUB=1 (tests/test_bench_launcher.py): no GPU, no model -- gloo on the host, a stub step; exercises exactly this path."""
    import socket
    import subprocess
    stub = os.environ.get("VT_BENCH_STUB") == "1"
    if not stub:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n_gpus:
            print(json.dumps({"metric": "visual-tokens+text-tokens/sec end-to-end prefill, 8-frame 336px clip, 1/2/4/8 GPU", "value": None,
                              "unit": "tokens/s", "n_gpus": n_gpus, "error": f"--gpus {n_gpus} requested but {have} GPU(s) visible on this box"}), flush=True)
            return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["VT_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def stub_main(args):
    """VT_BENCH_STUB=1: the launch / fence / max-over-ranks / one-JSON-line skeleton of main() with a stub step on the host (gloo).
    A plumbing check of the multi-process contract that runs without a GPU (tests/test_bench_launcher.py); never a measurement."""
    import torch
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
    S = args.frames * (args.image_size // 14) ** 2 + args.text_len
    x = torch.randn(64, 64)

    def step():
        time.sleep(0.002 * (1 + rank))                 # ranks differ: the reported time must be the slowest rank's
        return float((x @ x).sum())

    def fence():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "visual-tokens+text-tokens/sec end-to-end prefill, 8-frame 336px clip, 1/2/4/8 GPU", "value": world * S * args.steps / dt,
                          "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "stub", "data": "stub",
                          "config": {"workload": "VT_BENCH_STUB: launcher plumbing check, no GPU work", "self_launched": os.environ.get("VT_BENCH_SELF_LAUNCHED") == "1"}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def event_pair_overhead_ms(n=64):
    """How much a HIP-event pair around ONE short launch inflates it, measured on a decode-shaped weight-streaming GEMM (M = 4,
    22016 x 4096, the step's largest launch): n launches each inside its own pair (what vt_profile_* does) against the same n
    launches inside ONE pair; (sum of pairs - single pair) / n. The decode report subtracts launches x this from the per-class event
    times: a pair measures from the completion of the command before it, i.e. it also counts the dispatch gap in front of the kernel,
    which rocprofv3's kernel durations (profiles/r<N>_decode_rocprofv3_kernel_stats.csv) do not."""
    import torch

    from vitron_amd import _lib, ops
    dev = torch.device("cuda", torch.cuda.current_device())
    a = torch.randn((4, 4096), device=dev).to(torch.bfloat16)
    w = torch.randn((22016, 4096), device=dev).to(torch.bfloat16)
    out = torch.empty((4, 22016), device=dev, dtype=torch.bfloat16)
    for _ in range(8):
        ops.gemm(a, w, None, ops.EPI_BF16, out=out, cfg=_lib.CFG_SKINNY)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for e0, e1 in ev:
        e0.record()
        ops.gemm(a, w, None, ops.EPI_BF16, out=out, cfg=_lib.CFG_SKINNY)
        e1.record()
    torch.cuda.synchronize()
    per_pair = sum(e0.elapsed_time(e1) for e0, e1 in ev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm(a, w, None, ops.EPI_BF16, out=out, cfg=_lib.CFG_SKINNY)
    e1.record()
    torch.cuda.synchronize()
    return max(per_pair - e0.elapsed_time(e1), 0.0) / n


def cpu_baseline_line(args, dev):
    """The driver line's `cpu_baseline`: kind "reference" -- the reference's own modules on this box's host cores, at the headline's
    image size and at the other of {224, 336} -- whenever the staged modules are there, with the PORT (the oracle; bounded sample) kept
    beside it; the port at full depth (round 2-4 behaviour) when they are not or --cpu-baseline-kind port asks for it."""
    kind = args.cpu_baseline_kind
    ref = None
    if kind in ("auto", "reference"):
        try:
            sizes = sorted({args.image_size, 224 if args.image_size != 224 else 336})
            ref = cpu_baseline_reference(dev, sizes, args.frames, args.text_len, args.seed)
        except Exception as e:  # noqa: BLE001 -- a measurement aid must not take the bench line down
            ref = None
            print(f"[bench] cpu_baseline: the reference leg failed ({type(e).__name__}: {e}); falling back to the port", file=sys.stderr, flush=True)
    main_size = ref["by_image_size"].get(args.image_size) if ref else None
    if ref is None or not main_size or "tokens_per_s" not in main_size:
        out = cpu_baseline(args.image_size, args.frames, args.text_len, args.seed, mode=args.cpu_baseline)
        if ref is not None:
            out["reference_other_sizes"] = ref
        m = cpu_reference_measurement()
        if m is not None:
            out["reference_measured"] = {k: m[k] for k in ("kind", "tokens_per_s", "seconds_total", "cores", "where", "what") if k in m}
        return out
    port = cpu_baseline(args.image_size, args.frames, args.text_len, args.seed, mode="sample")
    out = {"value": main_size["tokens_per_s"], "unit": "tokens/s", "cores": ref["cores"], "kind": "reference",
           "sample": (f"the WHOLE workload at FULL depth, once, through the REFERENCE's own modules (oracle/_ref/vitron_ref.zip staged from the reference "
                      f"tree by oracle/stage_ref.py; imported via oracle/ref_shim.py; source: {ref['modules_from']}): LanguageBind video tower (24 layers) on the "
                      f"{args.frames}-frame {args.image_size}px clip, mlp2x_gelu projector, prepare_inputs_labels_for_multimodal, LlavaLlamaForCausalLM.forward over all "
                      f"{main_size['S']} positions (32 DISTINCT layers, eager attention), fp32, on {ref['cores']} torch threads of {ref['host_threads_available']} "
                      "host threads (fastest of the calibration)"),
           "seconds_total": main_size["seconds_total"], "seconds_towers_projector_splice": main_size["seconds_towers_projector_splice"],
           "seconds_decoder": main_size["seconds_decoder"], "by_image_size": ref["by_image_size"],
           "calibration_s_per_mlp_gemm": ref["calibration_s_per_mlp_gemm"], "seconds_model_build": ref["seconds_model_build"],
           "measured_seconds": ref["measured_seconds"] + port.get("measured_seconds", 0.0),
           "port": {k: port[k] for k in ("value", "unit", "cores", "kind", "sample", "est_seconds_per_step", "seconds_total") if k in port}}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--image-size", type=int, default=336)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--text-len", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=("full", "sample"), default="full",
                    help="N=1 only: time the CPU oracle on the whole workload at full depth (~2 min of CPU work) or on a bounded sample")
    ap.add_argument("--cpu-baseline-kind", choices=("auto", "reference", "port"), default="auto",
                    help="N=1 only: 'reference' = the reference's own modules (staged archive oracle/_ref/vitron_ref.zip) at full depth at 336 and 224 "
                         "(~3-4 min of CPU work) with a bounded sample of the port beside it; 'port' = the oracle only; 'auto' = reference when staged")
    ap.add_argument("--decode-steps", type=int, default=1024,
                    help="N=1 only: after the timed prefill region, BASELINE configs[4]: 4 x (image + box) prefill + this many batch-4 "
                         "greedy decode steps through generate() (0 = skip)")
    ap.add_argument("--c2-reps", type=int, default=10,
                    help="N=1 only: after the timed region, BASELINE configs[1] (one 336 px image + 512 tokens) this many times (0 = skip)")
    ap.add_argument("--reps-224", type=int, default=10,
                    help="N=1 only: after the timed region, configs[2] and configs[1] at the reference-native 224 px (S = 2560 / 768) this many times (0 = skip)")
    ap.add_argument("--no-empirical-peaks", action="store_true", help="skip the hipBLASLt-8192^3 / D2D-copy empirical peaks")
    ap.add_argument("--c4-steps", type=int, default=5,
                    help="after the timed region: the fixed 8-clip global batch of BASELINE configs[3] for this many steps (0 = skip)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--dtype", choices=("bf16", "fp16"), default="bf16",
                    help="operand format of the headline run: bf16 (BASELINE.json's dtype, libvitron_hip.so) or fp16 (the reference's own "
                         "inference dtype, libvitron_hip_f16.so)")
    ap.add_argument("--fp16-ab-steps", type=int, default=5,
                    help="N=1, --dtype bf16 only: after everything else, the SAME workload on the fp16-operand build for this many steps "
                         "(same box, same seed): step time next to the headline's and the distance of its logits from the bf16 run's (0 = skip)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child processes that fill roofline.traffic live (then the committed counter file is quoted)")
    ap.add_argument("--no-parity", action="store_true",
                    help="N=1 only: skip the `parity` object (one full-depth prefill of BASELINE configs[2] with all logit rows per mode -- the timed "
                         "mode and the at-tolerance mode -- against the reference's stored output, tests/golden/fulldepth_c3.npz)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if os.environ.get("VT_BENCH_STUB") == "1":
        return stub_main(args)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; running with {world} rank(s)", file=sys.stderr, flush=True)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # VT_BENCH_FORCE_DIST=1 (with torch.distributed.run --nproc-per-node 1) drives the N > 1 code path -- process group,
    # all-gather of the visual tokens, barrier, max-reduce -- on a single GPU: a plumbing check, not a measurement
    use_dist = world > 1 or os.environ.get("VT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev)

    from vitron_amd import _lib, ops, synth
    from vitron_amd.engine import SequenceState, llama_forward, pair_lo
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM

    _lib.load(operand=args.dtype)
    odt = _lib.torch_dtype(args.dtype)
    G = args.image_size // 14
    n_vis = args.frames * G * G
    S = n_vis + args.text_len
    vit_video = dict(synth.VIT_L14, image_size=args.image_size, add_time_attn=True, num_frames=args.frames)
    # the image tower is only needed by the secondary C2 / C5 reports (configs[1] / configs[4]) of a single-GPU run
    want_image = world == 1 and (args.c2_reps > 0 or args.decode_steps > 0)
    vit_image = dict(synth.VIT_L14, image_size=336) if want_image else None
    cfg = LlavaConfig(**synth.VICUNA_7B, mm_hidden_size=1024)
    model = LlavaLlamaForCausalLM(cfg)
    model.init_synthetic(dev, seed=args.seed, vit_image=vit_image, vit_video=vit_video, dtype=odt)

    # synthetic inputs (seed 4321 + rank): pixels N(0,1) in HBM, ids uniform in [3, 31999], BOS first, 8 x <image>.
    # The ids are resident in HBM like the pixels; their host copy (what a tokenizer returns before `.cuda()`) is handed over
    # too, so that building the integer splice plan never waits for the device (llava_arch.prepare_inputs_labels_for_multimodal)
    gen = synth.make_generator(4321 + rank, dev)
    clip = torch.randn((3, args.frames, args.image_size, args.image_size), generator=gen, device=dev).to(odt)
    text = torch.randint(3, 32000, (args.text_len - 1,), generator=gen, device=dev)
    ids = torch.cat([torch.tensor([1], device=dev), torch.full((args.frames,), -200, device=dev), text]).unsqueeze(0)
    ids_host = ids.cpu()
    assert ids.shape[1] == args.text_len + args.frames

    pending = []
    orig_encode = model.encode_videos
    if use_dist:  # clip-per-rank encode, ONE all-gather of visual tokens, then data-parallel prefill
        from vitron_amd.parallel import start_all_gather_visual_tokens

        def encode_videos_dist(videos):
            f = orig_encode(videos)                             # this rank's clip: [1, T, P, H]
            # every rank receives all clips' tokens (BASELINE config 4's exchange step); the transfer overlaps this rank's own
            # prefill, which only needs its own tokens -- the step waits for the gather before it counts as done
            pending.append(start_all_gather_visual_tokens(f))
            return f
        model.encode_videos = encode_videos_dist

    llama = model.get_model().llama
    model._ensure_kv((S + 63) // 64 + 4)

    def step():
        (_, _, _, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [clip], None,
                                                                             input_ids_host=ids_host)
        seq = SequenceState()
        lo = pair_lo(embeds)                      # (only in the verification mode of the fp16_ab report: operand pairs)
        logits = llama_forward(llama, model.kv, [seq], embeds[0], [embeds.shape[1]], embeds_lo=None if lo is None else lo[0])
        tok = ops.argmax(logits)
        model.kv.release(seq.pages)
        for g in pending:
            allv = g.wait()                                     # [world, T, P, H]: the gathered tokens of all clips
            assert allv.shape[0] == world
        pending.clear()
        return tok, embeds.shape[1], logits

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        tok, s_len, _ = step()
    assert s_len == S, (s_len, S)
    # ---- the timed region: EXACTLY args.steps steps, barrier + synchronize on both sides, nothing else inside. Each step boundary
    # also gets a HIP event on the stream the kernels run on (torch's current stream): per-step device times for the median.
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    fence()
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        tok, _, _ = step()
        marks[k + 1].record()
    fence()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * S * args.steps / dt
    # ---- per-kernel-class device times: a SEPARATE pass (an event pair around each of the ~390 instrumented launches of a step
    # perturbs the wall clock, so it must not sit inside the timed region)
    prof_steps = max(2, min(args.steps, 4))
    _lib.profile_begin()
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    model.encode_videos = orig_encode

    gather = gather_report(dev, world, dist, args.frames, args.image_size) if use_dist else None
    c4 = c4_report(model, llama, dev, world, rank, use_dist, dist, args.frames, args.image_size, args.text_len, args.c4_steps) \
        if args.c4_steps > 0 else None
    # the N = 1 denominator of c4's strong-scaling ratio, measured in THIS run on rank 0's GPU (all 8 clips through the one device, the
    # other ranks idle at the barrier below) instead of being read from a committed file of another box
    if c4 is not None and use_dist and (8 % world) == 0:      # (also under VT_BENCH_FORCE_DIST on one GPU: the plumbing test walks this leg)
        try:
            c4["uneven_prompts"] = c4_uneven_report(model, llama, dev, world, rank, dist, args.frames, args.image_size, max(2, args.c4_steps // 2))
        except Exception as e:  # noqa: BLE001  (never run on hardware here: an error is reported, the line survives)
            c4["uneven_prompts"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                dist.barrier()
            except Exception:  # noqa: BLE001
                pass
    c4_n1 = None
    if c4 is not None and world > 1 and "tokens_per_s" in c4:
        if rank == 0:
            c4_n1 = c4_report(model, llama, dev, 1, 0, False, None, args.frames, args.image_size, args.text_len, max(2, args.c4_steps // 2))
        dist.barrier()

    if rank == 0:
        gt = prof["gemm_tile"]
        achieved = gt["work"] / (gt["ms"] * 1e-3) / 1e12 if gt["ms"] > 0 else 0.0
        fl = algorithmic_flops(S, n_vis, args.frames, G * G + 1, G * G)
        traffic, traffic_build, traffic_file = pmc_traffic_per_launch()
        live_traffic, live_note = (None, "skipped (--no-live-traffic)") if (args.no_live_traffic or world > 1) else live_pmc_traffic(args.dtype)
        out = {
            "metric": "visual-tokens+text-tokens/sec end-to-end prefill, 8-frame 336px clip, 1/2/4/8 GPU",
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {
                "workload": (f"BASELINE configs[2]: {args.frames}-frame {args.image_size}x{args.image_size} clip "
                             f"({n_vis} visual tokens) + {args.text_len}-token prompt -> S={S}; LanguageBind video ViT-L/14 "
                             "(23 of 24 layers, temporal attention) + mlp2x_gelu projector + Vicuna-7B-shaped decoder prefill "
                             "(32 layers, paged KV), last-position logits + greedy token; random-init weights"),
                "clips_per_gpu": 1, "tokens_per_step_per_gpu": S, "parallelism": f"clip-parallel x{world} + all-gather of visual tokens" if world > 1 else "single GPU",
                "timing": "value / ms_per_step: wall clock over the K timed steps between barrier+synchronize fences (max over ranks); "
                          "ms_per_step_hipevent_median: median of the K per-step HIP-event intervals on the compute stream of rank 0",
                "ms_per_step_hipevent_median": step_ms[len(step_ms) // 2], "ms_per_step_hipevent_min": step_ms[0],
                "ms_per_step_hipevent_max": step_ms[-1],
                "algorithmic_tflop_per_step": fl["total"] / 1e12,
                "end_to_end_tflops_per_gpu": fl["total"] / 1e12 / (ms_per_step * 1e-3),
                "end_to_end_frac_of_mfma_peak": fl["total"] / 1e12 / (ms_per_step * 1e-3) / MFMA_BF16_PEAK_TFLOPS,
                "kernel_ms_per_step": {k: v["ms"] / prof_steps for k, v in prof.items() if v["launches"]},
                "kernel_ms_note": f"HIP-event time per kernel class from a separate pass of {prof_steps} steps after the timed region",
            },
            "roofline": {"bound": "mfma", "kernel": args.dtype + " MFMA tile GEMM class: gemm_w4_kernel<*> (256x256 / 320x256 tile, four waves of 128x128 / 160x128, ~96 % of the class time) + gemm_w4r_kernel<*> (160x128 tile on a four-deep LDS ring: N = 1024 projections) + gemm_p8_kernel<*,0,true> (4-phase ping-pong: activation epilogues on 256-row tiles, split-K) + gemm_bt_kernel<*> (small tiles) + splitk_reduce_resid_kernel, all epilogues",
                         "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_BF16_PEAK_TFLOPS,
                         # `traffic`: the dominant launch (gate/up, 5120 x 22016 x 4096) measured live when the profiler can run, else the class
                         # mean of the committed counter passes -- `traffic_source` says which
                         "traffic": live_traffic if live_traffic is not None else traffic,
                         "traffic_source": "live" if live_traffic is not None else ("committed" if traffic is not None else None),
                         "traffic_kernel": ("gate/up launch of the step: gemm_w4_kernel<SwiGLU, 256-row tile>, 5120 x 22016 x 4096; algorithmic bytes "
                                            "A 41.9 MB + W 180.4 MB + out 112.7 MB = 335 MB") if live_traffic is not None else "class mean over all tile-GEMM launches of the step",
                         "traffic_algorithmic_bytes": 335.0e6 if live_traffic is not None else None,
                         "traffic_note": live_note if live_traffic is not None else
                                         (("live collection failed (" + str(live_note) + "); mean bytes per launch, FETCH_SIZE x2 + WRITE_SIZE (KiB) from the COMMITTED rocprofv3 "
                                           f"--pmc passes {traffic_file} (measured on {traffic_build}); includes Infinity-Cache hits")
                                          if traffic is not None else "no live collection (" + str(live_note) + ") and no committed PMC traffic file"),
                         "traffic_committed_class_mean": traffic,
                         "launches_per_step": gt["launches"] / prof_steps,
                         "avg_launch_ms": gt["ms"] / max(gt["launches"], 1),
                         "algorithmic_gflop_per_launch": gt["work"] / max(gt["launches"], 1) / 1e9},
        }
        if c4 is not None:
            out["config"]["c4"] = c4
            # SURVEY.md 8(e)'s >= 6x target is defined on THIS quantity (fixed 8-clip batch), not on the weak-scaling `value`
            if c4_n1 is not None and "tokens_per_s" in c4_n1:
                c4["n1_tokens_per_s_same_run"] = c4_n1["tokens_per_s"]
                c4["n1_ms_per_step_same_run"] = c4_n1["ms_per_step"]
                c4["n1_steps"] = c4_n1["steps"]
                c4["scaling_vs_c4_n1"] = c4["tokens_per_s"] / c4_n1["tokens_per_s"]
                c4["scaling_note"] = "denominator = all 8 clips through rank 0's GPU alone, measured in this same run"
            n1 = committed_c4_n1()
            if n1 is not None and "tokens_per_s" in c4:
                c4["n1_tokens_per_s_committed"] = n1["tokens_per_s"]
                c4["n1_source"] = n1["source"]
                if "scaling_vs_c4_n1" not in c4 and world > 1:
                    c4["scaling_vs_c4_n1"] = c4["tokens_per_s"] / n1["tokens_per_s"]
                    c4["scaling_note"] = "denominator read from a committed file of another box (no same-run N = 1 measurement)"
        if gather is not None:
            out["config"]["visual_token_exchange"] = gather
        if world == 1 and args.c2_reps > 0:
            out["config"]["c2"] = c2_report(model, llama, dev, args.seed, args.c2_reps)
        if world == 1 and args.reps_224 > 0:
            # the reference-native image size: both BASELINE prefill configurations again at 224 px
            out["config"]["c3_224"] = prefill_report_224(model, llama, dev, args.seed, "clip", args.reps_224, args.frames, args.text_len)
            out["config"]["c2_224"] = prefill_report_224(model, llama, dev, args.seed, "image", args.reps_224, args.frames, args.text_len)
        if world == 1 and args.decode_steps > 0:
            out["decode"] = c5_report(model, llama, dev, args.decode_steps, args.seed)
            out["decode"]["synthetic_context_64_steps"] = decode_report_synthetic(model, llama, dev, 64)
        if world == 1 and not args.no_empirical_peaks:
            emp = empirical_peaks(dev)
            out["roofline"]["empirical_peaks"] = emp
            if emp.get("mfma_only_bf16_tflops_workload_like_operands"):
                out["roofline"]["frac_of_mfma_only_rate"] = achieved / emp["mfma_only_bf16_tflops_workload_like_operands"]
                out["roofline"]["frac_of_mfma_only_rate_note"] = ("achieved / what a loop of NOTHING BUT MFMAs sustains on this box under its power limit on "
                                                                  "workload-like operand values (roofline.empirical_peaks); `frac` stays against the 2.5 PFLOP/s dense bf16 peak")
            if emp.get("hipblaslt_bf16_gemm_8192_tflops"):
                out["roofline"]["frac_of_empirical_gemm_peak"] = achieved / emp["hipblaslt_bf16_gemm_8192_tflops"]
            if emp.get("d2d_copy_GBps") and "decode" in out:
                out["decode"]["roofline"]["frac_of_empirical_copy_rate"] = out["decode"]["roofline"]["achieved"] / emp["d2d_copy_GBps"]
            rd = max(emp.get("hbm_read_kernel_GBps") or 0.0, emp.get("hbm_read_kernel_nt_GBps") or 0.0)
            if rd > 0 and "decode" in out:      # the decode step only READS (weights + KV pages): held against a read-only kernel's rate on this box
                out["decode"]["roofline"]["frac_of_empirical_read_rate"] = out["decode"]["roofline"]["achieved"] / rd
        if world == 1 and args.dtype == "bf16" and args.fp16_ab_steps > 0:
            out["config"]["fp16_ab"] = fp16_ab_report(args, dev, model, step, ids, ids_host, clip, vit_image)
            # the mode north_star's 1e-3 is met in, as a first-class number beside the headline: the fp16-operand build (the
            # reference's own inference dtype) in precise level 3 on the SAME workload, inputs and box
            p3 = out["config"]["fp16_ab"]["fp16_precise_modes"]["precise3"]
            out["config"]["at_tolerance"] = {
                "mode": "fp16-operand build (libvitron_hip_f16.so), precise level 3: every decoder Linear of the prefill adds the MX-FP4 product of its A "
                        "operand's rounding remainder in the same launch (v_mfma_scale_f32_16x16x128_f8f6f4), towers' MLPs + projector on operand pairs, "
                        "lm_head operand a 16-bit pair (DESIGN.md 4)",
                "ms_per_step": p3["ms_per_step"], "tokens_per_s": S / (p3["ms_per_step"] * 1e-3), "steps": args.fp16_ab_steps,
                "frac_of_peak": fl["total"] / 1e12 / (p3["ms_per_step"] * 1e-3) / MFMA_BF16_PEAK_TFLOPS,
                "over_headline_step": p3["ms_per_step"] / ms_per_step,
                "logits_rel_l2": None, "tolerance": 1e-3,
                "logits_rel_l2_note": "filled from the `parity` object of this line (mode at_tolerance: rel_l2_proj32 over all rows; last_rel_l2 beside it)"}
        if world == 1 and not args.no_parity:
            modes = [("timed", args.dtype, 0)]
            if "at_tolerance" in out["config"]:
                modes.append(("at_tolerance", "fp16", 3))
            out["parity"] = parity_report(dev, modes)
            if "at_tolerance" in out["config"] and "at_tolerance" in out["parity"]["modes"]:
                pm = out["parity"]["modes"]["at_tolerance"]
                out["config"]["at_tolerance"]["logits_rel_l2"] = pm.get("rel_l2_proj32", pm["rel_l2_proj"])
                out["config"]["at_tolerance"]["last_position_logits_rel_l2"] = pm["last_rel_l2"]
                out["config"]["at_tolerance"]["within_tolerance"] = bool(out["config"]["at_tolerance"]["logits_rel_l2"] <= 1e-3 and pm["last_rel_l2"] <= 1e-3)
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.synchronize()
            out["cpu_baseline"] = cpu_baseline_line(args, dev)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
