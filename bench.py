"""bench.py -- end-to-end multimodal prefill throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic clip: pixels already in HBM -> LanguageBind video tower
(ViT-L/14 @336, temporal attention over 8 frames, the 23 layers hidden_states[-2] needs) -> mm_projector -> splice
with the 512-token prompt (S = 8*576 + 512 = 5120) -> 32-layer Vicuna-7B-shaped decoder prefill on a paged KV cache ->
last-position logits -> greedy first token. Everything runs through libvitron_hip.so.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1: one process per GPU, one clip per rank (weak scaling). Clips are encoded on the rank that owns them, the
visual tokens are exchanged with ONE RCCL all-gather over xGMI (BASELINE config 4) that overlaps the rank's own
prefill (started asynchronously, waited for before the step ends), and every rank prefills its own sequence. value = (tokens of all ranks) / (max over ranks of the timed region).

Prints ONE JSON line (rank 0) with the driver's contract + "roofline" (dominant kernel class = the MFMA tile GEMM,
timed live with HIP events on the kernel's stream) + "cpu_baseline" (the CPU oracle on a bounded sample).
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


def algorithmic_flops(S, n_vis_tokens, frames, N_vit, image_tokens):
    """SURVEY.md 8(d) accounting: 2 FLOP/MAC, ViT 23 layers, causal attention S(S+1)/2, lm_head last row only."""
    D, I_v, H, I, L, V = 1024, 4096, 4096, 11008, 32, 32000
    rows = frames * N_vit
    vit_lin = rows * 23 * (24 * D * D + 8 * D * D)            # spatial qkv/o + mlp (24 D^2) + temporal qkv/o (8 D^2)
    vit_att = rows * 23 * (4 * N_vit * D + 4 * frames * D)
    patch = frames * image_tokens * 2 * 588 * D
    proj = n_vis_tokens * 2 * (D * H + H * H)
    llm_lin = S * L * 2 * (4 * H * H + 3 * H * I)
    llm_att = L * 4 * H * S * (S + 1) / 2
    head = 2 * H * V
    return dict(vit=vit_lin + vit_att + patch, projector=proj, llm_linear=llm_lin, llm_attention=llm_att, lm_head=head,
                total=vit_lin + vit_att + patch + proj + llm_lin + llm_att + head)


def pmc_traffic_per_launch():
    """HBM-side bytes per launch of the MFMA tile GEMM class from the committed rocprofv3 PMC passes of THIS round's build
    (profiles/r2_pmc_traffic.json: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command, summarised by
    tools/pmc_summarize.py; the file records the commit it was measured on). Units are KiB; FETCH_SIZE is doubled as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B); the counters sit on the L2's
    fabric side, so Infinity-Cache hits are included. PMC counters cannot be collected inside this process: the value is a
    committed measurement, labelled as such on the JSON line; (None, None) when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r2_pmc_traffic.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        d = json.load(f)
    tot, n = 0.0, 0
    for k, v in d.items():
        if k.startswith(("gemm_bt_kernel", "gemm_p8_kernel", "gemm_w4_kernel", "gemm_w4r_kernel", "gemm_rp_kernel")) \
                and isinstance(v, dict) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            l = v["FETCH_SIZE"]["launches"]
            tot += l * (2.0 * v["FETCH_SIZE"]["mean_per_launch"] + v["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
            n += l
    return (tot / n if n else None), d.get("_measured_on", "unknown build")


def cpu_reference_measurement():
    """The measured CPU baseline: the reference's own modules at full depth on this workload, run ONCE in the build container
    (tests/golden/make_golden_fulldepth.py -> profiles/r2_cpu_reference.json). None when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r2_cpu_reference.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def cpu_baseline(image_size, frames, text_len, seed):
    """The CPU oracle (a port of the reference's algorithm, fp32, all host cores) on a bounded sample of the same
    workload; extrapolated to the full step. See DESIGN.md 'Measurement'."""
    import torch

    from oracle import vitron_oracle as O
    from vitron_amd import synth

    # PyTorch's CPU GEMMs stop scaling long before the 256 hardware threads of the GPU box (the sampled estimate on all of them came
    # out 8x SLOWER than the reference measured on 8 cores, profiles/r2_cpu_reference.json): use at most 64, and say how many
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    gen = synth.make_generator(seed)
    G = image_size // 14
    n_vis = frames * G * G
    S = n_vis + text_len
    vcfg = dict(synth.VIT_L14, image_size=image_size, add_time_attn=True, num_frames=frames, num_hidden_layers=1)
    vsd = {k: v.float() for k, v in synth.vit_state(vcfg, gen).items()}
    clip = torch.randn((1, 3, frames, image_size, image_size), generator=gen).to(torch.bfloat16).float()
    with torch.no_grad():
        t0 = time.perf_counter()
        O.vit_forward(vsd, vcfg, clip, 0)
        t1 = time.perf_counter()
        O.vit_forward(vsd, vcfg, clip, 1)
        t2 = time.perf_counter()
        t_vit = (t1 - t0) + 23 * max((t2 - t1) - (t1 - t0), 0.0)
        psd = {k: v.float() for k, v in synth.projector_state(1024, 4096, gen).items()}
        feats = torch.randn((n_vis, 1024), generator=gen)
        t3 = time.perf_counter()
        O.projector_forward(psd, feats)
        t_proj = time.perf_counter() - t3
        lcfg = dict(synth.VICUNA_7B, num_hidden_layers=1)
        lsd = {k: v.float() for k, v in synth.llama_state(lcfg, gen).items()}
        Ss = 1024
        emb = torch.randn((1, Ss, 4096), generator=gen) * 0.02
        t4 = time.perf_counter()
        O.llama_forward(lsd, lcfg, emb, num_layers=0)        # final norm + lm_head on every position (as the reference does)
        t5 = time.perf_counter()
        O.llama_forward(lsd, lcfg, emb, num_layers=1)
        t6 = time.perf_counter()
        t_head = (t5 - t4) * (S / Ss)
        t_layer = max((t6 - t5) - (t5 - t4), 0.0) * (S / Ss)
        t_llm = t_head + 32 * t_layer
    total = t_vit + t_proj + t_llm
    return {"value": S / total, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": (f"oracle fp32 on {cores} host threads (of {os.cpu_count()}): ViT embeddings + 1 of 23 layers on the full {frames}-frame {image_size}px clip, "
                       f"projector on all {n_vis} visual tokens, final-norm+lm_head and 1 of 32 decoder layers on the first {Ss} of {S} "
                       "positions; extrapolated linearly in depth and sequence length (attention's quadratic term is under-counted, "
                       "which flatters the CPU)"),
            "est_seconds_per_step": total, "measured_seconds": time.perf_counter() - t0}


def decode_report(model, llama, dev, steps, batch=4, ctx=609):
    """Secondary report (outside the timed region, not part of `value`): BASELINE configs[4]-shaped decode -- `batch`
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
    emb = (torch.randn((batch * ctx, H), generator=gen, device=dev) * 0.02).to(torch.bfloat16)
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
    gs = prof["gemm_skinny"]
    return {"workload": f"batch {batch} greedy decode, context {ctx}+, Vicuna-7B-shaped decoder, paged KV (64-token pages), synthetic context rows",
            "steps": steps, "ms_per_step": dt * 1e3, "tokens_per_s": batch / dt,
            "roofline": {"bound": "hbm", "kernel": "gemm_skinny_dma_kernel (weight-streaming GEMM, M <= 16)",
                         "achieved": gs["work"] / (gs["ms"] * 1e-3) / 1e9 if gs["ms"] > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (gs["work"] / (gs["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if gs["ms"] > 0 else 0.0,
                         "avg_launch_ms": gs["ms"] / max(gs["launches"], 1), "launches_per_step": gs["launches"] / 4,
                         "algorithmic_mbytes_per_launch": gs["work"] / max(gs["launches"], 1) / 1e6},
            "whole_step_gbytes": (wbytes + kv_bytes) / 1e9, "whole_step_GBps": (wbytes + kv_bytes) / dt / 1e9,
            "whole_step_frac_of_hbm_peak": (wbytes + kv_bytes) / dt / 1e9 / HBM_PEAK_GBS,
            "kernel_ms_per_step": {k: v["ms"] / 4 for k, v in prof.items() if v["launches"]}}


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
    clips = [torch.randn((3, frames, image_size, image_size), generator=gen, device=dev).to(torch.bfloat16) for _ in range(n_local)]
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


def gather_report(dev, world, dist, frames, image_size, iters=10):
    """The exchange step alone (blocking, nothing to hide behind): RCCL's all-gather collective vs the direct full-mesh
    point-to-point exchange (vitron_amd.parallel.all_gather_direct_p2p), on the per-rank message of BASELINE configs[3]
    (one clip's projected visual tokens, 37.75 MB at 336 px)."""
    import torch

    from vitron_amd.parallel import all_gather_direct_p2p, start_all_gather_visual_tokens

    G = image_size // 14
    local = torch.randn((1, frames, G * G, 4096), device=dev).to(torch.bfloat16)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--image-size", type=int, default=336)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--text-len", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-steps", type=int, default=64,
                    help="N=1 only: after the timed prefill region, also time this many batch-4 greedy decode steps (0 = skip)")
    ap.add_argument("--c4-steps", type=int, default=2,
                    help="after the timed region: the fixed 8-clip global batch of BASELINE configs[3] for this many steps (0 = skip)")
    ap.add_argument("--seed", type=int, default=1234)
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
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
    from vitron_amd.engine import SequenceState, llama_forward
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM

    _lib.load()
    G = args.image_size // 14
    n_vis = args.frames * G * G
    S = n_vis + args.text_len
    vit_video = dict(synth.VIT_L14, image_size=args.image_size, add_time_attn=True, num_frames=args.frames)
    cfg = LlavaConfig(**synth.VICUNA_7B, mm_hidden_size=1024)
    model = LlavaLlamaForCausalLM(cfg)
    model.init_synthetic(dev, seed=args.seed, vit_image=None, vit_video=vit_video)

    # synthetic inputs (seed 4321 + rank): pixels N(0,1) in HBM, ids uniform in [3, 31999], BOS first, 8 x <image>.
    # The ids are resident in HBM like the pixels; their host copy (what a tokenizer returns before `.cuda()`) is handed over
    # too, so that building the integer splice plan never waits for the device (llava_arch.prepare_inputs_labels_for_multimodal)
    gen = synth.make_generator(4321 + rank, dev)
    clip = torch.randn((3, args.frames, args.image_size, args.image_size), generator=gen, device=dev).to(torch.bfloat16)
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
        logits = llama_forward(llama, model.kv, [seq], embeds[0], [embeds.shape[1]])
        tok = ops.argmax(logits)
        model.kv.release(seq.pages)
        for g in pending:
            allv = g.wait()                                     # [world, T, P, H]: the gathered tokens of all clips
            assert allv.shape[0] == world
        pending.clear()
        return tok, embeds.shape[1]

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        tok, s_len = step()
    assert s_len == S, (s_len, S)
    # ---- the timed region: EXACTLY args.steps steps, barrier + synchronize on both sides, nothing else inside. Each step boundary
    # also gets a HIP event on the stream the kernels run on (torch's current stream): per-step device times for the median.
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    fence()
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        tok, _ = step()
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

    if rank == 0:
        gt = prof["gemm_tile"]
        achieved = gt["work"] / (gt["ms"] * 1e-3) / 1e12 if gt["ms"] > 0 else 0.0
        fl = algorithmic_flops(S, n_vis, args.frames, G * G + 1, G * G)
        traffic, traffic_build = pmc_traffic_per_launch()
        out = {
            "metric": "visual-tokens+text-tokens/sec end-to-end prefill, 8-frame 336px clip, 1/2/4/8 GPU",
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
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
            "roofline": {"bound": "mfma", "kernel": "bf16 MFMA tile GEMM class: gemm_w4_kernel<*> (256x256 / 320x256 tile, four waves of 128x128 / 160x128, ~96 % of the class time) + gemm_w4r_kernel<*> (160x128 tile on a four-deep LDS ring: N = 1024 projections) + gemm_p8_kernel<*,0,true> (4-phase ping-pong: activation epilogues on 256-row tiles, split-K) + gemm_bt_kernel<*> (small tiles) + splitk_reduce_resid_kernel, all epilogues",
                         "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_note": ("mean bytes per launch, FETCH_SIZE x2 + WRITE_SIZE (KiB) from the committed rocprofv3 --pmc passes "
                                          f"profiles/r2_pmc_traffic.json (measured on {traffic_build}); includes Infinity-Cache hits; not collected live")
                                         if traffic is not None else "no PMC traffic file for this round's build (profiles/r2_pmc_traffic.json)",
                         "launches_per_step": gt["launches"] / prof_steps,
                         "avg_launch_ms": gt["ms"] / max(gt["launches"], 1),
                         "algorithmic_gflop_per_launch": gt["work"] / max(gt["launches"], 1) / 1e9},
        }
        if c4 is not None:
            out["config"]["c4"] = c4
        if gather is not None:
            out["config"]["visual_token_exchange"] = gather
        if world == 1 and args.decode_steps > 0:
            out["decode"] = decode_report(model, llama, dev, args.decode_steps)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.image_size, args.frames, args.text_len, args.seed)
            ref = cpu_reference_measurement()
            if ref is not None:
                out["cpu_baseline"]["reference_measured"] = {k: ref[k] for k in ("kind", "tokens_per_s", "seconds_total", "cores", "where", "what") if k in ref}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
