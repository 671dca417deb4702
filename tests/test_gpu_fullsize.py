"""BASELINE-size checks through the C ABI. The CPU oracle cannot run these sizes in test time, so they use the
size-independent properties the domain offers (the oracle-vs-kernel parity proper lives in test_gpu_kernels.py /
test_gpu_model.py at small sizes):

  * GEMM at the decoder's real shapes (M = 5120 = C3's sequence, every tile path incl. the M-split and the weight-streaming
    kernel): a one-hot activation matrix must return the selected weight columns BIT-EXACTLY (every tile, every K step, the
    epilogue layout); integer-valued operands must give exact integer sums in fp32; sampled rows against fp64.
  * attention at S = 5120, 32 heads, hd 128, causal, paged: softmax rows sum to one (a constant V comes back), sampled query
    rows against an fp64 restatement, and the K / V^T pages hold exactly the rotated keys / transposed values.
  * the 7B-shaped decoder: one-shot prefill == chunked prefill == prefill + decode steps (same logits within the bf16 noise floor
    of a 32-layer chain), batch order does not matter.
  * towers: images of a batch are independent (permuting the batch permutes the features bit-exactly).
"""
import math

import pytest
import torch

from tests.util import bf16r, rel_l2
from vitron_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from vitron_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


LLM_SHAPES = [(5120, 12288, 4096, "BF16"), (5120, 4096, 4096, "F32_RESID"), (5120, 22016, 4096, "SWIGLU_BF16"),
              (5120, 4096, 11008, "F32_RESID"), (1088, 12288, 4096, "BF16"), (4616, 4096, 1024, "BF16_GELU"),
              (1088, 22016, 4096, "SWIGLU_BF16"), (4616, 3072, 1024, "BF16"), (2436, 22016, 4096, "SWIGLU_BF16"),   # 224-row tiles, ragged last tile row
              (4, 12288, 4096, "BF16"), (4, 4096, 11008, "F32_RESID"), (4, 32000, 4096, "F32")]


DT = {"bf16": torch.bfloat16, "fp16": torch.float16}
ULP = {"bf16": 2.0 ** -7, "fp16": 2.0 ** -10}        # one unit in the last place of a value in [1, 2), x 2 (an ulp either way)


@pytest.mark.parametrize("op", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K,epi_name", LLM_SHAPES)
def test_gemm_full_shapes_one_hot_and_integers(dev, M, N, K, epi_name, op):
    from vitron_amd import ops
    epi = getattr(ops, "EPI_" + epi_name)
    dt = DT[op]
    bf16r = lambda t: t.to(dt).float()  # noqa: E731 -- "round to the operand format" (the name the bf16-only version of this test used)
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    w = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(dt)
    # 1) one-hot rows: C[m][n] = W[n][sel[m]] exactly (a sum with a single non-zero term), for every (m, n) of every tile
    sel = torch.randint(0, K, (M,), generator=g, device=dev)
    a = torch.zeros((M, K), device=dev, dtype=dt)
    a[torch.arange(M, device=dev), sel] = 1.0
    picked = w[:, sel].t().float()                                   # [M, N]
    if epi == ops.EPI_F32_RESID:
        resid = torch.randn((M, N), generator=g, device=dev)
        out = ops.gemm(a, w, None, epi, out=resid.clone())
        assert torch.equal(out, resid + picked)
    elif epi == ops.EPI_SWIGLU_BF16:
        out = ops.gemm(a, w, None, epi)
        p4 = picked.view(M, N // 32, 2, 16)
        gate, up = p4[:, :, 0], p4[:, :, 1]
        ref = (gate / (1.0 + torch.exp(-gate)) * up).reshape(M, N // 2)
        assert rel_l2(out.float(), bf16r(ref)) <= 1e-3                 # silu goes through the fast exp: not bit-exact
        assert (out.float() - bf16r(ref)).abs().max() <= ULP[op] * ref.abs().max()
    elif epi == ops.EPI_BF16_GELU:
        out = ops.gemm(a, w, None, epi)
        assert rel_l2(out.float(), bf16r(torch.nn.functional.gelu(picked))) <= 1e-3
    else:
        out = ops.gemm(a, w, None, epi)
        assert torch.equal(out.float(), picked)                       # 16-bit store of a 16-bit value / fp32 store: exact
    # 2) small integers: every partial sum is an exact integer in fp32, so the result is exact whatever the order
    if epi in (ops.EPI_F32, ops.EPI_F32_RESID):
        ai = torch.randint(-2, 3, (M, K), generator=g, device=dev).to(dt)
        wi = torch.randint(-2, 3, (N, K), generator=g, device=dev).to(dt)
        base = torch.zeros((M, N), device=dev)
        out = ops.gemm(ai, wi, None, epi, out=base if epi == ops.EPI_F32_RESID else None)
        rows = torch.randint(0, M, (min(M, 16),), generator=g, device=dev)
        ref = ai[rows].double() @ wi.double().t()
        assert torch.equal(out[rows].double(), ref)
        # checksum of checksums over the whole output: column sums of A (as integers) dotted with row sums of W
        tot = (ai.double().sum(0) * wi.double().sum(0)).sum()
        assert float(out.double().sum()) == float(tot)


@pytest.mark.parametrize("op", ["bf16", "fp16"])
def test_attention_full_size_properties(dev, op):
    """S = 5120, 32 heads, hd = 128, causal, through kv_tiles (rotary) + flash attention on a shuffled page table."""
    from oracle import vitron_oracle as O
    from vitron_amd import ops
    dt = DT[op]
    bf16r = lambda t: t.to(dt).float()  # noqa: E731
    S, heads, hd = 5120, 32, 128
    D = heads * hd
    g = torch.Generator(device=dev).manual_seed(99)
    qkv = (torch.randn((S, 3 * D), generator=g, device=dev)).to(dt)
    ntile = S // 64
    kt = torch.full((ntile * heads * 64 * hd,), float("nan"), dtype=dt, device=dev)
    vt = torch.full_like(kt, float("nan"))
    table = torch.randperm(ntile, generator=torch.Generator().manual_seed(1)).to(torch.int32).to(dev)
    cos, sin = O.rope_tables(hd, S)
    cd, sd_ = cos.to(dev).contiguous(), sin.to(dev).contiguous()
    pos = torch.arange(S, dtype=torch.int32, device=dev)
    desc = torch.tensor([[0, S, S, 0]], dtype=torch.int32, device=dev)
    x = qkv.clone()
    ops.kv_tiles(x, 0, D, 2 * D, kt, vt, table, desc, ntile, heads, hd, cd, sd_, pos)
    # pages hold the rotated keys (bf16 of the fp32 rotation, one ulp of fma-vs-mul slack) and the transposed values as fp16
    k = qkv[:, D:2 * D].float().view(S, heads, hd)
    k1, k2 = k[..., :hd // 2], k[..., hd // 2:]
    c, s_ = cd[:S].unsqueeze(1), sd_[:S].unsqueeze(1)
    krot = torch.cat([k1 * c - k2 * s_, k2 * c + k1 * s_], -1).to(dt)           # [S, heads, hd]
    kp = kt.view(ntile, heads, 64, hd)[table.long()]                                   # logical tile order
    got_k = kp.permute(0, 2, 1, 3).reshape(S, heads, hd).float()
    # the kernel contracts a*c - b*s into an fma, torch rounds the two products first: equal up to one bf16 ulp, rarely
    assert float((got_k != krot.float()).float().mean()) < 0.02
    assert ((got_k - krot.float()).abs() <= ULP[op] * krot.float().abs().clamp_min(2.0 ** -6)).all()
    vp = vt.view(ntile, heads, hd, 64)[table.long()]
    v = qkv[:, 2 * D:].view(S, heads, hd)
    # V^T pages hold fp16 (vt_common.h): the bf16 values exactly wherever fp16 is normal, within 2^-25 below that
    got_v = vp.view(torch.float16).permute(0, 3, 1, 2).reshape(S, heads, hd).float()
    normal = v.float().abs() >= 2.0 ** -14
    assert torch.equal(got_v[normal], v.float()[normal])
    assert float((got_v - v.float()).abs().max()) <= 2.0 ** -25
    scale = 1.0 / math.sqrt(hd)
    out = ops.flash_attn(x, kt, vt, table, desc, S, heads, hd, True, scale)
    # sampled query rows against fp64 on the same rotated bf16 q / k
    q = x[:, :D].float().view(S, heads, hd)                                            # rotated in place by kv_tiles
    for r in (0, 1, 63, 64, 65, 2047, 4095, 5119):
        sc = torch.einsum("hd,khd->hk", q[r].double(), got_k[:r + 1].double()) * scale          # the keys the kernel really holds
        p = torch.softmax(sc, -1)
        ref = torch.einsum("hk,khd->hd", p, v[:r + 1].double()).reshape(D)
        assert rel_l2(out[r].float(), bf16r(ref.float())) <= (1.4e-3 if op == "bf16" else 6e-4), r     # measured 9.0e-4 (bf16 operands, fp16 softmax weights)
    # softmax rows sum to one: with V == const (per head-dim channel) the output is that constant
    const = torch.linspace(-2, 2, D, device=dev).to(dt)
    x2 = qkv.clone()
    x2[:, 2 * D:] = const
    ops.kv_tiles(x2, 0, D, 2 * D, kt, vt, table, desc, ntile, heads, hd, cd, sd_, pos)
    out2 = ops.flash_attn(x2, kt, vt, table, desc, S, heads, hd, True, scale)
    assert (out2.float() - const.float()).abs().max() <= ULP[op] * 2.0


@pytest.fixture(scope="module")
def model7b(dev):
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM
    m = LlavaLlamaForCausalLM(LlavaConfig(**synth.VICUNA_7B, mm_hidden_size=1024, kv_prefix_reuse=False))
    m.init_synthetic(dev, seed=1234, vit_image=dict(synth.VIT_L14, image_size=336), vit_video=None)
    return m


def test_decoder_7b_chunking_and_batch_invariance(dev, model7b):
    """Vicuna-7B-shaped decoder at C2's length (S = 1088): one-shot prefill vs two chunks vs prefill + 3 decode steps;
    a batch of two different sequences gives each the logits it gets alone."""
    from vitron_amd.engine import PagedKVCache, SequenceState, llama_forward
    llama = model7b.get_model().llama
    g = torch.Generator(device=dev).manual_seed(5)
    S = 1088
    emb = (torch.randn((S, 4096), generator=g, device=dev) * 0.02).bfloat16()
    emb_b = (torch.randn((300, 4096), generator=g, device=dev) * 0.02).bfloat16()
    kv = PagedKVCache(llama, 64)
    rows = [0, 500, 1084, 1085, 1086, 1087]

    def run(chunks):
        s = SequenceState()
        outs, o = [], 0
        for n in chunks:
            lr = [r - o for r in rows if o <= r < o + n]
            lg = llama_forward(llama, kv, [s], emb[o:o + n], [n], logit_rows=lr)
            if lr:
                outs.append(lg)
            o += n
        kv.release(s.pages)
        return torch.cat(outs, 0)
    full = run([S])
    assert torch.isfinite(full).all()
    chunked = run([700, 388])
    decoded = run([1085, 1, 1, 1])          # the last three rows come from single-token decode steps (folded-norm path)
    # 32 layers of bf16 operands: two differently tiled paths sit ~1e-2 (same kernels, re-chunked attention) to ~2e-2 (decode
    # kernels: weight-streaming GEMMs, folded RMSNorm, fused attention) apart -- the deep-chain noise floor of DESIGN.md 4
    assert rel_l2(chunked, full) <= 2e-2 and rel_l2(decoded, full) <= 3e-2, (rel_l2(chunked, full), rel_l2(decoded, full))
    assert float((chunked.argmax(-1) == full.argmax(-1)).float().mean()) >= 0.8
    # batch of two sequences == each alone
    # (the two-sequence pass is 320 attention workgroups of 256 rows: automatic selection hands it to the one-wave-per-SIMD attention
    # kernel while the single 1088-row prompt stays on 128-row blocks -- two kernels, two roundings, 32 layers deep: measured 2.5e-2 on
    # the one compared row. Batch invariance itself is checked with the kernel held fixed; the automatic choice gets the decode bound.)
    from vitron_amd import ops
    sa, sb = SequenceState(), SequenceState()
    both_auto = llama_forward(llama, kv, [sb, sa], torch.cat([emb_b, emb]), [300, S])
    kv.release(sa.pages)
    kv.release(sb.pages)
    assert rel_l2(both_auto[1:2], full[-1:]) <= 3.7e-2, rel_l2(both_auto[1:2], full[-1:])
    ops.flash_attn_select(1)
    try:
        sa, sb = SequenceState(), SequenceState()
        both = llama_forward(llama, kv, [sb, sa], torch.cat([emb_b, emb]), [300, S])
    finally:
        ops.flash_attn_select(0)
    kv.release(sa.pages)
    kv.release(sb.pages)
    s1 = SequenceState()
    alone_b = llama_forward(llama, kv, [s1], emb_b, [300])
    kv.release(s1.pages)
    assert rel_l2(both[1:2], full[-1:]) <= 2e-2 and rel_l2(both[0:1], alone_b) <= 2e-2


def test_prefill_folded_rmsnorm_matches_separate_norms(dev, model7b):
    """rows > 64: RMSNorm folded into the MFMA tile GEMMs (producers: 4-phase kernel, small-tile remainder, split-K reduce pass;
    consumers: qkv / gate_up) against the same pass with separate norm launches (vt_llama_model.prefill_norm_fold = 0). The only difference is
    where bf16 rounding happens (x*w before the row factor instead of after): two independently rounded bf16 paths: ~6e-3 apart after two layers at the 7B width (each equally far from fp32:
    tests/test_gpu_model.py::test_prefill_folded_rmsnorm_vs_oracle), the
    deep-chain noise floor after 32."""
    import os
    import time
    from vitron_amd.engine import PagedKVCache, SequenceState, llama_forward
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM
    shallow = LlavaLlamaForCausalLM(LlavaConfig(**dict(synth.VICUNA_7B, num_hidden_layers=2), mm_hidden_size=1024, kv_prefix_reuse=False))
    shallow.init_synthetic(dev, seed=77, vit_image=None, vit_video=None)
    g = torch.Generator(device=dev).manual_seed(15)
    for name, model, tol_logits, tol_hidden in (("2 layers", shallow, 1e-2, 1e-2), ("32 layers", model7b, 4e-2, 4e-2)):
        llama = model.get_model().llama
        kv = PagedKVCache(llama, 96)
        for S in (5120, 1088, 300):       # M-split + split-K remainder / one round of 256x256 tiles / small tiles + split-K
            emb = (torch.randn((S, 4096), generator=g, device=dev) * 0.02).bfloat16()
            res = {}
            for fold in ("1", "0"):
                llama.set_prefill_norm_fold(fold == "1")
                try:
                    times = []
                    for _ in range(2):
                        s_ = SequenceState()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        lg, hid = llama_forward(llama, kv, [s_], emb, [S], logit_rows=[0, S // 2, S - 1], return_hidden=True)
                        torch.cuda.synchronize()
                        times.append(time.perf_counter() - t0)
                        kv.release(s_.pages)
                    res[fold] = (lg, hid, min(times))
                finally:
                    llama.set_prefill_norm_fold(False)
            (la, ha, ta), (lb, hb, tb) = res["1"], res["0"]
            assert torch.isfinite(la).all() and torch.isfinite(ha).all()
            print(f"prefill {name} S={S}: folded {ta * 1e3:.2f} ms, separate norms {tb * 1e3:.2f} ms, "
                  f"rel_l2 logits {rel_l2(la, lb):.2e} hidden {rel_l2(ha, hb):.2e}")
            assert rel_l2(la, lb) <= tol_logits and rel_l2(ha, hb) <= tol_hidden, (name, S, rel_l2(la, lb), rel_l2(ha, hb))


def test_decoder_packed_batch_of_eight_clip_prompts(dev):
    """BASELINE configs[3]'s per-node load on one GPU: eight 5120-row prompts (8-frame clip + 512 tokens each) packed into ONE
    decoder pass of 40960 rows (activation buffers of 1 GB: element offsets up to 5e8, byte offsets beyond 2^31) through a
    7B-wide two-layer decoder. Every sequence must get the logits it gets alone (same kernels, other tile rows), and two
    sequences with identical rows identical logits."""
    from vitron_amd.engine import PagedKVCache, SequenceState, llama_forward
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM
    m = LlavaLlamaForCausalLM(LlavaConfig(**dict(synth.VICUNA_7B, num_hidden_layers=2), mm_hidden_size=1024, kv_prefix_reuse=False))
    m.init_synthetic(dev, seed=78, vit_image=None, vit_video=None)
    llama = m.get_model().llama
    g = torch.Generator(device=dev).manual_seed(25)
    S, B = 5120, 8
    emb = (torch.randn((B * S, 4096), generator=g, device=dev) * 0.02).bfloat16()
    emb[7 * S:] = emb[2 * S:3 * S]                                  # sequence 7 repeats sequence 2
    kv = PagedKVCache(llama, B * (S // 64) + 8)
    seqs = [SequenceState() for _ in range(B)]
    rows = [b * S + r for b in range(B) for r in (0, S // 2, S - 1)]
    packed = llama_forward(llama, kv, seqs, emb, [S] * B, logit_rows=rows).view(B, 3, -1)
    assert torch.isfinite(packed).all()
    assert rel_l2(packed[7], packed[2]) <= 2e-3                      # same rows at another offset: other tiles, same values
    for s_ in seqs:
        kv.release(s_.pages)
    for b in (0, 5, 7):
        s1 = SequenceState()
        alone = llama_forward(llama, kv, [s1], emb[b * S:(b + 1) * S], [S], logit_rows=[0, S // 2, S - 1])
        kv.release(s1.pages)
        assert rel_l2(packed[b], alone) <= 5e-3, (b, rel_l2(packed[b], alone))
    assert len(kv.free) == kv.num_pages


def test_image_tower_full_size_batch_independence(dev, model7b):
    """LanguageBind image tower at 336 px (ViT-L/14, 23 layers, 577 tokens): permuting the batch permutes the projected
    features bit-exactly, and a batch item equals the same image encoded alone."""
    g = torch.Generator(device=dev).manual_seed(6)
    imgs = torch.randn((3, 3, 336, 336), generator=g, device=dev).bfloat16()
    f = model7b.encode_images(imgs)[0]
    assert f.shape == (3, 576, 4096) and torch.isfinite(f.float()).all()
    perm = [2, 0, 1]
    f2 = model7b.encode_images(imgs[perm])[0]
    assert torch.equal(f2.view(torch.int16), f[perm].view(torch.int16))
    f1 = model7b.encode_images(imgs[1:2])[0]
    assert rel_l2(f1.float(), f[1:2].float()) <= 8e-3     # a different M picks different GEMM tiles / split-K: same maths, bf16 noise over 23 layers


def test_bench_distributed_path_on_one_gpu(dev):
    """bench.py's N > 1 plumbing (RCCL process group, all-gather of the visual tokens, barrier, max-reduce of the time) driven
    on ONE GPU through `torch.distributed.run --nproc-per-node 1` + VT_BENCH_FORCE_DIST=1: must print exactly one JSON line
    with the driver's keys. (The collective's cross-rank behaviour is covered on gloo, world size 2, in test_parallel_gloo.py.)"""
    import json
    import os
    import random
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VT_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = str(random.randint(20000, 40000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--decode-steps", "0", "--c2-reps", "0", "--no-empirical-peaks", "--c4-steps", "1", "--no-parity", "--no-live-traffic",
           "--reps-224", "0", "--fp16-ab-steps", "0"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak" and d["roofline"]["bound"] == "mfma"
    # the C4 object (fixed global batch of 8 clips: all eight through this one GPU, 40 960 packed decoder rows) and the exchange
    # step's two transports went through the distributed code path too
    c4 = d["config"]["c4"]
    assert c4["global_clips"] == 8 and c4["clips_per_gpu"] == 8 and c4["scaling"] == "strong" and c4["tokens_per_s"] > 0
    assert c4["tokens_per_s"] > 0.5 * d["value"]              # eight clips packed are no slower per token than one
    # the leg that CONSUMES the gathered tokens (uneven prompts, cost-balanced prefill placement; world 1: every sequence on this rank)
    un = c4["uneven_prompts"]
    assert "error" not in un, un
    assert un["placement"] == [0] * 8 and un["tokens_per_s"] > 0.4 * d["value"], un
    ex = d["config"]["visual_token_exchange"]
    assert isinstance(ex["rccl_all_gather_ms"], float) and isinstance(ex["direct_p2p_ms"], float), ex
    assert d["config"]["ms_per_step_hipevent_median"] > 0


def test_fused_qkv_epilogue_matches_separate_kv_tiles_pass(dev):
    """Prefill at the 7B width: rotary + K / V^T page writes inside the QKV GEMM's epilogue (vt_llama_model.qkv_fuse = 1) against the
    separate vt_kv_tiles pass (the default): the same fp32 accumulators rounded to bf16, rotated by the same expression
    and stored to the same slots -- the KV pool and the logits must be BIT-IDENTICAL. Cases: one long prompt (whole pages + a
    ragged last page whose tail must be zero-filled), S = 5120 (the benchmark's pass), and a packed batch of two sequences
    that both continue an earlier prefill (past lengths that are not multiples of 8 or 64: per-row slot look-ups, page-straddling
    8-token groups)."""
    from vitron_amd.engine import PackedLlama, PagedKVCache, SequenceState, llama_forward
    cfg = dict(synth.VICUNA_7B, num_hidden_layers=2)
    llama = PackedLlama(synth.llama_state(cfg, synth.make_generator(5, dev), dev), cfg, dev)
    g = torch.Generator(device=dev).manual_seed(6)

    def run(fuse, plan):
        llama.set_qkv_fuse(fuse)
        kv = PagedKVCache(llama, 200)
        kv.k.fill_(float("nan"))            # whatever is not written must not matter -- and what must be zero has to be written
        kv.vt.fill_(float("nan"))
        seqs = [SequenceState() for _ in plan[0]]
        outs = []
        gg = torch.Generator(device=dev).manual_seed(7)
        for lens in plan:
            rows = sum(lens)
            emb = (torch.randn((rows, 4096), generator=gg, device=dev) * 0.02).bfloat16()
            active = [(s, n) for s, n in zip(seqs, lens) if n > 0]
            outs.append(llama_forward(llama, kv, [s for s, _ in active], emb, [n for _, n in active]))
        llama.set_qkv_fuse(False)
        return outs, kv, seqs

    for plan in ([[1088]], [[5120]], [[100, 37], [1000, 88]], [[64, 3], [1085, 3]]):
        (a, kva, sa), (b, kvb, sb) = run(True, plan), run(False, plan)
        for s1, s2 in zip(sa, sb):
            assert s1.pages == s2.pages and s1.length == s2.length
            L, hd, heads = 2, 128, 32
            ka = kva.k.view(L, kva.num_pages, heads, 64, hd)[:, s1.pages]
            kb = kvb.k.view(L, kvb.num_pages, heads, 64, hd)[:, s2.pages]
            va = kva.vt.view(L, kva.num_pages, heads, hd, 64)[:, s1.pages]
            vb = kvb.vt.view(L, kvb.num_pages, heads, hd, 64)[:, s2.pages]
            dk = (ka.view(torch.int16) != kb.view(torch.int16)).nonzero()
            dv = (va.view(torch.int16) != vb.view(torch.int16)).nonzero()
            if dk.numel():
                i = tuple(dk[:8].t())
                print("K mismatch samples (fused, separate):", ka[i].float().tolist(), kb[i].float().tolist())
                print("K mismatch by dim:", torch.bincount(dk[:, 4], minlength=128).tolist())
                print("K mismatch by slot:", torch.bincount(dk[:, 3], minlength=64).tolist())
                print("K mismatch by page:", torch.bincount(dk[:, 1]).tolist(), "by layer", torch.bincount(dk[:, 0]).tolist())
            assert dk.numel() == 0, (plan, "K pages differ at [layer, page, head, slot, dim]", dk.shape[0], dk[:6].tolist())   # bit patterns (NaN-safe)
            assert dv.numel() == 0, (plan, "V^T pages differ at [layer, page, head, dim, slot]", dv.shape[0], dv[:6].tolist())
            assert torch.isfinite(ka.float()).all() and torch.isfinite(va.float()).all()   # tails were zero-filled, nothing left NaN
        for x, y in zip(a, b):
            assert torch.equal(x, y), plan
