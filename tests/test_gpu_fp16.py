"""The fp16-operand build (libvitron_hip_f16.so; round 4) through the C ABI: the reference's own inference dtype
(/root/reference/vitron/model/builder.py:47 torch_dtype=float16; towers :153,161).

Same sources as the bf16 library, compiled with -DVT_OPERAND_F16=1 (vitron_amd/csrc/vt_common.h): every 16-bit tensor -- weights, norm
outputs, fused QKV, rotated q / K pages, attention output, activations, embeddings -- holds IEEE fp16 (11 mantissa bits instead of 8),
the MFMAs are v_mfma_f32_16x16x32_f16 / 32x32x16_f16, operand stores saturate at +-65504. What this file pins:

  * every GEMM kernel family and epilogue, the norms, attention (prefill / decode / temporal), splice / feed / im2col / preprocessing
    against fp64 math on the SAME fp16-rounded inputs: rel-L2 <= TOL16 = 3e-4 (one fp16 store of the output is 2.1e-4 by itself when the
    reference is not rounded; with the reference rounded too what is left is rounding flips);
  * saturation: values beyond fp16's range come out as +-65504 -- finite -- from every store path (tile epilogues direct and LDS-staged,
    SwiGLU, weight-streaming kernels, norms, the weight packer);
  * fp16 checkpoints are packed bit for bit (no detour through bf16);
  * both libraries live in one process (same symbol names, -Bsymbolic): interleaved calls do not disturb each other;
  * the drop-in surface in fp16: load_pretrained_model(torch_dtype=float16) / .half() -> generate(), decode == prefill, vs the oracle's
    fp16-storage emulation and plain fp32.
The BASELINE-width and full-depth numbers of the fp16 build are in tests/test_gpu_parity_{fullwidth,fulldepth,decode,ops}.py (op = "fp16").
"""
import math

import pytest
import torch

from oracle import vitron_oracle as O
from tests.golden import cases
from tests.util import f32, rel_l2
from vitron_amd import synth

pytestmark = pytest.mark.gpu
F16 = torch.float16
TOL16 = 3e-4
F16_MAX = 65504.0


def r16(x):
    return x.to(F16).to(torch.float32)


def rand16(shape, seed, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return r16(torch.randn(shape, generator=g) * std)


@pytest.fixture(scope="module")
def dev():
    from vitron_amd import _lib
    lib = _lib.load(operand="fp16")
    assert lib.vt_operand_format() == _lib.OPERAND_FP16 and _lib.load(operand="bf16").vt_operand_format() == _lib.OPERAND_BF16
    return torch.device("cuda:0")


def _gemm_ref(a, w, bias, epi, resid=None):
    from vitron_amd import ops
    y = a.double() @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    if epi == ops.EPI_BF16_GELU:
        y = torch.nn.functional.gelu(y)
    elif epi == ops.EPI_BF16_QGELU:
        y = y * torch.sigmoid(1.702 * y)
    elif epi == ops.EPI_BF16_RELU:
        y = torch.relu(y)
    elif epi == ops.EPI_SWIGLU_BF16:
        M, N = y.shape
        y4 = y.view(M, N // 32, 2, 16)
        y = (torch.nn.functional.silu(y4[:, :, 0]) * y4[:, :, 1]).reshape(M, N // 2)
    elif epi == ops.EPI_F32_RESID:
        y = y + resid.double()
    y = y.float()
    return y if epi in (ops.EPI_F32, ops.EPI_F32_RESID) else r16(y.clamp(-F16_MAX, F16_MAX))


# cfg ids of include/vitron_hip.h: small tiles, ping-pong 8- / 4-phase, register-pipelined, the four-wave kernels on 256 / 320 / 224-row tiles
TILE_CFGS = [2, 3, 4, 5, 6, 8, 10, 13, 14, 16]


@pytest.mark.parametrize("cfg", TILE_CFGS)
@pytest.mark.parametrize("M,N,K", [(300, 384, 256), (577, 1024, 640 + 128), (1000, 512, 1024)])
def test_fp16_gemm_tile_kernels(dev, cfg, M, N, K):
    from vitron_amd import ops
    a, w, b = rand16((M, K), 1), rand16((N, K), 2, 0.05), rand16((N,), 3).float()
    resid = rand16((M, N), 4).float()
    for epi in (ops.EPI_BF16, ops.EPI_F32, ops.EPI_BF16_GELU, ops.EPI_BF16_QGELU, ops.EPI_BF16_RELU, ops.EPI_SWIGLU_BF16, ops.EPI_F32_RESID):
        bias = None if epi == ops.EPI_SWIGLU_BF16 else b
        out = resid.to(dev).clone() if epi == ops.EPI_F32_RESID else None
        try:
            got = ops.gemm(a.to(dev).to(F16), w.to(dev).to(F16), None if bias is None else bias.to(dev), epi, out=out, cfg=cfg)
        except Exception as e:                       # a forced tile configuration may not carry every epilogue: that is an argument error,
            assert "status -1" in str(e), e          # never a wrong result
            continue
        assert got.dtype == (torch.float32 if epi in (ops.EPI_F32, ops.EPI_F32_RESID) else F16)
        assert rel_l2(got.float(), _gemm_ref(a, w, bias, epi, resid)) <= TOL16, (cfg, epi)


@pytest.mark.parametrize("M,N,K", [(577, 1024, 1024), (4616, 1024, 1024), (161, 128, 768)])
def test_fp16_gemm_ring_kernel_and_split_k(dev, M, N, K):
    from vitron_amd import _lib, ops
    a, w, b = rand16((M, K), 5), rand16((N, K), 6, 0.05), rand16((N,), 7).float()
    resid = rand16((M, N), 8).float()
    got = ops.gemm(a.to(dev).to(F16), w.to(dev).to(F16), b.to(dev), ops.EPI_F32_RESID, out=resid.to(dev).clone(), cfg=_lib.CFG_160x128_W4)
    assert rel_l2(got, _gemm_ref(a, w, b, ops.EPI_F32_RESID, resid)) <= TOL16
    ks = 4 if K >= 1024 else 2                  # a split keeps at least 256 of K
    got = ops.gemm_resid_splitk(a.to(dev).to(F16), w.to(dev).to(F16), resid.to(dev).clone(), b.to(dev), ks,
                                torch.empty(ks * M * N, device=dev))
    assert rel_l2(got, _gemm_ref(a, w, b, ops.EPI_F32_RESID, resid)) <= TOL16


@pytest.mark.parametrize("epi_name", ["BF16", "BF16_GELU", "BF16_QGELU", "BF16_RELU", "F32_RESID", "F32", "SWIGLU_BF16"])
@pytest.mark.parametrize("M", [1, 4, 16, 17, 32, 40, 64, 200, 1088])
def test_fp16_gemm_auto_dispatch_all_row_counts(dev, epi_name, M):
    """AUTO over the weight-streaming kernels (M <= 16, 17..64) and the planner's tile choice, 7B-ish N / K."""
    from vitron_amd import ops
    epi = getattr(ops, "EPI_" + epi_name)
    N, K = (2048, 1024) if M > 64 else (4096 + 32, 1408)
    a, w, b = rand16((M, K), 11), rand16((N, K), 12, 0.05), rand16((N,), 13).float()
    resid = rand16((M, N), 14).float()
    bias = None if epi == ops.EPI_SWIGLU_BF16 else b
    out = resid.to(dev).clone() if epi == ops.EPI_F32_RESID else None
    got = ops.gemm(a.to(dev).to(F16), w.to(dev).to(F16), None if bias is None else bias.to(dev), epi, out=out)
    assert rel_l2(got.float(), _gemm_ref(a, w, bias, epi, resid)) <= TOL16, (epi_name, M)


def test_fp16_stores_saturate_at_65504(dev):
    """North of fp16's range every operand store clamps to +-65504 (the reference's fp16 path would write inf and read NaN one operator
    later): tile epilogues (LDS-staged and direct store forms, SwiGLU, GELU), the weight-streaming kernels, the norms, the packer."""
    from vitron_amd import ops
    from vitron_amd.engine import _op
    K = 256
    for M, N in ((300, 512), (300, 516), (4, 512), (24, 512)):          # N % 8 != 0 -> the direct-store epilogue
        a = torch.full((M, K), 16.0)
        w = torch.zeros((N, K))
        w[0::2] = 32.0                                              # row sums +-131072: beyond fp16
        w[1::2] = -32.0
        w[5] = 0.125                                                # and one in-range column: 16 * 0.125 * 256 = 512
        for epi in (ops.EPI_BF16, ops.EPI_BF16_RELU, ops.EPI_BF16_GELU):
            got = ops.gemm(a.to(dev).to(F16), w.to(dev).to(F16), None, epi).float().cpu()
            assert torch.isfinite(got).all(), (M, N, epi)
            assert float(got[:, 0].min()) == F16_MAX and float(got[:, 5].max()) == 512.0
            assert float(got[:, 1].max()) == (-F16_MAX if epi == ops.EPI_BF16 else 0.0) or epi == ops.EPI_BF16_GELU
    # SwiGLU: silu(gate) * up with gate = up = 131072 -> 1.7e10 -> 65504
    a = torch.full((300, K), 16.0)
    w = torch.full((64, K), 32.0)
    got = ops.gemm(a.to(dev).to(F16), w.to(dev).to(F16), None, ops.EPI_SWIGLU_BF16).float().cpu()
    assert torch.isfinite(got).all() and float(got.min()) == F16_MAX
    # norms: a gain that pushes the normalised row out of range
    x = torch.randn((5, 256)).to(dev)
    y = ops.rmsnorm(x, torch.full((256,), 1.0e6, device=dev), 1e-5, dtype=F16).float().cpu()
    assert torch.isfinite(y).all() and float(y.abs().max()) == F16_MAX
    y = ops.layernorm(x, torch.full((256,), 1.0e6, device=dev), torch.zeros(256, device=dev), 1e-5, dtype=F16).float().cpu()
    assert torch.isfinite(y).all() and float(y.abs().max()) == F16_MAX
    # the packer: a bf16 / fp32 checkpoint value beyond the range saturates instead of becoming inf
    t = _op(torch.tensor([1.0e5, -3.0e38, 1.5, 65504.0]).bfloat16(), dev, F16).float().cpu()
    assert t.tolist() == [F16_MAX, -F16_MAX, 1.5, 65504.0 if float(torch.tensor(65504.0).bfloat16()) <= F16_MAX else F16_MAX]


def test_fp16_checkpoint_is_packed_bit_for_bit(dev):
    """A real Vitron / Vicuna checkpoint is fp16 (reference builder.py:47): its 11-bit mantissas must reach the kernels untouched.
    (Round 3 cast every weight to bf16 first: three mantissa bits lost before the first kernel ran.)"""
    from vitron_amd.engine import PackedLlama, interleave_gate_up
    cfg = dict(cases.LLM)
    g = torch.Generator().manual_seed(77)
    sd = {k: (torch.randn(v.shape, generator=g) * 0.05).to(F16) if v.dim() == 2 else torch.ones(v.shape, dtype=F16)
          for k, v in synth.llama_state(cfg, synth.make_generator(1), w_std=0.05).items()}
    assert not torch.equal(sd["lm_head.weight"].float(), sd["lm_head.weight"].bfloat16().float())      # genuinely beyond bf16's mantissa
    llama = PackedLlama(sd, cfg, dev, dtype=F16)
    assert llama.dtype == F16 and llama.lm_head.dtype == F16
    V = sd["lm_head.weight"].shape[0]
    assert torch.equal(llama.lm_head[:V].cpu(), sd["lm_head.weight"])
    assert torch.equal(llama.embed.cpu(), sd["model.embed_tokens.weight"])
    p = "model.layers.0."
    keep = {t.data_ptr(): t for t in llama._keep}
    wqkv = keep[llama.layers[0].wqkv].cpu()
    assert torch.equal(wqkv, torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"]], 0))
    wgu = keep[llama.layers[0].wgu].cpu()
    assert torch.equal(wgu, interleave_gate_up(sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]))


def test_both_operand_builds_in_one_process(dev):
    """libvitron_hip.so and libvitron_hip_f16.so export the same symbols; they are linked -Bsymbolic and loaded RTLD_LOCAL, so each binds
    to its own kernels. Interleaved calls must return what each library returns alone, bit for bit."""
    from vitron_amd import ops
    a, w = rand16((300, 512), 21), rand16((384, 512), 22, 0.05)        # values exact in fp16; bf16 rounds them
    a16, w16 = a.to(dev).to(F16), w.to(dev).to(F16)
    ab, wb = a.to(dev).bfloat16(), w.to(dev).bfloat16()
    first16, firstb = ops.gemm(a16, w16).clone(), ops.gemm(ab, wb).clone()
    for _ in range(3):
        assert torch.equal(ops.gemm(a16, w16), first16) and torch.equal(ops.gemm(ab, wb), firstb)
    assert first16.dtype == F16 and firstb.dtype == torch.bfloat16
    assert rel_l2(first16.float(), r16(a @ w.t())) <= TOL16
    assert rel_l2(firstb.float(), (ab.float().cpu() @ wb.float().cpu().t())) <= 3e-3
    with pytest.raises(Exception, match="dtype"):
        ops.gemm(a16, wb)                                               # mixing the formats in one call is an error, not a reinterpretation


def test_fp16_norms_splice_feed_im2col_temporal(dev):
    from vitron_amd import ops
    x = torch.randn((300, 1024), generator=torch.Generator().manual_seed(31))
    g, b = torch.rand(1024, generator=torch.Generator().manual_seed(32)) + 0.5, torch.randn(1024, generator=torch.Generator().manual_seed(33))
    y = ops.layernorm(x.to(dev), g.to(dev), b.to(dev), 1e-5, dtype=F16)
    assert y.dtype == F16 and rel_l2(y.float(), r16(torch.nn.functional.layer_norm(x.double(), (1024,), g.double(), b.double(), 1e-5).float())) <= TOL16
    for rows in (300, 4000):                                           # row-block kernel (< 2048 rows) and the persistent-wave kernel
        xr = torch.randn((rows, 1024), generator=torch.Generator().manual_seed(34))
        y = ops.rmsnorm(xr.to(dev), g.to(dev), 1e-5, dtype=F16)
        assert rel_l2(y.float(), r16((xr.double() * torch.rsqrt(xr.double().pow(2).mean(-1, keepdim=True) + 1e-5) * g.double()).float())) <= TOL16
    # splice + feed move rows untouched
    table, vis = rand16((50, 64), 35).to(dev).to(F16), rand16((7, 64), 36).to(dev).to(F16)
    plan = torch.tensor([[0, 3], [1, 6], [0, 49], [3, 0], [1, 0]], dtype=torch.int32, device=dev)
    out = ops.embed_splice(table, vis, None, plan)
    assert out.dtype == F16 and torch.equal(out[0], table[3]) and torch.equal(out[1], vis[6]) and float(out[3].abs().max()) == 0.0
    # im2col keeps fp16 pixels exact
    pix = rand16((2, 3, 28, 28), 37)
    pat = ops.im2col(pix.to(dev).to(F16), 14, 640)
    ref = torch.nn.functional.unfold(pix, 14, stride=14).transpose(1, 2).reshape(-1, 588)
    assert pat.dtype == F16 and torch.equal(pat[:, :588].float().cpu(), ref) and float(pat[:, 588:].abs().max()) == 0.0
    # temporal attention T = 8 and a generic T
    for B, T, N, heads in ((1, 8, 33, 2), (2, 4, 17, 2)):
        D = heads * 64
        qkv = rand16((B * T * N, 3 * D), 38)
        got = ops.attn_temporal(qkv.to(dev).to(F16), B, T, N, heads).float().cpu()
        q, k, v = (qkv[:, i * D:(i + 1) * D].view(B, T, N, heads, 64).permute(0, 2, 3, 1, 4).double() for i in range(3))
        ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).permute(0, 3, 1, 2, 4).reshape(B * T * N, D).float()
        assert rel_l2(got, r16(ref)) <= TOL16, (B, T)


def _attn_ref(q, k, v, scale, causal, past):
    s = (q.double() @ k.double().transpose(-1, -2)) * scale
    if causal:
        Sq, Sk = q.shape[1], k.shape[1]
        i = torch.arange(Sq)[:, None] + past
        j = torch.arange(Sk)[None, :]
        s = s.masked_fill(j > i, float("-inf"))
    return (torch.softmax(s, -1) @ v.double()).float()


@pytest.mark.parametrize("hd,heads,lens,causal", [(64, 3, [577, 64, 1, 130], False), (128, 2, [300, 129, 64], True), (128, 1, [1000], True)])
def test_fp16_flash_attention_and_decode(dev, hd, heads, lens, causal):
    """kv_tiles (fp16 q / k rotated + stored, V^T pages), the prefill flash kernel with BOTH products on the f16 MFMA, then one decode step per
    sequence through the fused decode kernel: against fp64 on the fp16 inputs. (bf16 build: 8.4e-4 measured for the same cases.)"""
    from vitron_amd import ops
    D = heads * hd
    rows = sum(lens)
    qkv = rand16((rows, 3 * D), 41)
    qd = qkv.to(dev).to(F16)
    table, desc, r0 = [], [], 0
    for L in lens:
        desc.append([r0, L, L, len(table)])
        table += list(range(len(table), len(table) + (L + 1 + 63) // 64))     # room for one more token
        r0 += L
    npages = len(table)
    kt = torch.zeros(npages * heads * 64 * hd, dtype=F16, device=dev)
    vt = torch.zeros(npages * heads * 64 * hd, dtype=F16, device=dev)
    table_t, desc_t = torch.tensor(table, dtype=torch.int32, device=dev), torch.tensor(desc, dtype=torch.int32, device=dev)
    ops.kv_tiles(qd, 0, D, 2 * D, kt, vt, table_t, desc_t, max((L + 63) // 64 for L in lens), heads, hd)
    scale = 1.0 / math.sqrt(hd)
    out = ops.flash_attn(qd, kt, vt, table_t, desc_t, max(lens), heads, hd, causal, scale)
    assert out.dtype == F16 and torch.isfinite(out.float()).all()
    worst = 0.0
    for (r0, L, _, _) in desc:
        x = qkv[r0:r0 + L]
        q, k, v = (x[:, i * D:(i + 1) * D].view(L, heads, hd).transpose(0, 1) for i in range(3))
        ref = _attn_ref(q, k, v, scale, causal, 0).transpose(0, 1).reshape(L, D)
        worst = max(worst, rel_l2(out[r0:r0 + L].float(), r16(ref)))
    print(f"[fp16-flash] hd={hd} lens={lens}: worst rel_l2 {worst:.3e}", flush=True)
    assert worst <= 6e-4, worst            # P in fp16 against the running maximum is the one rounding the reference does not share
    # one decode step per sequence (q_len 1, appended to the pages)
    new = rand16((len(lens), 3 * D), 42)
    dd = torch.tensor([[i, 1, L + 1, d[3]] for i, (d, L) in enumerate(zip(desc, lens))], dtype=torch.int32, device=dev)
    got = ops.attn_decode_fused(new.to(dev).to(F16), 0, D, 2 * D, kt, vt, table_t, dd, heads, hd, scale).float().cpu()
    for i, (r0, L, _, _) in enumerate(desc):
        x = torch.cat([qkv[r0:r0 + L], new[i:i + 1]], 0)
        q = new[i:i + 1, :D].view(1, heads, hd).transpose(0, 1)
        k, v = (x[:, j * D:(j + 1) * D].view(L + 1, heads, hd).transpose(0, 1) for j in (1, 2))
        ref = _attn_ref(q, k, v, scale, False, 0).transpose(0, 1).reshape(1, D)
        assert rel_l2(got[i:i + 1], r16(ref)) <= TOL16, i


def test_fp16_flash_attention_range_edges(dev):
    """q / k at the edge of fp16's range (scores ~ +-1e5 before the scale: fp32 accumulators, nothing overflows), V saturating, a row
    whose maximum jumps at every tile."""
    from vitron_amd import ops
    hd, heads, L = 128, 2, 300
    D = heads * hd
    scale = 1.0 / math.sqrt(hd)
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn((L, 3 * D), generator=g)
    qkv[:, :2 * D] *= 3.0                       # wide score range (spans beyond +-30)
    qkv[:, 2 * D:] *= 3.0e4                     # V near the top of fp16
    qkv[7, 2 * D + 5], qkv[100, 2 * D + 130] = 1.0e5, -2.0e5
    qd = qkv.to(dev).float().clamp(-F16_MAX, F16_MAX).to(F16)
    npages = (L + 63) // 64
    kt, vt = torch.zeros(npages * heads * 64 * hd, dtype=F16, device=dev), torch.zeros(npages * heads * 64 * hd, dtype=F16, device=dev)
    table, desc = torch.arange(npages, dtype=torch.int32, device=dev), torch.tensor([[0, L, L, 0]], dtype=torch.int32, device=dev)
    ops.kv_tiles(qd, 0, D, 2 * D, kt, vt, table, desc, npages, heads, hd)
    pages = vt.view(-1, heads, hd, 64).float()
    assert torch.isfinite(pages).all() and float(pages.abs().max()) == F16_MAX
    out = ops.flash_attn(qd, kt, vt, table, desc, L, heads, hd, True, scale).float().cpu()
    x = qd.float().cpu()
    q, k, v = (x[:, i * D:(i + 1) * D].view(L, heads, hd).transpose(0, 1) for i in range(3))
    ref = _attn_ref(q, k, v, scale, True, 0).transpose(0, 1).reshape(L, D)
    assert torch.isfinite(out).all() and rel_l2(out, r16(ref.clamp(-F16_MAX, F16_MAX))) <= 8e-4


def _tiny_states():
    return {
        "image_tower": synth.vit_state(cases.VIT_IMAGE, synth.make_generator(cases.SEED_VIT), **cases.VIT_INIT),
        "video_tower": synth.vit_state(cases.VIT_VIDEO, synth.make_generator(cases.SEED_VIT), **cases.VIT_INIT),
        "projector": synth.projector_state(cases.MM_HIDDEN, cases.LLM["hidden_size"], synth.make_generator(cases.SEED_PROJ), **cases.MLP_INIT),
        "region": synth.region_state(cases.MM_HIDDEN, cases.LLM["hidden_size"], synth.make_generator(cases.SEED_REGION), **cases.MLP_INIT),
        "llama": synth.llama_state(cases.LLM, synth.make_generator(cases.SEED_LLM), **cases.LLM_INIT),
    }


def _tiny_model(dev, dtype):
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM
    st = _tiny_states()
    cfg = LlavaConfig(**cases.LLM, mm_hidden_size=cases.MM_HIDDEN, mm_image_tower="golden/LanguageBind_Image",
                      mm_video_tower="golden/LanguageBind_Video_merge", kv_prefix_reuse=False)
    m = LlavaLlamaForCausalLM(cfg)
    m.get_image_tower().load_state(cases.VIT_IMAGE, st["image_tower"])
    m.get_video_tower().load_state(cases.VIT_VIDEO, st["video_tower"])
    sd = dict(st["llama"])
    sd.update({"model.mm_projector." + k: v for k, v in st["projector"].items()})
    sd.update({"model.region_extractor." + k: v for k, v in st["region"].items()})
    m.load_state_dict(sd)
    return m.to(dev, dtype=dtype), st


def test_fp16_model_surface_prefill_decode_and_generate(dev):
    """The reference's own call pattern -- model in fp16, images in fp16 (inference_image.py:25-29 `.to(model.device, dtype=torch.float16)`)
    -- through forward / generate on the glue cases: logits vs the reference goldens and the oracle's fp16 emulation, decode == prefill,
    every component reports and computes in fp16, and the bf16 model built from the same state dicts gives the same greedy ids where the
    reference's margin allows."""
    import numpy as np
    import os
    m16, st = _tiny_model(dev, F16)
    assert m16.dtype == F16 and m16.get_image_tower().dtype == F16 and m16.get_video_tower().dtype == F16
    assert m16.get_model().llama.embed.dtype == F16 and m16.get_model().mm_projector.packed.dtype == F16
    with pytest.raises(RuntimeError, match="already packed"):
        m16.to(dtype=torch.bfloat16)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "glue_llm.npz"))
    w = {k: f32(v) for k, v in st.items()}
    cfgs = {"image": cases.VIT_IMAGE, "video": cases.VIT_VIDEO, "llama": cases.LLM}
    worst = {}
    for name, case in cases.glue_cases().items():
        if case["input_ids"].shape[0] != 1 or case.get("max_length"):
            continue
        ids = case["input_ids"].to(dev)
        images = [im.to(dev).to(F16) for im in case["images"]]
        out = m16(input_ids=ids, images=images, regions=case["regions"], use_cache=False)
        logits = out.logits[0].float().cpu()
        with torch.no_grad():
            e, mask, pos = O.multimodal_prepare(w, cfgs, case["input_ids"], None, case["images"], case["regions"], emulate_bf16="fp16")
            lem, _ = O.llama_forward(w["llama"], cases.LLM, e, emulate_bf16="fp16")
            e32, _, _ = O.multimodal_prepare(w, cfgs, case["input_ids"], None, case["images"], case["regions"])
            l32, _ = O.llama_forward(w["llama"], cases.LLM, e32)
        d_emu, d_f32, emu_f32 = rel_l2(logits, lem[0]), rel_l2(logits, l32[0]), rel_l2(lem[0], l32[0])
        worst[name] = (d_emu, d_f32, emu_f32)
        key = f"{name}_logits"
        if key in g:
            worst[name] += (rel_l2(logits, torch.as_tensor(g[key]).reshape(logits.shape)),)
        # decode == prefill: feed all but the last 3 rows, then 3 single-token steps
        assert d_f32 <= 1.25 * emu_f32 + 5e-4, (name, d_f32, emu_f32)
    print("[fp16-tiny] " + ", ".join(f"{k}: " + "/".join(f"{v:.2e}" for v in vs) for k, vs in worst.items()), flush=True)
    # the tiny-width goldens use w_std 0.05 / attn_std 0.12-0.15 (chaotic deep chains: bf16 sits at 1.7e-2 there, TOL_DEEP 2.6e-2 in
    # tests/test_gpu_model.py); fp16's 8x finer stores must show up as a several-fold smaller distance from fp32
    assert max(v[1] for v in worst.values()) <= 6e-3, worst
    # generate(): greedy ids, fp16 vs the bf16 model on a text + image prompt
    case = cases.glue_cases()["image_region"]
    ids = case["input_ids"].to(dev)
    o16 = m16.generate(ids, images=[im.to(dev).to(F16) for im in case["images"]], regions=case["regions"], do_sample=False,
                       max_new_tokens=6, eos_token_id=-1)
    assert o16.shape[1] == ids.shape[1] + 6


def test_fp16_load_pretrained_model_defaults_to_the_references_dtype(dev, tmp_path):
    """load_pretrained_model on a checkpoint directory (LoRA layout, tower directories): fp16 unless the caller says otherwise
    (reference builder.py:47,153,161), processors produce fp16 pixels, `torch_dtype=torch.bfloat16` still gives the bf16 build, and the
    two agree to what bf16 storage allows."""
    from tests.test_checkpoint_loader import _write_checkpoint        # the synthetic checkpoint writer of the loader tests
    from vitron_amd.model.builder import load_pretrained_model
    ck = _write_checkpoint(str(tmp_path))
    _, m, proc, _ = load_pretrained_model(ck["ckpt"], ck["base"], "vitron-7b-lora", device="cuda", tokenizer=object())
    assert m.dtype == F16 and m.get_model().llama.dtype == F16 and m.get_image_tower().dtype == F16 and m.get_video_tower().dtype == F16
    assert proc["image"].dtype == F16 and proc["video"].dtype == F16
    _, mb, procb, _ = load_pretrained_model(ck["ckpt"], ck["base"], "vitron-7b-lora", device="cuda", tokenizer=object(),
                                            torch_dtype=torch.bfloat16)
    assert mb.dtype == torch.bfloat16 and procb["image"].dtype == torch.bfloat16
    case = cases.glue_cases()["image_region"]
    ids = case["input_ids"].to(dev)
    a = m(input_ids=ids, images=[im.to(dev).to(F16) for im in case["images"]], regions=case["regions"], use_cache=False).logits
    b = mb(input_ids=ids, images=[im.to(dev).bfloat16() for im in case["images"]], regions=case["regions"], use_cache=False).logits
    assert rel_l2(a, b) <= 2.6e-2          # TOL_DEEP of tests/test_gpu_model.py: the bf16 chain's own distance from fp32 at these widths


def test_precise_modes_on_the_tiny_model_forward_and_generate():
    """model.set_precise(1 | 2) through the reference-shaped surface at the tiny golden widths (fp16 build): forward() on an image +
    region prompt and on a clip moves TOWARDS the oracle's fp32 logits with every level (level 2: towers' MLPs, projector, spliced
    embeddings and every decoder GEMM on operand pairs), generate() runs its prefill in the mode and returns the standard ids."""
    from oracle import vitron_oracle as O
    from tests.golden import cases
    from tests.util import rel_l2
    from vitron_amd import synth
    from vitron_amd.engine import pair_lo
    from vitron_amd.model import LlavaConfig, LlavaLlamaForCausalLM
    dev = torch.device("cuda:0")
    init = dict(w_std=0.02)
    st = {"image_tower": synth.vit_state(cases.VIT_IMAGE, synth.make_generator(cases.SEED_VIT), **init),
          "video_tower": synth.vit_state(cases.VIT_VIDEO, synth.make_generator(cases.SEED_VIT), **init),
          "projector": synth.projector_state(cases.MM_HIDDEN, cases.LLM["hidden_size"], synth.make_generator(cases.SEED_PROJ), **init),
          "region": synth.region_state(cases.MM_HIDDEN, cases.LLM["hidden_size"], synth.make_generator(cases.SEED_REGION), **init),
          "llama": synth.llama_state(cases.LLM, synth.make_generator(cases.SEED_LLM), **init)}
    model = LlavaLlamaForCausalLM(LlavaConfig(**cases.LLM, mm_hidden_size=cases.MM_HIDDEN, mm_image_tower="p/LanguageBind_Image",
                                              mm_video_tower="p/LanguageBind_Video_merge", kv_prefix_reuse=False))
    model.get_image_tower().load_state(cases.VIT_IMAGE, st["image_tower"])
    model.get_video_tower().load_state(cases.VIT_VIDEO, st["video_tower"])
    sd = dict(st["llama"])
    sd.update({"model.mm_projector." + k: v for k, v in st["projector"].items()})
    sd.update({"model.region_extractor." + k: v for k, v in st["region"].items()})
    model.load_state_dict(sd)
    model.to(dev, dtype=torch.float16)
    w = {k: {n: t.float() for n, t in v.items()} for k, v in st.items()}
    cfgs = {"image": cases.VIT_IMAGE, "video": cases.VIT_VIDEO, "llama": cases.LLM}
    for name in ("image_region", "video"):
        case = cases.glue_cases()[name]
        ids = case["input_ids"].to(dev)
        images = [im.to(dev).half() for im in case["images"]]
        ref32 = O.multimodal_forward(w, cfgs, case["input_ids"], None, case["images"], case["regions"])[0]
        dist, toks = {}, {}
        for level in (0, 1, 2):
            model.set_precise(level)
            try:
                (_, _, _, _, embeds, _) = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, images, case["regions"])
                assert (pair_lo(embeds) is not None) == (level == 2)
                out = model(input_ids=ids, images=images, regions=case["regions"], use_cache=False)
                dist[level] = rel_l2(out.logits.float().cpu(), ref32)
                toks[level] = model.generate(ids, images=images, regions=case["regions"], do_sample=False, max_new_tokens=4, eos_token_id=-1)[0, -4:].tolist()
            finally:
                model.set_precise(0)
        print(f"[fp16-precise-tiny] {name}: logits vs fp32 standard {dist[0]:.3e} precise_qk {dist[1]:.3e} precise2 {dist[2]:.3e}", flush=True)
        # (measured on image_region: 4.70e-4 / 4.60e-4 / 3.00e-4 -- at these tiny widths the attention paths and the 16-bit region feature,
        #  which level 2 leaves alone, are a larger share than at the 7B width, where level 2 buys x 0.3)
        assert dist[2] <= 0.75 * dist[0] and dist[2] <= 4.5e-4 and dist[1] <= 1.05 * dist[0], dist
        assert toks[1] == toks[0] and toks[2] == toks[0], toks
