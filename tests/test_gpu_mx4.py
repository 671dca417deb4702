"""Precise level 3's MX-FP4 operators through the C ABI (vt_mx4_quant_weights / vt_mx4_quant_lo / vt_rmsnorm_mx / vt_gemm_mx) against the
oracle's restatement (oracle/vitron_oracle.py mx4_quant): codes and exponents bit for bit, the fused GEMM against the sum of the two
products the oracle's emulation forms (exactly, where the arithmetic is exact: small integers and powers of two)."""
import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]


@pytest.fixture(scope="module")
def dev():
    from vitron_amd import _lib
    _lib.load()
    _lib.load(operand="fp16")
    return torch.device("cuda:0")


def _unpack(codes_u8):
    """uint8 [R][K/2] -> int codes [R][K] (even k in bits 3:0)."""
    c = codes_u8.cpu().to(torch.int32)
    return torch.stack([c & 15, c >> 4], dim=-1).reshape(c.shape[0], -1)


def _aexp_rows(aexp, M, K):
    """the GEMM-order scale array -> biased exponents [M][K/32]"""
    KB = K // 32
    a = aexp.cpu().to(torch.int32)[: ((M + 63) // 64) * KB * 64].view(-1, KB, 16, 4)      # [m / 64][kb][m % 16][(m % 64) / 16]
    m = torch.arange(M)
    return a[m // 64, :, m % 16, (m % 64) // 16]


E2M1 = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])


def _deq(codes, exps_biased, block):
    v = E2M1[codes & 7] * torch.where((codes & 8) != 0, -1.0, 1.0)
    sc = torch.ldexp(torch.ones(exps_biased.shape), exps_biased - 127)
    return (v.view(codes.shape[0], -1, block) * sc.view(codes.shape[0], -1, 1)).reshape(codes.shape)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,K", [(7, 64), (260, 4096), (33, 11008)])
def test_quant_weights_bit_exact(dev, dtype, N, K):
    from oracle import vitron_oracle as O
    from vitron_amd import ops
    g = torch.Generator().manual_seed(N * 7 + K)
    w = (torch.randn((N, K), generator=g) * 0.02).to(dtype)
    w[0, :8] = torch.tensor([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, 6.0]).to(dtype)   # exact ties once the row scale is a power of two
    w[0, 8] = 6.0
    w[1] = 0.0                                                                         # an all-zero row
    w4, wexp = ops.mx4_quant_weights(w.to(dev))
    _, q, e = O.mx4_quant(w.float(), None)
    assert torch.equal(wexp.cpu().to(torch.int32), (e.view(-1) + 127))
    assert torch.equal(_unpack(w4), O.mx4_codes(q))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K", [(5, 64), (300, 4096), (70, 11008)])
def test_quant_lo_bit_exact(dev, dtype, M, K):
    from oracle import vitron_oracle as O
    from vitron_amd import ops
    g = torch.Generator().manual_seed(M * 3 + K)
    v = torch.randn((M, K), generator=g) * torch.rand((M, 1), generator=g) * 8
    hi = v.to(dtype)
    lo = (v - hi.float()).to(dtype)
    lo[0, :32] = 0
    a4, aexp = ops.mx4_quant_lo(lo.to(dev))
    _, q, e = O.mx4_quant(lo.float(), 32)
    assert torch.equal(_aexp_rows(aexp, M, K), e + 127)
    assert torch.equal(_unpack(a4), O.mx4_codes(q))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (300, 260, 384), (1088, 512, 4096), (77, 1024, 1024)])
def test_gemm_mx_exact_on_integers(dev, dtype, M, N, K):
    """Small-integer 16-bit operands and arbitrary 4-bit codes with exponents 126..128: every product and every partial sum is exactly
    representable in fp32, so the launch must return the fp64 result bit for bit -- wrong fragment / scale / op_sel wiring cannot hide."""
    from vitron_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-3, 4, (M, K), generator=g).to(dtype)
    w = torch.randint(-2, 3, (N, K), generator=g).to(dtype)
    a4 = torch.randint(0, 256, (M, K // 2), generator=g, dtype=torch.uint8)
    w4 = torch.randint(0, 256, (N, K // 2), generator=g, dtype=torch.uint8)
    ea = torch.randint(126, 129, (M, K // 32), generator=g)
    ew = torch.randint(126, 128, (N,), generator=g)
    aexp = torch.zeros(ops.mx4_aexp_bytes(M, K), dtype=torch.uint8).view(-1, K // 32, 16, 4)
    m = torch.arange(M)
    aexp[m // 64, :, m % 16, (m % 64) // 16] = ea.to(torch.uint8)
    ref = a.double() @ w.double().t() + _deq(_unpack(a4), ea, 32).double() @ _deq(_unpack(w4), ew.view(-1, 1), K).double().t()
    assert float(ref.abs().max()) * 16 < 2 ** 24 and torch.equal(ref * 16, (ref * 16).round())    # every partial sum is an fp32 value
    got = ops.gemm_mx(a.to(dev), a4.to(dev), aexp.view(-1).to(dev), w.to(dev), w4.to(dev), ew.to(torch.uint8).to(dev), None, ops.EPI_F32)
    assert torch.equal(got.cpu().double(), ref)
    # the 16-bit product alone (all-zero codes) and the 4-bit product alone (zero 16-bit operand)
    z4 = torch.zeros_like(a4)
    got = ops.gemm_mx(a.to(dev), z4.to(dev), aexp.view(-1).to(dev), w.to(dev), w4.to(dev), ew.to(torch.uint8).to(dev), None, ops.EPI_F32)
    assert torch.equal(got.cpu().double(), a.double() @ w.double().t())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(520, 768, 1024), (300, 544, 4096)])
def test_gemm_mx_epilogues_vs_oracle(dev, dtype, M, N, K):
    """The level 3 Linear on real-shaped data against the oracle's _lin_mx (same quantised operands through both products), every epilogue."""
    from oracle import vitron_oracle as O
    from tests.test_gpu_kernels import _gemm_ref
    from vitron_amd import ops
    emu = "fp16" if dtype == torch.float16 else True
    g = torch.Generator().manual_seed(K + M)
    v = torch.randn((M, K), generator=g)
    w = (torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(dtype)
    b = torch.randn((N,), generator=g)
    resid = torch.randn((M, N), generator=g)
    hi = O._r(v, emu)
    lo = (v - hi).to(dtype)
    a4, aexp = ops.mx4_quant_lo(lo.to(dev))
    w4, wexp = ops.mx4_quant_weights(w.to(dev))
    lin = hi.double() @ w.double().t() + O.mx4_quant(lo.float(), 32)[0].double() @ O.mx4_quant(w.float(), None)[0].double().t()
    for epi in (ops.EPI_F32, ops.EPI_BF16, ops.EPI_BF16_GELU, ops.EPI_BF16_QGELU, ops.EPI_F32_RESID, ops.EPI_SWIGLU_BF16):
        bias = None if epi in (ops.EPI_SWIGLU_BF16, ops.EPI_F32_RESID) else b
        out = resid.to(dev).clone() if epi == ops.EPI_F32_RESID else None
        got = ops.gemm_mx(hi.to(dtype).to(dev), a4, aexp, w.to(dev), w4, wexp, None if bias is None else bias.to(dev), epi, out=out)
        y = lin + (0 if bias is None else bias.double())
        if epi == ops.EPI_BF16_GELU:
            y = torch.nn.functional.gelu(y)
        elif epi == ops.EPI_BF16_QGELU:
            y = y * torch.sigmoid(1.702 * y)
        elif epi == ops.EPI_SWIGLU_BF16:
            y4 = y.view(M, N // 32, 2, 16)
            y = (torch.nn.functional.silu(y4[:, :, 0]) * y4[:, :, 1]).reshape(M, N // 2)
        elif epi == ops.EPI_F32_RESID:
            y = y + resid.double()
        ref = y.float() if epi in (ops.EPI_F32, ops.EPI_F32_RESID) else O._r(y.float(), emu)
        assert got.shape == ref.shape
        assert rel_l2(got.float(), ref) <= (2e-5 if epi in (ops.EPI_F32, ops.EPI_F32_RESID) else 1e-3), epi
    # and what the second product buys: distance from the exact product of the fp32 values
    exact = v.double() @ w.double().t()
    got = ops.gemm_mx(hi.to(dtype).to(dev), a4, aexp, w.to(dev), w4, wexp, None, ops.EPI_F32)
    plain = ops.gemm(hi.to(dtype).to(dev), w.to(dev), None, ops.EPI_F32)
    d_mx, d_plain = rel_l2(got, exact), rel_l2(plain, exact)
    assert d_mx < 0.3 * d_plain, (d_mx, d_plain)


@pytest.mark.parametrize("dtype", DTYPES)
def test_rmsnorm_mx(dev, dtype):
    from oracle import vitron_oracle as O
    from vitron_amd import ops
    g = torch.Generator().manual_seed(5)
    rows, D = 300, 4096
    x = torch.randn((rows, D), generator=g) * (1 + 60 * (torch.rand((rows, 1), generator=g) < 0.3))
    w = 1 + 0.1 * torch.randn((D,), generator=g)
    y, a4, aexp = ops.rmsnorm_mx(x.to(dev), w.to(dev), 1e-5, dtype)
    # the 16-bit half is vt_rmsnorm's output (same expressions; the two kernels may contract the sum of squares differently: an ulp of rstd)
    y0 = ops.rmsnorm(x.to(dev), w.to(dev), 1e-5, dtype=dtype).cpu().float()
    assert rel_l2(y.cpu().float(), y0) < 1e-4 and float((y.cpu().float() != y0).float().mean()) < 1e-2
    v = O.rmsnorm(x, w, 1e-5)
    lo_ref = v - y.cpu().float()
    lo_got = _deq(_unpack(a4), _aexp_rows(aexp, rows, D), 32)
    # the remainder's 4-bit image carries it to ~13 % (e2m1 against a block scale); hi + image must be far closer to v than hi alone
    assert rel_l2(lo_got, lo_ref) < 0.2
    assert rel_l2(y.cpu().float() + lo_got, v) < 0.25 * rel_l2(y.cpu().float(), v)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 256, 512), (700, 1408, 1024)])
def test_gemm_mx_swiglu_operand_out(dev, dtype, M, N, K):
    """Level 3 gate/up: the 16-bit output is the plain SwiGLU epilogue's, bit for bit; its 4-bit image restores most of what the 16-bit
    store dropped (checked against the fp64 value of the same two products)."""
    from oracle import vitron_oracle as O
    from vitron_amd import ops
    emu = "fp16" if dtype == torch.float16 else True
    g = torch.Generator().manual_seed(N + K)
    v = torch.randn((M, K), generator=g)
    w = (torch.randn((N, K), generator=g) * 0.05).to(torch.bfloat16).to(dtype)
    hi = O._r(v, emu)
    lo = (v - hi).to(dtype)
    a4, aexp = ops.mx4_quant_lo(lo.to(dev))
    w4, wexp = ops.mx4_quant_weights(w.to(dev))
    a = hi.to(dtype).to(dev)
    h, h4, hexp = ops.gemm_mx_swiglu(a, a4, aexp, w.to(dev), w4, wexp)
    base = ops.gemm_mx(a, a4, aexp, w.to(dev), w4, wexp, None, ops.EPI_SWIGLU_BF16)
    assert torch.equal(h.cpu(), base.cpu())
    y = hi.double() @ w.double().t() + O.mx4_quant(lo.float(), 32)[0].double() @ O.mx4_quant(w.float(), None)[0].double().t()
    y4 = y.view(M, N // 32, 2, 16)
    val = (torch.nn.functional.silu(y4[:, :, 0]) * y4[:, :, 1]).reshape(M, N // 2).float()
    img = _deq(_unpack(h4), _aexp_rows(hexp, M, N // 2), 32)
    rem = val - h.cpu().float()
    assert rel_l2(img, rem) < 0.2, rel_l2(img, rem)
    assert rel_l2(h.cpu().float() + img, val) < 0.25 * rel_l2(h.cpu().float(), val)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(5120, 4096, 512), (1300, 1024, 768), (300, 512, 1024)])
def test_gemm_mx_resid_with_split_k_tail(dev, dtype, M, N, K):
    """x += A.W^T + A4.W4^T with the trailing row blocks as K ranges (5120 x 4096: one round + 64 tiles x 4 ranges) == one plain launch, up to
    the order of fp32 additions."""
    from oracle import vitron_oracle as O
    from vitron_amd import ops
    emu = "fp16" if dtype == torch.float16 else True
    g = torch.Generator().manual_seed(M + K)
    v = torch.randn((M, K), generator=g)
    w = (torch.randn((N, K), generator=g) * 0.05).to(torch.bfloat16).to(dtype)
    x0 = torch.randn((M, N), generator=g)
    hi = O._r(v, emu)
    a4, aexp = ops.mx4_quant_lo((v - hi).to(dtype).to(dev))
    w4, wexp = ops.mx4_quant_weights(w.to(dev))
    a = hi.to(dtype).to(dev)
    plain = ops.gemm_mx_resid(a, a4, aexp, w.to(dev), w4, wexp, x0.to(dev).clone())
    part = torch.empty((256 * 256 * 256,), device=dev, dtype=torch.float32)
    split = ops.gemm_mx_resid(a, a4, aexp, w.to(dev), w4, wexp, x0.to(dev).clone(), part)
    one = ops.gemm_mx(a, a4, aexp, w.to(dev), w4, wexp, None, ops.EPI_F32_RESID, out=x0.to(dev).clone())
    assert torch.equal(plain.cpu(), one.cpu())
    assert rel_l2(split, plain) < 2e-6
    if M <= 1300:
        ref = x0.double() + hi.double() @ w.double().t() + O.mx4_quant((v - hi).to(dtype).float(), 32)[0].double() @ O.mx4_quant(w.float(), None)[0].double().t()
        assert rel_l2(split, ref) < 2e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("heads,lens", [(3, [600, 70]), (32, [2304])])
def test_flash_attn_mx_operand_out(dev, dtype, heads, lens):
    """The attention kernels' level 3 output (two-waves-per-SIMD kernel for the short problem, one-wave-per-SIMD kernel for the long one): O is the
    standard launch's, bit for bit; with q = 0 every visible key weighs the same and v holds small integers, so the fp32 value behind O is known
    exactly (sum / count, the kernel's own expression) and the image must be the quantiser's image of its remainder, code for code."""
    import math
    from oracle import vitron_oracle as O
    from tests.test_gpu_attn_w4 import _problem
    from vitron_amd import ops
    op = "fp16" if dtype == torch.float16 else "bf16"
    hd, D = 128, heads * 128
    q, kt, vt, table, desc, kv_full = _problem(dev, dtype, heads, lens, [0] * len(lens))
    scale = 1.0 / math.sqrt(hd)
    o, o4, oexp = ops.flash_attn_mx(q, kt, vt, table, desc, max(lens), heads, scale)
    base = ops.flash_attn(q, kt, vt, table, desc, max(lens), heads, hd, True, scale)
    assert torch.equal(o.cpu(), base.cpu())
    rows = sum(lens)
    img = _deq(_unpack(o4), _aexp_rows(oexp, rows, D), 32)
    assert torch.isfinite(img).all() and float(img.abs().max()) <= float(o.float().abs().max()) * (2 ** -8 if op == "bf16" else 2 ** -11) * 1.01
    # exact case: q = 0, integer v
    g = torch.Generator().manual_seed(9)
    xs = []
    for n in lens:
        x = torch.zeros((n, 3 * D))
        x[:, D:2 * D] = torch.randn((n, D), generator=g)
        x[:, 2 * D:] = torch.randint(-8, 9, (n, D), generator=g).float()
        xs.append(x.to(dev).to(dtype))
    ntl = [(n + 63) // 64 for n in lens]
    kt2 = torch.zeros_like(kt)
    vt2 = torch.zeros_like(vt)
    toff = 0
    for i, x in enumerate(xs):
        d = torch.tensor([[0, lens[i], lens[i], toff]], dtype=torch.int32, device=dev)
        ops.kv_tiles(x, 0, D, 2 * D, kt2, vt2, table, d, ntl[i], heads, hd)
        toff += ntl[i]
    q0 = torch.cat([x[:, :D] for x in xs], 0).contiguous()
    o, o4, oexp = ops.flash_attn_mx(q0, kt2, vt2, table, desc, max(lens), heads, scale)
    vals = []
    for x, n in zip(xs, lens):
        cs = torch.cumsum(x[:, 2 * D:].float().cpu(), 0)                      # exact: small integers
        cnt = torch.arange(1, n + 1, dtype=torch.float32)[:, None]
        # the kernel's expression: oacc * (1 / l) with P = 2^7 on every visible key (P_BIAS): (128 sum) * (1 / (128 count)) in fp32
        vals.append((cs * 128.0) * (1.0 / (cnt * 128.0)))
    val = torch.cat(vals, 0)
    emu = "fp16" if dtype == torch.float16 else True
    hi = O._r(val, emu)
    assert torch.equal(o.cpu().float(), hi)
    _, qv, e = O.mx4_quant(val - hi, 32)
    # the two-waves-per-SIMD kernel evaluates exactly this expression: exponents and codes must match one for one. The one-wave-per-SIMD kernel
    # folds the scale into its exponent differently (its fp32 value can sit an ulp away from ours, which moves a remainder of 2^-12 of the value
    # by 2^-12 of itself): there the images may differ where a remainder sits on a rounding boundary
    e_got, c_got, c_ref = _aexp_rows(oexp, rows, D), _unpack(o4), O.mx4_codes(qv)
    if heads == 3:
        assert torch.equal(e_got, e + 127)
        assert torch.equal(c_got & 7, c_ref & 7)                                   # (the sign of a zero code is not compared)
        nz = (c_ref & 7) != 0
        assert torch.equal((c_got & 8)[nz], (c_ref & 8)[nz])
    else:
        assert float((e_got != e + 127).float().mean()) < 2e-2
        same_e = (e_got == e + 127).repeat_interleave(32, dim=1)
        assert float(((c_got & 7) != (c_ref & 7))[same_e].float().mean()) < 2e-2
        img = _deq(c_got, e_got, 32)
        assert rel_l2(img, val - hi) < 0.2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("quick", [False, True])
def test_tower_mlp_on_the_mx_pipe(dev, dtype, quick):
    """The towers' precise level 1 at ViT-L width: LayerNorm with the operand out -> fc1 + GELU with the operand out -> fc2 into the residual stream
    (one whole-problem split-K launch pair: 76 tiles), against fp64 of the exact values -- the MLP's distance from it must be a fraction of
    the standard 16-bit operators' on the same input."""
    from vitron_amd import ops
    g = torch.Generator().manual_seed(11)
    R, D, I = 1154, 1024, 4096
    x = torch.randn((R, D), generator=g) * 3
    gam, bet = 1 + 0.1 * torch.randn((D,), generator=g), 0.1 * torch.randn((D,), generator=g)
    w1 = (torch.randn((I, D), generator=g) * 0.03).to(torch.bfloat16).to(dtype)
    w2 = (torch.randn((D, I), generator=g) * 0.02).to(torch.bfloat16).to(dtype)
    b1, b2 = 0.1 * torch.randn((I,), generator=g), 0.1 * torch.randn((D,), generator=g)
    xd = x.to(dev)
    y, y4, yexp = ops.layernorm_mx(xd, gam.to(dev), bet.to(dev), 1e-5, dtype)
    y0 = ops.layernorm(xd.clone(), gam.to(dev), bet.to(dev), 1e-5, dtype=dtype)
    assert rel_l2(y.float().cpu(), y0.float().cpu()) < 1e-4
    w14, w1e = ops.mx4_quant_weights(w1.to(dev))
    w24, w2e = ops.mx4_quant_weights(w2.to(dev))
    h, h4, hexp = ops.gemm_mx_gelu(y, y4, yexp, w1.to(dev), w14, w1e, b1.to(dev), quick)
    part = torch.empty((256 * 256 * 64,), device=dev, dtype=torch.float32)
    out = ops.gemm_mx_resid(h, h4, hexp, w2.to(dev), w24, w2e, xd.clone(), part)
    # fp64 of the exact values (no rounding anywhere); the bias of fc2 is added on the host (the public resid entry takes none)
    ln = torch.nn.functional.layer_norm(x.double(), (D,), gam.double(), bet.double(), 1e-5)
    a = ln @ w1.double().t() + b1.double()
    act = a * torch.sigmoid(1.702 * a) if quick else torch.nn.functional.gelu(a)
    mlp_ref = act @ w2.double().t()
    mlp_got = out.double().cpu() - x.double()
    # the standard operators on the same input
    hs = ops.gemm(y0, w1.to(dev), b1.to(dev), ops.EPI_BF16_QGELU if quick else ops.EPI_BF16_GELU)
    std = ops.gemm(hs, w2.to(dev), None, ops.EPI_F32_RESID, out=xd.clone()).double().cpu() - x.double()
    d_mx, d_std = rel_l2(mlp_got, mlp_ref), rel_l2(std, mlp_ref)
    assert d_mx < 0.35 * d_std, (d_mx, d_std)
    assert rel_l2(h.float().cpu(), hs.float().cpu()) < (3e-3 if dtype == torch.bfloat16 else 4e-4)      # the 16-bit halves agree up to an ulp of a few elements
