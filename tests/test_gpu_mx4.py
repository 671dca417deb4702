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
