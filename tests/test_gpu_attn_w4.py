"""flash_attn_w4_kernel (vitron_amd/csrc/vt_attn_w4.hip; round 4): the one-wave-per-SIMD prefill attention kernel for head_dim 128 --
64 query rows per wave, 32-key sub tiles software-pipelined three deep, the softmax placed into the MFMA gaps by
tools/gen_attn_w4.py -- held to (a) fp64 attention on the same 16-bit q / k / v, (b) the two-waves-per-SIMD kernel it replaces for long
sequences, (c) its own unplaced form (the same pipeline with the stages run one after the other), in both operand builds, on ragged /
multi-sequence / chunked (past > 0) / non-causal / single-tile / long problems over a shuffled page table, and at the score-range edges
of the fp16 softmax weights. vt_flash_attn_select picks the kernel (include/vitron_hip.h)."""
import math

import pytest
import torch

from tests.util import randn, rel_l2

pytestmark = pytest.mark.gpu
DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


@pytest.fixture(scope="module")
def dev():
    from vitron_amd import _lib
    _lib.load()
    _lib.load(operand="fp16")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _auto_kernel_afterwards():
    yield
    from vitron_amd import ops
    ops.flash_attn_select(0)


def _problem(dev, dt, heads, lens, pasts, seed=200, scale_qk=1.0):
    from vitron_amd import ops
    hd = 128
    D = heads * hd
    kv_lens = [p + q for p, q in zip(pasts, lens)]
    ntl = [(n + 63) // 64 for n in kv_lens]
    npages = sum(ntl)
    perm = torch.randperm(npages, generator=torch.Generator().manual_seed(5)).tolist()
    kt = torch.full((npages * heads * 64 * hd,), float("nan"), dtype=dt, device=dev)
    vt = torch.full((npages * heads * 64 * hd,), float("nan"), dtype=torch.float16, device=dev)
    table, desc_new, q_rows, kv_full = [], [], [], []
    row0 = 0
    for i, (p, q) in enumerate(zip(pasts, lens)):
        toff = len(table)
        table += perm[toff:toff + ntl[i]]
        x = randn((p + q, 3 * D), seed + i)
        x[:, :2 * D] *= scale_qk
        x = x.to(dev).to(dt)
        kv_full.append(x)
        desc_new.append([row0, q, p + q, toff])
        row0 += q
    table_t = torch.tensor(table, dtype=torch.int32, device=dev)
    for i, x in enumerate(kv_full):
        d = torch.tensor([[0, kv_lens[i], kv_lens[i], desc_new[i][3]]], dtype=torch.int32, device=dev)
        ops.kv_tiles(x, 0, D, 2 * D, kt, vt, table_t, d, ntl[i], heads, hd)
        q_rows.append(x[pasts[i]:, :D])
    return torch.cat(q_rows, 0).contiguous(), kt, vt, table_t, torch.tensor(desc_new, dtype=torch.int32, device=dev), kv_full


def _reference(kv_full, heads, lens, pasts, causal, dt):
    hd = 128
    D = heads * hd
    scale = 1.0 / math.sqrt(hd)
    outs = []
    for x, p, q in zip(kv_full, pasts, lens):
        x = x.double().cpu()
        qq = x[p:, :D].view(q, heads, hd)
        kk, vv = x[:, D:2 * D].view(p + q, heads, hd), x[:, 2 * D:].view(p + q, heads, hd)
        s = torch.einsum("qhd,khd->hqk", qq, kk) * scale
        if causal:
            mask = torch.arange(p + q)[None, :] > (p + torch.arange(q))[:, None]
            s = s.masked_fill(mask[None], float("-inf"))
        outs.append(torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), vv).reshape(q, D).float())
    return torch.cat(outs, 0).to(dt).float()


CASES = [(True, [700, 300, 64, 1], [0, 0, 0, 0]), (False, [577, 130], [0, 0]), (True, [257, 40], [100, 1000]), (True, [2304], [0]),
         (True, [256, 255, 257], [0, 31, 64]), (False, [64], [0]), (True, [1], [0])]


@pytest.mark.parametrize("op", ["bf16", "fp16"])
@pytest.mark.parametrize("kernel", [2, 3, 4, 4 | (8 << 8), 5])
@pytest.mark.parametrize("causal,lens,pasts", CASES)
def test_w4_kernel_vs_fp64_and_vs_the_two_wave_kernel(dev, causal, lens, pasts, kernel, op):
    from vitron_amd import ops
    dt = DT[op]
    heads, hd = 3, 128
    q, kt, vt, table, desc, kv_full = _problem(dev, dt, heads, lens, pasts)
    scale = 1.0 / math.sqrt(hd)
    ops.flash_attn_select(1)
    base = ops.flash_attn(q, kt, vt, table, desc, max(lens), heads, hd, causal, scale).float().cpu()
    ops.flash_attn_select(kernel)
    got = ops.flash_attn(q, kt, vt, table, desc, max(lens), heads, hd, causal, scale)
    torch.cuda.synchronize()
    got = got.float().cpu()
    assert torch.isfinite(got).all()
    ref = _reference(kv_full, heads, lens, pasts, causal, dt)
    tol = 1.3e-3 if op == "bf16" else 6e-4          # the two-wave kernel's own bounds (tests/test_gpu_kernels.py, tests/test_gpu_fp16.py)
    e_new, e_old = rel_l2(got, ref), rel_l2(base, ref)
    assert e_new <= tol, (e_new, e_old)
    assert e_new <= 1.25 * e_old + 1e-4, (e_new, e_old)               # no less accurate than the kernel it replaces
    assert rel_l2(got, base) <= tol, rel_l2(got, base)                # and the two agree to a few output roundings


@pytest.mark.parametrize("op", ["bf16", "fp16"])
def test_w4_placed_equals_unplaced_bit_for_bit(dev, op):
    """Same arithmetic, same order per row: the placement must not change a bit."""
    from vitron_amd import ops
    dt = DT[op]
    heads, hd, lens, pasts = 4, 128, [1300, 70], [0, 200]
    q, kt, vt, table, desc, _ = _problem(dev, dt, heads, lens, pasts, seed=300)
    scale = 1.0 / math.sqrt(hd)
    outs = []
    for kernel in (2, 3, 4, 4 | (3 << 8)):
        ops.flash_attn_select(kernel)
        outs.append(ops.flash_attn(q, kt, vt, table, desc, max(lens), heads, hd, True, scale).clone())
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])


@pytest.mark.parametrize("op", ["bf16", "fp16"])
@pytest.mark.parametrize("heads,lens,pasts,cap", [(8, [1024, 768, 512, 256], [0, 0, 0, 0], 8), (8, [1024, 300, 1280, 64], [0, 100, 0, 256], 16),
                                                  (3, [1536, 700], [0, 0], 2), (16, [2048], [0], 0), (8, [512, 512, 512], [64, 0, 128], 8)])
def test_w4_persistent_form_walks_the_block_list(dev, op, heads, lens, pasts, cap):
    """The persistent form (vt_flash_attn_select(4)): workgroups take (sequence, block, head) tickets per XCD, the K / V^T stream runs on
    across block seams whenever a block has a multiple of 4 tiles (prefill blocks without past: warm seams) and restarts cold otherwise
    (ragged tails, blocks with past) -- mixed here, with the workgroup count capped so that every workgroup walks many blocks. Same
    arithmetic per row as the one-block-per-workgroup launch: bit-identical output, also on the second launch (the ticket counters reset
    themselves) and next to rows the launch must not touch."""
    from vitron_amd import ops
    dt = DT[op]
    hd = 128
    q, kt, vt, table, desc, _ = _problem(dev, dt, heads, lens, pasts, seed=500)
    scale = 1.0 / math.sqrt(hd)
    ops.flash_attn_select(2)
    want = ops.flash_attn(q, kt, vt, table, desc, max(lens), heads, hd, True, scale).clone()
    ops.flash_attn_select(4 | (cap << 8))
    for rep in range(3):
        out = torch.full_like(want, float("nan"))
        ops.flash_attn(q, kt, vt, table, desc, max(lens), heads, hd, True, scale, out=out)
        torch.cuda.synchronize()
        assert torch.equal(out, want), (rep, heads, lens)
    ops.flash_attn_select(2)
    want_nc = ops.flash_attn(q, kt, vt, table, desc, max(lens), heads, hd, False, scale).clone()
    ops.flash_attn_select(4 | (cap << 8))
    got_nc = ops.flash_attn(q, kt, vt, table, desc, max(lens), heads, hd, False, scale)
    torch.cuda.synchronize()
    assert torch.equal(got_nc, want_nc)


@pytest.mark.parametrize("variant", ["wide", "ascending"])
def test_w4_kernel_score_range_edges(dev, variant):
    """Rows whose scores span > 60 (peaked rows) and rows whose running maximum jumps at every sub tile (keys sorted ascending in score):
    the deferred rescale and the pending O rescale between sub-iterations."""
    from vitron_amd import ops
    hd, heads, L = 128, 2, 700
    D = heads * hd
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn((L, 3 * D), generator=g)
    if variant == "wide":
        qkv[:, :2 * D] *= 3.0
    else:
        u = torch.sign(torch.randn((1, D), generator=g))
        qkv[:, D:2 * D] = torch.linspace(-3.0, 3.0, L).view(L, 1) * u + 0.05 * qkv[:, D:2 * D]
        qkv[:, :D] = (1.0 + torch.rand((L, 1), generator=g)) * u + 0.05 * qkv[:, :D]
    qd = qkv.to(dev).bfloat16()
    npages = (L + 63) // 64
    kt = torch.zeros(npages * heads * 64 * hd, dtype=torch.bfloat16, device=dev)
    vt = torch.zeros(npages * heads * 64 * hd, dtype=torch.float16, device=dev)
    table, desc = torch.arange(npages, dtype=torch.int32, device=dev), torch.tensor([[0, L, L, 0]], dtype=torch.int32, device=dev)
    ops.kv_tiles(qd, 0, D, 2 * D, kt, vt, table, desc, npages, heads, hd)
    scale = 1.0 / math.sqrt(hd)
    x = qd.double().cpu()
    q, k, v = (x[:, i * D:(i + 1) * D].view(L, heads, hd) for i in range(3))
    s = torch.einsum("qhd,khd->hqk", q, k) * scale
    assert float(s.max() - s.min()) > 60.0
    s = s.masked_fill((torch.arange(L)[None, :] > torch.arange(L)[:, None])[None], float("-inf"))
    ref = torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), v).reshape(L, D).float().bfloat16().float()
    for kernel in (2, 3, 4):
        ops.flash_attn_select(kernel)
        out = ops.flash_attn(qd, kt, vt, table, desc, L, heads, hd, True, scale).float().cpu()
        err = rel_l2(out, ref)
        print(f"[w4-edges] {variant} kernel {kernel}: rel_l2 {err:.3e}", flush=True)
        assert torch.isfinite(out).all() and err <= 1.1e-3, (variant, kernel, err)


@pytest.mark.parametrize("op", ["bf16", "fp16"])
def test_w4_planned_dispatch_order_changes_nothing_but_the_schedule(dev, op):
    """One long causal sequence on 32 heads (here 2304 rows: 288 blocks, more than one round of the 256 CUs): kernel 2 dispatches its
    blocks in the order vt_flash_attn_block_order planned, kernel 5 in the grid's natural order -- the same blocks, the same arithmetic
    per block: identical bits. (The timing side is tools/attn_bench.py ATTN_BENCH_KERNELS=2,5.)"""
    from vitron_amd import ops
    heads, hd = 32, 128
    scale = 1.0 / math.sqrt(hd)
    for S, past in ((2304, 0), (2100, 192), (5120, 0)):
        q, kt, vt, table, desc, _ = _problem(dev, DT[op], heads, [S], [past], seed=700)
        outs = []
        for kernel in (5, 2):
            ops.flash_attn_select(kernel)
            outs.append(ops.flash_attn(q, kt, vt, table, desc, S, heads, hd, True, scale).clone())
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]), (S, past)


def test_w4_kernel_is_what_a_long_prefill_runs(dev):
    """Automatic selection: the decoder's S = 5120 prefill (32 heads: 640 workgroups) goes to the one-wave-per-SIMD kernel, a single
    1088-row prompt (160 workgroups of 256 rows) stays on 128-row blocks; both give the result of the forced choice bit for bit."""
    from vitron_amd import ops
    heads, hd = 32, 128
    scale = 1.0 / math.sqrt(hd)
    for S, forced in ((2304, 2), (1088, 1)):
        q, kt, vt, table, desc, _ = _problem(dev, torch.bfloat16, heads, [S], [0], seed=400)
        ops.flash_attn_select(0)
        auto = ops.flash_attn(q, kt, vt, table, desc, S, heads, hd, True, scale).clone()
        ops.flash_attn_select(forced)
        want = ops.flash_attn(q, kt, vt, table, desc, S, heads, hd, True, scale).clone()
        assert torch.equal(auto, want), S
