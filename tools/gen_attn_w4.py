#!/usr/bin/env python3
"""Generator of the hand-placed sub-iterations of flash_attn_w4_kernel (vitron_amd/csrc/vt_attn_w4.hip).

    python tools/gen_attn_w4.py            -> vitron_amd/csrc/vt_attn_w4_si0.inc, vt_attn_w4_si1.inc

One sub-iteration SI(sub), sub = 0 / 1, of the steady-state loop body of tile T runs
    P.V(T, sub)  ||  softmax of the OTHER S buffer (sub tile (T, 1) in SI0, (T+1, 0) in SI1)  ||  Q.K^T(T+1, sub)
as 32 MFMAs (score and P.V MFMAs alternating: dependent MFMAs are 4 / 16 issues apart) with everything else placed into the 32 gaps
behind them, every gap fenced by sched_barrier(0) so that the compiler keeps the placement (it still allocates the registers and
counts lgkmcnt for the fragment reads). What goes where:

  fragment windows   K fragments of k-step ks live in kfr[ks % 3], V^T fragments p = (j, db) in vfr[p % 3]; a read may only be issued
                     behind the last MFMA that uses the register it overwrites, and should sit >= 3 gaps ahead of its first use
  K reads            ks = 3..7 of this sub-iteration, then ks' = 2, 0, 1 of the NEXT one (gaps 23 / 27 / 31)
  V^T reads          p = 3..7, then p' = 2, 0, 1 of the next sub-iteration (gaps 24 / 28 / 31)
  LDS-DMA (SI0)      the 4 K pieces of tile T+3 in gaps 0..6 (they are waited for at the end of the tile), its 4 V^T pieces later
  softmax            row max (two interleaved chains of 8 v_max3), the half-wave exchange and the wave-wide rescale test, then 32 x
                     (v_fma, v_exp, v_add) + 16 v_cvt_pk_f16 spread evenly over the remaining gaps, row-half 0 and 1 interleaved

The VALU stream of a gap is ONE multi-instruction asm statement:
  * compiler-visible arithmetic (builtins) does not stay where it is written -- instruction selection sinks it to its use, across
    the sched_barriers, which only bind the machine scheduler;
  * one asm statement per instruction does stay, but the hazard recognizer counts an asm statement as zero wait states and puts an
    s_nop between every dependent pair of them (one per exponential);
  * a block per gap stays in place, carries its own hazard distances (a transcendental result is consumed at least one instruction
    later) and meets the next dependent block behind an MFMA and, usually, a fragment read.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "vitron_amd", "csrc")


def mfma(n, sub):
    q = n // 2
    if n % 2 == 0:      # score MFMA q: k-step ks, row half rh
        ks, rh = q // 2, q % 2
        form = "W4A_QK0" if ks == 0 else "W4A_QK"
        return f"{form}(sacc[{sub}][{rh}], kfr[{ks % 3}], qf[{rh}][{ks}]);"
    p, rh = q // 2, q % 2          # P.V MFMA: fragment p = (j, db), row half rh
    j, db = p // 4, p % 4
    return f"W4A_PV(oacc[{rh}][{db}], vfr[{p % 3}], pfr[{sub}][{rh}][{j}]);"


def memory_ops(sub):
    """{gap: [statements]} of the fragment reads and the DMA pieces."""
    g = {}

    def put(gap, stmt):
        g.setdefault(gap, []).append(stmt)

    # K fragments of THIS sub-iteration: tile T+1 (slot so_n), sub
    # (read positions: K reads in gaps = 2 mod 4, V^T reads in gaps = 1 mod 4; the compiler's lgkmcnt waits fall into gaps = 3 and = 0
    #  mod 4, right before the consuming MFMAs -- so every gap carries one compiler-visible instruction, and the hazard recognizer,
    #  which counts an asm block as zero wait states, finds its wait state between two dependent blocks without adding an s_nop)
    for ks in range(3, 8):
        put(4 * ks - 10, f"kfr[{ks % 3}] = *(const bf16x8*)(smem + so_n + {sub * 8192} + kfix[{ks}]);")
    # ... and the first three of the next one: SI0 -> (T+1, sub 1), SI1 -> (T+2, sub 0)
    nxt_k = "so_n + 8192" if sub == 0 else "so_nn"
    for ks, gap in ((2, 22), (0, 26), (1, 30)):
        put(gap, f"kfr[{ks}] = *(const bf16x8*)(smem + {nxt_k} + kfix[{ks}]);")
    # V^T fragments of THIS sub-iteration: tile T (slot so_c), sub
    for p in range(3, 8):
        j, db = p // 4, p % 4
        put(4 * p - 7, f"vfr[{p % 3}] = *(const f16x8*)(smem + so_c + {db * 4096} + vfix[{sub}][{j}]);")
    nxt_v = ("so_c", 1) if sub == 0 else ("so_n", 0)
    for p, gap in ((2, 25), (0, 29), (1, 31)):
        put(gap, f"vfr[{p}] = *(const f16x8*)(smem + {nxt_v[0]} + {p * 4096} + vfix[{nxt_v[1]}][0]);")
    if sub == 0:
        for i, gap in enumerate((0, 3, 4, 7)):
            put(gap, f"W4A_DMA_K(rk, slot_d, {i});")
        for i, gap in enumerate((11, 15, 19, 23)):
            put(gap, f"W4A_DMA_V(rv, slot_d, {i});")
    return g


class Block:
    """One asm statement: instructions over symbolic operands -> text + constraint lists."""

    def __init__(self):
        self.ins = []          # (opcode, dst, [srcs])

    def add(self, op, dst, *srcs):
        self.ins.append((op, dst, list(srcs)))

    def __len__(self):
        return len(self.ins)

    def emit(self, indent="  "):
        if not self.ins:
            return []
        order, first_is_write, written, read = [], {}, set(), set()
        for op, dst, srcs in self.ins:
            for s in srcs:
                if s not in first_is_write:
                    first_is_write[s] = False
                    order.append(s)
                read.add(s)
            if dst not in first_is_write:
                first_is_write[dst] = True
                order.append(dst)
            written.add(dst)
        outs, ins = [], []
        for s in order:
            if s in written:
                outs.append(s)
            else:
                ins.append(s)
        idx = {s: i for i, s in enumerate(outs + ins)}
        lines = []
        for op, dst, srcs in self.ins:
            lines.append(f"{op} %{idx[dst]}, " + ", ".join(f"%{idx[s]}" for s in srcs))
        text = "\\n\\t".join(lines)

        def cons(s, out):
            if out:
                # read before the first write inside the block -> in/out; else early-clobber output (the block reads other operands later)
                return f'"+v"({s})' if not first_is_write[s] else f'"=&v"({s})'
            return f'"s"({s})' if s == "scale_log2e" else f'"v"({s})'
        o = ", ".join(cons(s, True) for s in outs)
        i = ", ".join(cons(s, False) for s in ins)
        return [f'{indent}asm volatile("{text}" : {o} : {i});']


def softmax_program(sub):
    """(max blocks: list of Blocks for gaps 0.., exp instruction list [(op, dst, srcs)], tail statements)."""
    sb = 1 - sub
    S = lambda rh, i: f"sacc[{sb}][{rh}][{i}]"      # noqa: E731
    # row max of the 16 scores of a lane as a TREE (5 independent v_max3, then 2, then 1: depth 3) -- a chain of 8 dependent v_max3
    # per row half makes the first four gaps of a sub-iteration wait on VALU latency
    chains = []
    for rh in range(2):
        t = [f"mx{rh}t{i}" for i in range(5)]
        c = [("v_max3_f32", t[i], [S(rh, 3 * i), S(rh, 3 * i + 1), S(rh, 3 * i + 2)]) for i in range(5)]
        c += [("v_max3_f32", t[0], [t[0], t[1], t[2]]),
              ("v_max3_f32", t[3], [t[3], t[4], S(rh, 15)]),
              ("v_max_f32", f"mx{rh}", [t[0], t[3]])]
        chains.append(c)
    maxins = []
    for a, b in zip(*chains):
        maxins += [a, b]
    ex = []
    pending_cvt = None          # the pack of a pair is issued behind the NEXT pair's exponentials: a transcendental result may not be
    for k in range(8):          # consumed by the instruction right behind it (pair k: scores 2k, 2k+1 -> word w of fragment j)
        j, w = k // 4, k % 4
        for rh in range(2):
            i0, i1 = 2 * k, 2 * k + 1
            # temporaries, reused pair after pair (per row half); the first pair's exponentials ARE the row-sum accumulators
            ya, yb = (f"ps{rh}a", f"ps{rh}b") if k == 0 else (f"y{rh}a", f"y{rh}b")
            ex += [("v_fma_f32", ya, [S(rh, i0), "scale_log2e", f"mb[{rh}]"]),
                   ("v_fma_f32", yb, [S(rh, i1), "scale_log2e", f"mb[{rh}]"]),
                   ("v_exp_f32", ya, [ya]),
                   ("v_exp_f32", yb, [yb])]
            if pending_cvt is not None:
                ex.append(pending_cvt)
            if k > 0:
                ex += [("v_add_f32", f"ps{rh}a", [f"ps{rh}a", ya]),
                       ("v_add_f32", f"ps{rh}b", [f"ps{rh}b", yb])]
            pending_cvt = ("v_cvt_pk_f16_f32", f"pfr[{sb}][{rh}][{j}][{w}]", [ya, yb])
    ex.append(pending_cvt)
    # (the last pack follows two adds: its inputs were written three instructions earlier)
    for n in range(1, len(ex)):
        if ex[n - 1][0] == "v_exp_f32":
            assert ex[n - 1][1] not in ex[n][2], (n, ex[n - 1], ex[n])
    return maxins, ex


def generate(sub):
    mem = memory_ops(sub)
    maxins, ex = softmax_program(sub)
    sb = 1 - sub
    n_gaps = 32
    max_gaps = 4                      # 16 v_max3/v_max: 4 per gap
    out = [f"// generated by tools/gen_attn_w4.py -- sub-iteration SI{sub} of flash_attn_w4_kernel; do not edit by hand",
           "{",
           "  if constexpr (!(ABL & 128)) apply_pending();",
           f"  if constexpr (!(ABL & 128)) mask_stage(sacc[{sb}], {'T * 64 + 32' if sub == 0 else '(T + 1) * 64'});",
           "  float mx0, mx1, mxh[2];",
           "  float mx0t0, mx0t1, mx0t2, mx0t3, mx0t4, mx1t0, mx1t1, mx1t2, mx1t3, mx1t4;",
           "  float y0a, y0b, y1a, y1b;",
           "  float ps0a, ps0b, ps1a, ps1b;"]
    counts = []
    qi = 0
    for gap in range(n_gaps):
        out.append(f"  // ---- MFMA {gap}")
        out.append(f"  if constexpr (!(ABL & {16 if gap % 2 else 32})) " + mfma(gap, sub))      # ablations 16 / 32: no P.V / no score MFMAs
        blk = Block()
        post = []
        if gap < max_gaps:
            share = -(-len(maxins) // max_gaps)
            for ins in maxins[gap * share:(gap + 1) * share]:
                blk.add(ins[0], ins[1], *ins[2])
            if gap == max_gaps - 1:
                # wave-wide test on the half-row maxima (two compares + a scalar or); the exchange and the scaling live in the rare path
                post += ["mxh[0] = mx0;", "mxh[1] = mx1;",
                         "if (__builtin_expect((__builtin_amdgcn_ballot_w64(mx0 > mthr[0]) | __builtin_amdgcn_ballot_w64(mx1 > mthr[1])) != 0, 0)) rescale(mxh);"]
        else:
            left_gaps = n_gaps - gap
            left = len(ex) - qi
            take = -(-left // left_gaps)
            if mem.get(gap) and left_gaps > 1 and take > 3:
                take -= 1              # a gap that carries memory instructions takes one VALU instruction fewer
            if gap == n_gaps - 1:
                take = left
            for ins in ex[qi:qi + take]:
                blk.add(ins[0], ins[1], *ins[2])
            qi += take
        emitted = blk.emit()
        if emitted:          # timing ablations (test library): 2 = no exp phase, 64 = no max phase / rescale test
            out.append(f"  if constexpr (!(ABL & {2 if gap >= max_gaps else 64})) {{")
            out += ["  " + e for e in emitted]
            out.append("  }")
        if post:
            out.append("  if constexpr (!(ABL & 64)) {")
            out += ["    " + s for s in post]
            out.append("  }")
        for s in mem.get(gap, []):
            if "W4A_DMA" in s:
                out.append("  if constexpr (!(ABL & 1)) " + s)        # timing ablation 1: no LDS-DMA in the loop
            else:
                out.append("  if constexpr (!(ABL & 4)) " + s)        # timing ablation 4: no fragment reads
        if gap == n_gaps - 1:
            out.append("  lrun[0] += ps0a + ps0b;")
            out.append("  lrun[1] += ps1a + ps1b;")
        out.append("  __builtin_amdgcn_sched_barrier(0);")
        counts.append(len(blk) + len(mem.get(gap, [])) + (5 if post else 0))
    assert qi == len(ex)
    out.append("}")
    out.insert(1, f"// instructions placed per MFMA gap (without the compiler's own waits / address adds): {counts}  total {sum(counts)}")
    return "\n".join(out) + "\n"


def main():
    for sub in (0, 1):
        path = os.path.join(OUT, f"vt_attn_w4_si{sub}.inc")
        with open(path, "w") as f:
            f.write(generate(sub))
        print("wrote", os.path.relpath(path, ROOT))


if __name__ == "__main__":
    sys.exit(main())
