"""Two checks of the generated device code. (1) No scratch access between the first LDS-DMA and the last MFMA of any kernel that waits for its
DMA pieces by COUNT (s_waitcnt vmcnt(N)). (2) check_mfma_tail: no read of the last asm MFMAs' destinations in front of the wait states that
cover their latency. (3) check_mfma_head: the zeroing of an accumulator is at least 4 instructions away from the first asm MFMA that reads it.

    python tools/check_scratch_window.py file.s [file.s ...]        (device assembly: hipcc -S --cuda-device-only, or the build's saved temps)

A register spill puts scratch_store / scratch_load instructions into the stream; they count in vmcnt like the LDS-DMA pieces and may retire out of
order with them, so a counted wait that leaves N pieces in flight can return while an OLDER piece is still on its way (round 6 found one such
store in the prologue of gemm_w4_kernel<16-bit store, 256-row tile> of the bf16 build). Prints the offenders; exit code 1 if there are any.
tests/test_host_logic.py runs this over the saved assembly of both product builds."""
import re
import sys


def _regs(tok):
    """'a[252:255]' / 'v[94:97]' / 'a3' -> (file, set of indices)"""
    m = re.match(r"([av])\[(\d+):(\d+)\]$", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.match(r"([av])(\d+)$", tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None, set()


def check_mfma_tail(path, last_n=4):
    """Second check (round 6): the destinations of the last `last_n` MFMAs of a kernel whose MFMAs are asm statements may not be read between
    the last MFMA and the `s_nop 15` that covers their latency (the compiler does not know it; a bare asm volatile orders nothing against a
    register copy). Returns (kernels checked, [(name, line, text)])."""
    s = open(path).read()
    bad, n = [], 0
    for m in re.finditer(r"^(_Z\w+):", s, flags=re.M):
        i = m.start()
        j = s.find(".Lfunc_end", i)
        if j < 0:
            continue
        lines = s[i:j].split("\n")
        mf = [a for a, l in enumerate(lines) if "v_mfma" in l and a > 0 and "ASMSTART" in lines[a - 1]]
        if not mf:
            continue
        # the straight-line path behind the last MFMA: fall through conditional branches (the loop's back edge), follow `s_branch`
        labels = {l.split(":")[0]: a for a, l in enumerate(lines) if re.match(r"\.LBB\w+:", l)}
        path, a, steps = [], mf[-1] + 1, 0
        while a < len(lines) and steps < 4000:
            steps += 1
            l = lines[a].strip()
            if "s_nop 15" in l:
                break
            mb = re.match(r"s_branch\s+(\.LBB\w+)", l)
            if mb and mb.group(1) in labels:
                a = labels[mb.group(1)]
                continue
            path.append(a)
            a += 1
        else:
            continue
        if a >= len(lines):
            continue
        n += 1
        hot = {"a": set(), "v": set()}
        for k in mf[-last_n:]:
            dst = lines[k].split()[1].rstrip(",")
            f, r = _regs(dst)
            if f:
                hot[f] |= r
        for a in path:
            l = lines[a].strip()
            if not l or l.startswith((";", ".", "s_")):
                continue
            ops = [t.strip().rstrip(",") for t in l.split(None, 1)[1].split(",")] if " " in l else []
            for t in ops[1:]:                                  # source operands
                f, r = _regs(t)
                if f and (hot[f] & r):
                    bad.append((m.group(1), a, l))
                    break
    return n, bad


def check(path):
    s = open(path).read()
    bad, n = [], 0
    for m in re.finditer(r"^(_Z\w+):", s, flags=re.M):
        i = m.start()
        j = s.find(".Lfunc_end", i)
        if j < 0:
            continue
        lines = s[i:j].split("\n")
        dma = [a for a, l in enumerate(lines) if (" lds" in l and "buffer_load" in l) or "global_load_lds" in l]
        mf = [a for a, l in enumerate(lines) if "v_mfma" in l]
        if not dma or not mf:
            continue
        n += 1
        hit = [a for a, l in enumerate(lines) if "scratch_" in l and dma[0] <= a <= mf[-1]]
        if hit:
            bad.append((m.group(1), hit))
    return n, bad


def check_mfma_head(path, first_n=8, min_gap=4):
    """Third check (round 6): the accumulators are zeroed with v_accvgpr_write (or v_accvgpr_mov) and the compiler is free to sink those
    writes to the loop's doorstep, behind the `s_nop` the source places there; an asm MFMA that reads one of them as its C operand must still
    be at least `min_gap` instructions away from the write (checked for the first `first_n` MFMAs: later ones are further away by
    construction, an MFMA per 16 cycles). Returns (kernels checked, [(name, mfma line, gap)])."""
    s = open(path).read()
    bad, n = [], 0
    for m in re.finditer(r"^(_Z\w+):", s, flags=re.M):
        i = m.start()
        j = s.find(".Lfunc_end", i)
        if j < 0:
            continue
        lines = s[i:j].split("\n")
        mf = [a for a, l in enumerate(lines) if "v_mfma" in l and a > 0 and "ASMSTART" in lines[a - 1]]
        if not mf:
            continue
        n += 1
        for a in mf[:first_n]:
            ops = [t.strip().rstrip(",") for t in lines[a].split(None, 1)[1].split(",")]
            f, regs = _regs(ops[-1]) if ops else (None, set())
            if f != "a":
                continue
            gap = 0
            for b in range(a - 1, -1, -1):
                l = lines[b].strip()
                if not l or l.startswith((";", ".")):
                    continue
                mw = re.match(r"v_accvgpr_(?:write_b32|mov_b32)\s+a(\d+),", l)
                if mw and int(mw.group(1)) in regs:
                    if gap < min_gap:
                        bad.append((m.group(1), a, gap))
                    break
                gap += 1
                if gap > 64:
                    break
    return n, bad


if __name__ == "__main__":
    rc = 0
    for f in sys.argv[1:]:
        n, bad = check(f)
        print(f"{f}: {n} kernels with LDS-DMA + MFMA checked, {len(bad)} with scratch accesses inside the window")
        for name, hit in bad:
            print("   ", name, hit[:8])
            rc = 1
        n2, bad2 = check_mfma_tail(f)
        print(f"{f}: {n2} kernels with asm MFMAs + a tail fence checked, {len(bad2)} reads of the last MFMAs' destinations in front of the fence")
        for name, line, text in bad2[:16]:
            print("   ", name, line, text)
            rc = 1
        n3, bad3 = check_mfma_head(f)
        print(f"{f}: {n3} kernels with asm MFMAs checked, {len(bad3)} first MFMAs closer than 4 instructions to the zeroing of their C operand")
        for name, line, gap in bad3[:16]:
            print("   ", name, line, gap)
            rc = 1
    sys.exit(rc)
