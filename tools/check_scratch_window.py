"""No scratch access between the first LDS-DMA and the last MFMA of any kernel that waits for its DMA pieces by COUNT (s_waitcnt vmcnt(N)).

    python tools/check_scratch_window.py file.s [file.s ...]        (device assembly: hipcc -S --cuda-device-only, or the build's saved temps)

A register spill puts scratch_store / scratch_load instructions into the stream; they count in vmcnt like the LDS-DMA pieces and may retire out of
order with them, so a counted wait that leaves N pieces in flight can return while an OLDER piece is still on its way (round 6 found one such
store in the prologue of gemm_w4_kernel<16-bit store, 256-row tile> of the bf16 build). Prints the offenders; exit code 1 if there are any.
tests/test_host_logic.py runs this over the saved assembly of both product builds."""
import re
import sys


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


if __name__ == "__main__":
    rc = 0
    for f in sys.argv[1:]:
        n, bad = check(f)
        print(f"{f}: {n} kernels with LDS-DMA + MFMA checked, {len(bad)} with scratch accesses inside the window")
        for name, hit in bad:
            print("   ", name, hit[:8])
            rc = 1
    sys.exit(rc)
