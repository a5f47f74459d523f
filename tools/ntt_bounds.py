"""Lazy-reduction bounds of the radix-4 transforms, per register, and the reductions they need -- the exploration behind the
schedules `cufhe_amd/csrc/ntt_r4.h` checks at compile time (FwdDigits, PointwiseSum, Inverse).  Units of p.

    python tools/ntt_bounds.py [digit_max]          (default 32 = Bg/2 of the BASELINE parameter set)

Model: an element index e has 10 bits; layout A keeps e[9:6] in the register index, B e[5:2], C e[3:0] (ntt_wave.h).  The code
is uniform over lanes, so a decision (wide or narrow product, reduce or not) is taken per REGISTER on the maximum over lanes.
A Cooley-Tukey radix-4 pass on bits (c, f): A = x0, B = w x_c, C = u x_f, D = uw x_cf; out0/out_f = (A + B) +- (C + D),
out_c/out_cf = (A - B) +- I (C - D).  Gentleman-Sande: z0 = (x0 + xf) + (xc + xcf), z_c = w (..-..), z_f, z_cf = v (P -+ Q).
"""
import sys

G, GADD, LN, LW, MI, RED = 0.09723, 0.14585, 5.142, 10.285, 0.5000005, 0.5001
N = 1024
REGBITS = {"A": [9, 8, 7, 6], "B": [5, 4, 3, 2], "C": [3, 2, 1, 0]}
P = 875781160960001.0


def am(b):
    assert b < LW, b
    return 0.5 + G * b if b < LN else 1.0 + G * b


def reg(e, layout):
    r = 0
    for b in REGBITS[layout]:
        r = (r << 1) | ((e >> b) & 1)
    return r


def regmax(bd, layout):
    m = [0.0] * 16
    for e in range(N):
        r = reg(e, layout)
        m[r] = max(m[r], bd[e])
    return m


def ct_pass(bd, c, f, layout):
    m = regmax(bd, layout)
    out = list(bd)
    for e in range(N):
        if (e >> c) & 1 or (e >> f) & 1:
            continue
        a = m[reg(e, layout)]
        mb, mc, md = am(m[reg(e | 1 << c, layout)]), am(m[reg(e | 1 << f, layout)]), am(m[reg(e | 1 << c | 1 << f, layout)])
        assert mc + md < LW and a + mb + mc + md < LW, (a, mb, mc, md)
        out[e] = out[e | 1 << f] = a + mb + mc + md
        out[e | 1 << c] = out[e | 1 << c | 1 << f] = a + mb + MI
    return out


def gs_pass(bd, c, f, layout):
    m = regmax(bd, layout)
    out = list(bd)
    for e in range(N):
        if (e >> c) & 1 or (e >> f) & 1:
            continue
        s0 = m[reg(e, layout)] + m[reg(e | 1 << f, layout)]
        s1 = m[reg(e | 1 << c, layout)] + m[reg(e | 1 << c | 1 << f, layout)]
        assert s1 < LW and s0 + s1 < LW, (s0, s1)
        out[e] = s0 + s1
        out[e | 1 << c] = am(s0 + s1)
        out[e | 1 << f] = out[e | 1 << c | 1 << f] = am(s0 + MI)
    return out


def reduce_regs(bd, layout, pick):
    m = regmax(bd, layout)
    regs = {r for r in range(16) if pick(r, m[r])}
    return [RED if reg(e, layout) in regs else bd[e] for e in range(N)], len(regs)


def show(tag, bd, layout):
    print(f"  {tag:34s}" + " ".join(f"{v:5.2f}" for v in regmax(bd, layout)))


def forward(digit_max):
    print(f"forward transform of digits |d| <= {digit_max}")
    s1 = digit_max * (1.0 + 29593600.0 + 5440.0 + 5440.0**3)
    assert s1 < 2**53 / 64
    bd = [s1 / P] * N
    bd = ct_pass(bd, 7, 6, "A"); show("stages 2-3 (A)", bd, "A")
    bd = ct_pass(bd, 5, 4, "B"); show("stages 4-5 (B)", bd, "B")
    bd = ct_pass(bd, 3, 2, "B"); show("stages 6-7 (B)", bd, "B"); show("  ... seen from layout C", bd, "C")
    bd, n = reduce_regs(bd, "C", lambda r, v: r % 4 == 0)
    bd = ct_pass(bd, 1, 0, "C"); show("stages 8-9 (C): the spectrum", bd, "C")
    print(f"  reductions: {n} registers ({3 * n} operations)")
    return regmax(bd, "C")


def inverse(b0, limit=2.57):
    print(f"inverse transform, input bound per register {['%.2f' % v for v in b0[:4]]}...")
    bd = [b0[reg(e, 'C')] for e in range(N)]
    cost = 0
    for tag, c, f, layout, nxt in (("stages 9-8 (C)", 1, 0, "C", "B"), ("stages 7-6 (B)", 3, 2, "B", "B"), ("stages 5-4 (B)", 5, 4, "B", "A"),
                                   ("stages 3-2 (A)", 7, 6, "A", "A"), ("stages 1-0 (A)", 9, 8, "A", None)):
        bd = gs_pass(bd, c, f, layout)
        show(tag, bd, layout)
        if nxt:
            bd, n = reduce_regs(bd, layout, lambda r, v: v > limit)
            cost += 3 * n
    m = regmax(bd, "A")
    full = [r for r in range(16) if m[r] >= 2.57]
    print(f"  reductions: {cost} operations; the full lift on registers {full} (+{len(full)})")


if __name__ == "__main__":
    dmax = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    spec = forward(dmax)
    rows = 6
    sums = []
    for s in spec:
        prod = am(s)
        last = (0.5 if s < LN else 1.0) + GADD * s
        assert (rows - 1) * prod + last < LW
        sums.append(last)
    print(f"pointwise: {rows} products per sum, the last one reducing it: bound per register " + " ".join(f"{v:.2f}" for v in sums))
    inverse(sums)
    print("the same with a separate reduction in front (LDS sums):")
    inverse([RED] * 16)
