#!/usr/bin/env python3
"""Exact-integer algebra behind the block form of the 5-stage, R=512 CIC
decimator (reference: FPGA/rx_cic.vhd:147-404), plus a slow Python model of the
GPU kernel's data flow that the CPU tests compare with the clock-level oracle.

The hardware output is  y[m] = wrap60(sum_k h[k] * x[512 m - 6 - k]) >> 44  with
h = (1,...,1)_512 convolved five times (2556 taps).  The GPU never runs the
recursion sample by sample.  It cuts the stream into windows of 512 samples that
end 6 samples before each output strobe ("shifted blocks": block b holds samples
512 b - 5 .. 512 b + 506), and uses that h restricted to one window is a degree-4
polynomial p_j(d) in the distance d of a sample from the window end:

    y[m] = sum_{j=0..4} sum_{d=0..511} p_j(d) * x[block m-1-j][d]
         = sum_{j=0..4} sum_{e=0..4} COEF[j][e] * B[m-1-j][e],
    B[b][e] = sum_d C(d+e, e) * x[b][d]            (five moments per block).

A thread builds B[b][.] from 32-sample chunks: per chunk five small moments
mu_i = sum_s PSI[i][s] * x[s] (16-bit weights, 32-bit sums: v_dot2_i32_i16),
then a cascade of running sums over the 17 chunk steps of a block
T[i][q] += T[i][q-1] gives sum_a C(a'+q, q) mu_i(a), and a constant matrix
GAMMA turns the 15 T values into B.  All of it is exact modulo 2^64, which is
enough because |sum h*x| < 2^59.

Running this file regenerates csrc/ddc_cic_consts.h.
"""
import os
from math import comb

R = 512          # decimation
NST = 5          # stages
CH = 32          # samples per chunk
NCHUNK = R // CH  # 16
SHIFT = 5        # a window ends 6 samples before its strobe => natural blocks shifted by 5
MASK64 = (1 << 64) - 1


def binom_poly(n, k):
    """C(n, k) as a polynomial in n (valid for negative n too)."""
    num = 1
    for i in range(k):
        num *= (n - i)
    den = 1
    for i in range(1, k + 1):
        den *= i
    assert num % den == 0
    return num // den


def h_taps():
    h = [1]
    for _ in range(NST):
        out = [0] * (len(h) + R - 1)
        for i, v in enumerate(h):
            for j in range(R):
                out[i + j] += v
        h = out
    return h


def p_piece(j, u):
    """h[512 j + u] as a polynomial in u (u may lie outside 0..511)."""
    return sum((-1) ** i * comb(NST, i) * binom_poly(R * (j - i) + u + NST - 1, NST - 1)
               for i in range(j + 1))


def solve_in_basis(f, basis, n):
    """coefficients c with f(d) = sum_e c[e] * basis(e, d), triangular by degree."""
    # evaluate on d = 0..n-1 and solve the (unimodular after ordering) system exactly
    from fractions import Fraction
    A = [[Fraction(basis(e, d)) for e in range(n)] for d in range(n)]
    y = [Fraction(f(d)) for d in range(n)]
    # Gaussian elimination
    for c in range(n):
        piv = next(r for r in range(c, n) if A[r][c] != 0)
        A[c], A[piv] = A[piv], A[c]
        y[c], y[piv] = y[piv], y[c]
        for r in range(n):
            if r != c and A[r][c] != 0:
                f_ = A[r][c] / A[c][c]
                A[r] = [a - f_ * b for a, b in zip(A[r], A[c])]
                y[r] -= f_ * y[c]
    sol = [y[i] / A[i][i] for i in range(n)]
    assert all(s.denominator == 1 for s in sol)
    return [int(s) for s in sol]


# ---- constants -------------------------------------------------------------

def psi(i, s):
    """chunk-local weight of sample s (time order) for moment i: C(c+i, i), c = 15 - s."""
    return binom_poly(15 - s + i, i)


def coef_table():
    """COEF[j][e]: p_j(d) = sum_e COEF[j][e] * C(d+e, e) for d = 0..511."""
    tab = []
    for j in range(NST):
        tab.append(solve_in_basis(lambda d: p_piece(j, d),
                                  lambda e, d: binom_poly(d + e, e), NST))
    return tab


def gamma_table():
    """GAMMA[k][q]: C(32 a' + 11 + k - 1, k) = sum_q GAMMA[k][q] * C(a'+q, q).

    A sample at chunk-age a' (0 = newest natural chunk of the block) and centred
    local position c has window distance d = 32 a' + 16 + c - 5, and
    C(d+e, e) = sum_i C(c+i, i) * C(32 a' + 11 + (e-i) - 1, e-i).
    """
    off = CH // 2 - SHIFT  # 11
    tab = []
    for k in range(NST):
        tab.append(solve_in_basis(lambda a: binom_poly(CH * a + off + k - 1, k),
                                  lambda q, a: binom_poly(a + q, q), k + 1) + [0] * (NST - 1 - k))
    return tab


def sf_constants(sf_table):
    """(K, r) with (fi*K + r) >> 24 == round(8191*sin(2 pi fi / 2^22)) for fi < 2048."""
    import math
    c = 8191.0 * 2.0 * math.pi / 4194304.0
    K0 = int(round(c * (1 << 24)))
    for K in range(K0 - 64, K0 + 65):
        for r in ((1 << 23), (1 << 23) - K // 2, (1 << 23) + K // 2):
            if all(((fi * K + r) >> 24) == sf_table[fi] for fi in range(2048)):
                return K, r
    raise AssertionError("no (K, r) reproduces the fine sine ROM")


def cascade_widths(xmax=2 * 16384):
    """Bits each running sum T[i][q] of the GPU cascade (and each chunk moment mu_i)
    needs, for mixer fields of magnitude <= xmax (the kernel carries 2 x, |x| <= 2^14).

    After the sixteen natural chunk steps of a window and the hand-over of the five
    samples before it (weight C(16+q, q)),
        |T[i][q]| <= xmax * (sum_s |PSI[i][s]| * C(16+q, q+1) + sum_{s>=27} |PSI[i][s]| * C(16+q, q)),
    and every intermediate value is bounded by the same sum.  Returns
    {"mu": [bits...], "T": {(i, q): bits}} with bits counting the sign.
    """
    out = {"mu": [], "T": {}}
    for i in range(NST):
        sw = sum(abs(psi(i, s)) for s in range(CH))
        tail = sum(abs(psi(i, s)) for s in range(CH - SHIFT, CH))
        out["mu"].append((xmax * sw).bit_length() + 1)
        for q in range(NST - i):
            b = xmax * (sw * comb(16 + q, q + 1) + tail * comb(16 + q, q))
            out["T"][(i, q)] = b.bit_length() + 1
    return out


# ---- slow model of the kernel data flow --------------------------------------

def block_moments(x_prev5, x_blk):
    """B[e], e=0..4 for one shifted block, the way a GPU thread builds it.

    x_blk: the 512 mixer outputs of the natural block; x_prev5: the 5 before it.
    """
    G = gamma_table()
    T = [[0] * NST for _ in range(NST)]

    def step(mu):
        for i in range(NST):
            T[i][0] += mu[i]
            for q in range(1, NST - i):
                T[i][q] += T[i][q - 1]

    # step 0: the five samples before the natural block sit at s = 27..31 of the
    # previous natural chunk (chunk-age 16)
    mu = [sum(psi(i, 27 + t) * x_prev5[t] for t in range(5)) for i in range(NST)]
    step(mu)
    for a in range(NCHUNK):
        xs = x_blk[CH * a: CH * a + CH]
        mu = [sum(psi(i, s) * xs[s] for s in range(CH)) for i in range(NST)]
        if a == NCHUNK - 1:  # the newest five samples belong to the next window
            mu = [mu[i] - sum(psi(i, s) * xs[s] for s in range(27, 32)) for i in range(NST)]
        step(mu)
    # after 17 steps T[i][q] = sum_a C(a'+q, q) mu_i(a), a' = 16 .. 0
    B = []
    for e in range(NST):
        acc = 0
        for i in range(e + 1):
            for q in range(e - i + 1):
                acc += G[e - i][q] * T[i][q]
        B.append(acc)
    return B


def cic_from_moments(B_hist):
    """y[m] for m = 0.. from a list B_hist[b] of block moments, b = 0..; blocks
    before the stream start are zero."""
    C = coef_table()
    ys = []
    for m in range(len(B_hist) + 1):
        acc = 0
        for j in range(NST):
            b = m - 1 - j
            if b < 0:
                continue
            for e in range(NST):
                acc += C[j][e] * B_hist[b][e]
        acc &= (1 << 60) - 1
        if acc >> 59:
            acc -= 1 << 60
        ys.append(acc >> 44)
    return ys


def c_i64(v):
    v &= MASK64
    if v >> 63:
        v -= 1 << 64
    return "%dLL" % v if v != -(1 << 63) else "(-9223372036854775807LL-1)"


def emit_header(path):
    import math
    sf = [int(math.floor(8191 * math.sin(2 * math.pi * i / 4194304) + 0.5)) for i in range(2048)]
    K, r = sf_constants(sf)
    C = coef_table()
    G = gamma_table()
    with open(path, "w") as o:
        o.write("/* GENERATED by tools/cic_math.py -- exact integer constants of the block form\n"
                " * of the R=512, N=5 CIC decimator (FPGA/rx_cic.vhd) and of the NCO fine-angle\n"
                " * sine (FPGA/nco.qsys).  Do not edit. */\n"
                "#ifndef DDC_CIC_CONSTS_H\n#define DDC_CIC_CONSTS_H\n\n")
        o.write("#define DDC_SF_K %du\n#define DDC_SF_R %du\n\n" % (K, r))
        o.write("/* PSI[i][s] = C(15 - s + i, i): weight of sample s of a 32-sample chunk in local moment i */\n")
        o.write("#define DDC_PSI_TABLE { \\\n")
        for i in range(NST):
            o.write("  { " + ", ".join(str(psi(i, s)) for s in range(CH)) + " }, \\\n")
        o.write("}\n\n")
        o.write("/* GAMMA[k][q]: C(32a+10+k, k) = sum_q GAMMA[k][q] C(a+q, q) */\n")
        o.write("#define DDC_GAMMA_TABLE { \\\n")
        for k in range(NST):
            o.write("  { " + ", ".join(c_i64(v) for v in G[k]) + " }, \\\n")
        o.write("}\n\n")
        o.write("/* COEF[j][e]: h[512 j + d] = sum_e COEF[j][e] C(d+e, e), d = 0..511 */\n")
        o.write("#define DDC_COEF_TABLE { \\\n")
        for j in range(NST):
            o.write("  { " + ", ".join(c_i64(v) for v in C[j]) + " }, \\\n")
        o.write("}\n\n#endif\n")


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                       "ua3reo-ddc-transceiver_amd", "csrc", "ddc_cic_consts.h")
    emit_header(out)
    print("wrote", os.path.normpath(out))
    print("COEF", coef_table())
    print("GAMMA", gamma_table())
