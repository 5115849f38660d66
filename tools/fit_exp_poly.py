#!/usr/bin/env python3
"""Fit the polynomial used by radiativetransfer_amd/csrc/ftte_math.h.

exp(r) = 1 + r*g(r),  g(r) = expm1(r)/r = 1 + r*Q(r),  Q(r) = (exp(r)-1-r)/r^2
on |r| <= ln2/2 (+ margin).  Q is fitted by a degree-DEG Chebyshev-node
interpolant (near-minimax) in 60-digit arithmetic, coefficients are rounded to
binary64 and the error of the *rounded* polynomial is reported.
"""
import sys
import mpmath as mp

mp.mp.dps = 60
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 9
H = mp.log(2) / 2 * mp.mpf("1.0001")


def Q(r):
    if abs(r) < mp.mpf("1e-12"):
        return mp.mpf(1) / 2 + r / 6 + r * r / 24
    return (mp.e ** r - 1 - r) / (r * r)


coef, err = mp.chebyfit(Q, [-H, H], DEG + 1, error=True)
coef = coef[::-1]  # c0 .. cDEG
dbl = [float(c) for c in coef]
print("degree", DEG, "chebyfit max err", mp.nstr(err, 5))


def horner(cs, r):
    acc = mp.mpf(cs[-1])
    for c in reversed(cs[:-1]):
        acc = acc * r + mp.mpf(c)
    return acc


worst_e = worst_g = 0
N = 4001
for i in range(N):
    r = -H + 2 * H * i / (N - 1)
    q = horner(dbl, r)
    g = 1 + r * q
    e = 1 + r * g
    ge = mp.expm1(r) / r if r != 0 else mp.mpf(1)
    worst_g = max(worst_g, abs(g / ge - 1))
    worst_e = max(worst_e, abs(e / mp.e ** r - 1))
print("rounded-coefficient poly: max rel err g =", mp.nstr(worst_g, 5), " exp =", mp.nstr(worst_e, 5))
for i, c in enumerate(dbl):
    print(f"#define FTTE_EXPQ_C{i} {c.hex()} /* {c!r} */")
