#!/usr/bin/env python3
"""Coefficients for ftte_log in radiativetransfer_amd/csrc/ftte_math.h.

log(m) for m in [sqrt(1/2), sqrt(2)):  s = (m-1)/(m+1), |s| <= 0.1716,  log(m) = 2 atanh(s) = 2 s (1 + z P(z)), z = s^2,
P(z) ~ 1/3 + z/5 + z^2/7 + ...   P is fitted (Chebyshev nodes, 60 digits) on [0, zmax], coefficients rounded to
binary64, and the error of the rounded polynomial is reported relative to log(m).
"""
import sys
import mpmath as mp

mp.mp.dps = 60
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 8
smax = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1) * mp.mpf("1.0005")
zmax = smax * smax


def P(z):
    if z < mp.mpf("1e-20"):
        return mp.mpf(1) / 3 + z / 5
    s = mp.sqrt(z)
    return (mp.atanh(s) / s - 1) / z


coef, err = mp.chebyfit(P, [0, zmax], DEG + 1, error=True)
coef = coef[::-1]
dbl = [float(c) for c in coef]
print("degree", DEG, "fit err", mp.nstr(err, 5))
worst = 0
N = 4001
for i in range(N):
    s = -smax + 2 * smax * i / (N - 1)
    if s == 0:
        continue
    z = s * s
    acc = mp.mpf(dbl[-1])
    for c in reversed(dbl[:-1]):
        acc = acc * z + mp.mpf(c)
    val = 2 * s * (1 + z * acc)
    true = 2 * mp.atanh(s)
    worst = max(worst, abs(val / true - 1))
print("rounded-coefficient poly: max rel err of log(m)", mp.nstr(worst, 5))
for i, c in enumerate(dbl):
    print(f"    {c.hex()}, /* L{i} = {c!r} */")
