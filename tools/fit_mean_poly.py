#!/usr/bin/env python3
"""Coefficients for the log-mean of two nearby intensities in radiativetransfer_amd/csrc/ftte_math.h (ftte_segment_emit).

(Iin - Iout)/log(Iin/Iout) = A * s/atanh(s),  A = (Iin + Iout)/2,  s = (Iin - Iout)/(Iin + Iout);  for Iin/Iout < sqrt(2),
s < 0.1716 and s/atanh(s) = 1 + z h(z), z = s^2, h(z) ~ -1/3 - 4 z/45 - 44 z^2/945 - ...   h is fitted (Chebyshev nodes,
60 digits) on [0, zmax], coefficients rounded to binary64, and the error of the rounded polynomial is reported relative to
s/atanh(s)."""
import sys
import mpmath as mp

mp.mp.dps = 60
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 6
smax = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1) * mp.mpf("1.0005")
zmax = smax * smax


def h(z):
    if z < mp.mpf("1e-20"):
        return -mp.mpf(1) / 3 - 4 * z / 45
    s = mp.sqrt(z)
    return (s / mp.atanh(s) - 1) / z


coef, err = mp.chebyfit(h, [0, zmax], DEG + 1, error=True)
coef = coef[::-1]
dbl = [float(c) for c in coef]
print("degree", DEG, "fit err", mp.nstr(err, 5))
worst = 0
N = 4001
for i in range(1, N):
    s = smax * i / (N - 1)
    z = s * s
    acc = mp.mpf(dbl[-1])
    for c in reversed(dbl[:-1]):
        acc = acc * z + mp.mpf(c)
    val = 1 + z * acc
    true = s / mp.atanh(s)
    worst = max(worst, abs(val / true - 1))
print("rounded-coefficient poly: max rel err of s/atanh(s)", mp.nstr(worst, 5))
for i, c in enumerate(dbl):
    print(f"    {c.hex()}, /* M{i} = {c!r} */")
