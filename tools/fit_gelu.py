#!/usr/bin/env python3
"""coefficients of gelu_erf_f (tweediemix_amd/csrc/common.h): log2 Phi(-a) on [0, 5.5] as a degree-7 polynomial (Chebyshev-node least
squares ~ minimax), converted to the monomial basis, and the error of the fp32 Horner form against scipy's ndtr (CPU only)."""
import numpy as np
from scipy.special import log_ndtr, ndtr
from numpy.polynomial import chebyshev as C
from numpy.polynomial.polynomial import polypow
A, deg = 5.5, 7
t = np.cos(np.pi * (np.arange(3000) + 0.5) / 3000)
c = C.chebfit(t, log_ndtr(-(t + 1) / 2 * A) / np.log(2.0), deg)
mono = np.zeros(deg + 1)
for k, ck in enumerate(C.cheb2poly(c)):
    term = ck * polypow(np.array([-1.0, 2.0 / A]), k)
    mono[:len(term)] += term
print("coefficients (a^0 .. a^7):", ", ".join(f"{v:.9e}" for v in mono))
x = np.linspace(-8, 8, 2000001).astype(np.float32)
a = np.minimum(np.abs(x), np.float32(A))
m32 = mono.astype(np.float32)
p = np.full_like(a, m32[-1])
for k in range(deg - 1, -1, -1):
    p = (p * a + m32[k]).astype(np.float32)
e = np.exp2(p).astype(np.float32)
g = (x * np.where(x < 0, e, np.float32(1) - e)).astype(np.float32)
ref = x.astype(np.float64) * ndtr(x.astype(np.float64))
err = np.abs(g - ref)
sel = (np.abs(x) <= A) & (np.abs(ref) > 1e-6)
print(f"max abs error {err.max():.2e}; max relative error on |x| <= {A}: {(err[sel] / np.abs(ref[sel])).max():.2e}")
