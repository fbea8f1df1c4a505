#!/usr/bin/env python3
"""Fit of the polynomial behind the fused kernels' GELU (csrc/blk_common.h::erfc_q):
   erfc(|z|/sqrt 2) ~= 2^Q(|z|), Q of degree 5, minimising the worst-case error of GELU(z) = max(z,0) - |z|/2 * 2^Q(|z|)
evaluated in fp32.  Prints the coefficients (highest power first) and the achieved max |error| (~1.2e-6)."""
import numpy as np
from scipy.optimize import least_squares
from scipy.special import erf, erfc

z = np.linspace(0, 9, 20001)
tgt = erfc(z / np.sqrt(2))
deg = 5
m = z <= 6.5
c0 = np.polyfit(z[m], np.log2(tgt[m]), deg, w=(z * tgt + 1e-3)[m])


def resid(c):
    r = np.maximum(z, 0.05) * (np.exp2(np.polyval(c, z)) - tgt)
    return np.sign(r) * np.abs(r) ** 4 * 1e12        # p-norm ~ minimax


c = least_squares(resid, c0, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=4000).x
zz = np.linspace(-9, 9, 400001).astype(np.float32)
az = np.abs(zz)
q = np.zeros_like(zz)
for k in c.astype(np.float32):
    q = (q * az + k).astype(np.float32)
g = (np.maximum(zz, 0) - np.float32(0.5) * az * np.exp2(q).astype(np.float32)).astype(np.float32)
ex = 0.5 * zz.astype(np.float64) * (1 + erf(zz.astype(np.float64) / np.sqrt(2)))
print("coefficients (z^5 .. z^0):", [float(v) for v in c])
print("max |GELU error| in fp32: %.3e" % np.abs(g - ex).max())
