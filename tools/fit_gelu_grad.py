#!/usr/bin/env python3
"""Fit behind the fused backward's GELU' (csrc/blk_common.h::gelu_grad2):
   GELU'(-a) = 0.5 erfc(a/sqrt 2) - a phi(a) = E(a) W(x),   E = exp(-a^2/2) = 2^(-x^2),  x = a sqrt(log2(e)/2)
   W(x) = 0.5 erfcx(a/sqrt 2) - a/sqrt(2 pi)  ~  polynomial in x  (one v_exp_f32 per value instead of two)
   GELU'(z) = 0.5 + copysign(0.5 - E W, z);   GELU(z) = z (GELU'(z) - z phi(z)),  phi = E / sqrt(2 pi).
Minimises the worst-case |E (W - poly)| over a in [0, 6] (|z| is clamped to 6: GELUprime(-6) = -3e-8) (Lawson-reweighted least squares), evaluates in fp32."""
import sys
import numpy as np
from scipy.special import erf, erfcx

deg = int(sys.argv[1]) if len(sys.argv) > 1 else 6
S = np.sqrt(np.log2(np.e) / 2)
AMAX = 6.0
a = np.linspace(0, AMAX, 40001)
x = a * S
E = np.exp(-a * a / 2)
W = 0.5 * erfcx(a / np.sqrt(2)) - a / np.sqrt(2 * np.pi)
V = np.vander(x, deg + 1)
w = np.ones_like(a)
for it in range(200):
    A = V * (E * np.sqrt(w))[:, None]
    c = np.linalg.lstsq(A, W * E * np.sqrt(w), rcond=None)[0]
    err = np.abs(E * (V @ c - W))
    w = w * (err / err.max() + 1e-3)
    w /= w.sum()
print("degree", deg, "max weighted error (float64): %.3e" % err.max())
print("coefficients (x^%d .. x^0):" % deg, ", ".join("%.10ef" % v for v in c))
zz = np.linspace(-12, 12, 600001).astype(np.float32)
az = np.minimum(np.abs(zz), np.float32(AMAX))
xs = (az * np.float32(S)).astype(np.float32)
t = (-(xs * xs)).astype(np.float32)
Ef = np.exp2(t).astype(np.float32)
p = np.zeros_like(zz)
for k in c.astype(np.float32):
    p = (p * xs + k).astype(np.float32)
D = (Ef * p).astype(np.float32)
gp = (np.float32(0.5) + np.copysign(np.float32(0.5) - D, zz)).astype(np.float32)
z64 = zz.astype(np.float64)
ex = 0.5 * (1 + erf(z64 / np.sqrt(2))) + z64 * np.exp(-z64 * z64 / 2) / np.sqrt(2 * np.pi)
print("max |GELU' error| in fp32: %.3e" % np.abs(gp - ex).max())
Phi = (gp - zz * np.float32(1 / np.sqrt(2 * np.pi)) * Ef).astype(np.float32)
g = (zz * Phi).astype(np.float32)
exg = 0.5 * z64 * (1 + erf(z64 / np.sqrt(2)))
print("max |GELU error| via z (GELU' - z phi) in fp32: %.3e" % np.abs(g - exg).max())
