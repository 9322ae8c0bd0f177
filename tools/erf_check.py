#!/usr/bin/env python3
"""The activations of the fused epilogues (csrc/layers.h) restated in numpy float32: a two-branch single-precision erf (the first
replacement of the device library's erff, kept here as the yardstick) and nww_gelu, the direct x Phi(x) form that ships, against scipy's erf in float64: max absolute and ulp error over 2 x 10^6 points
in [-6, 6] plus the GELU it feeds.  usage: python tools/erf_check.py"""
import numpy as np
from scipy.special import erf, erfc

F = np.float32


def fma(a, b, c):          # float32 fma emulated through float64 (exact product, one rounding of the sum: same result as v_fma_f32)
    return (a.astype(np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F)


def nww_erff(a):
    a = a.astype(F)
    t, s = np.abs(a), (a * a).astype(F)
    r = fma(F(-1.72853470e-5) * np.ones_like(t), t, F(3.83197126e-4))
    u = fma(F(-3.88396438e-3) * np.ones_like(t), t, F(2.42546219e-2))
    r = fma(r, s, u)
    for c in (-1.06777877e-1, -6.34846687e-1, -1.28717512e-1):
        r = fma(r, t, F(c))
    r = fma(r, t, -t)
    r = (F(1.0) - np.exp2((np.maximum(r, F(-40.0)) * F(1.4426950408889634)).astype(F)).astype(F)).astype(F)
    r = np.copysign(r, a)
    q = np.full_like(t, F(-5.96761703e-4))
    for c in (4.99119423e-3, -2.67681349e-2, 1.12819925e-1, -3.76125336e-1, 1.28379166e-1):
        q = fma(q, s, F(c))
    q = fma(q, a, a)
    return np.where(t > F(0.927734375), r, q)


# ---- nww_gelu (csrc/layers.h): x Phi(x) with Phi(-t) = 2^P(t), one degree-10 polynomial
C = [F(v) for v in (-1.151106595993042, -0.45919492840766907, -0.05253036320209503, 0.0070888083428144455, -0.00012137762678321451,
                    -0.00021966989152133465, 5.882469122298062e-05, -7.931240361358505e-06, 5.763639592260006e-07, -1.7879411728927153e-08)]


def nww_gelu(x):
    x = x.astype(F)
    t = np.abs(x)
    r = np.full_like(t, C[9])
    for k in range(8, -1, -1):
        r = fma(r, t, C[k])
    p = np.maximum(fma(r, t, F(-1.0)), F(-126.0))
    h = np.exp2(p.astype(np.float64)).astype(F)
    return (x * np.where(x >= 0, (F(1.0) - h).astype(F), h)).astype(F)


def nww_silu(x):
    """csrc/layers.h: x * rcp(1 + exp2(-x log2 e)) (v_exp_f32 / v_rcp_f32 taken as correctly rounded here: the hardware's are ~1 ulp)"""
    x = x.astype(F)
    with np.errstate(over="ignore"):
        e = np.exp2((x * F(-1.4426950408889634)).astype(F).astype(np.float64)).astype(F)
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        return (x * (F(1.0) / (F(1.0) + e)).astype(F)).astype(F)


def gelu_errors(n=2_000_001):
    x = np.concatenate([np.linspace(-12, 12, n), np.random.default_rng(1).standard_normal(200_000) * 3]).astype(F)
    exact = 0.5 * x.astype(np.float64) * erfc(-x.astype(np.float64) / np.sqrt(2.0))
    ref32 = (F(0.5) * x * (F(1.0) + erf((x * F(0.70710678118654752440)).astype(np.float64)).astype(F))).astype(F)     # the reference's float32 form
    return float(np.abs(nww_gelu(x).astype(np.float64) - exact).max()), float(np.abs(ref32.astype(np.float64) - exact).max())


def silu_error(n=2_000_001):
    x = np.linspace(-100, 100, n).astype(F)
    xd = x.astype(np.float64)
    exact = xd / (1.0 + np.exp(-xd))
    # absolute error against the input's scale: deep in the negative tail (x e^x < 1e-4) the product x log2 e rounds before the exponential and the
    # RELATIVE error grows like |x| 6e-8 - on values that are themselves below 1e-4
    return float((np.abs(nww_silu(x).astype(np.float64) - exact) / np.maximum(np.abs(xd), 1.0)).max())


if __name__ == "__main__":
    x = np.concatenate([np.linspace(-6, 6, 2_000_001), np.random.default_rng(0).standard_normal(200_000) * 1e-3]).astype(F)
    got = nww_erff(x).astype(np.float64)
    ref = erf(x.astype(np.float64))
    ulp = np.abs(got - ref) / np.spacing(np.abs(ref).astype(F)).astype(np.float64)
    print(f"two-branch erf: max abs error {np.abs(got - ref).max():.3e}, max ulp error {ulp[np.abs(ref) > 1e-30].max():.2f}")
    g, r = gelu_errors()
    print(f"nww_gelu: max abs error {g:.3e}; 0.5 x (1 + erf(x / sqrt 2)) in float32: {r:.3e}")
    print(f"nww_silu: max |error| / max(1, |x|) {silu_error():.3e}")
