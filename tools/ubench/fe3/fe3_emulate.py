#!/usr/bin/env python3
"""CPU costing of the matrix-pipe frontend BEFORE any HIP is written: the 400-point real DFT as a
prime-factor 25 x 16 pair of small dense products in the two-term binary16 arithmetic, emulated in numpy on the golden
clips and judged by the parity criteria A / B / C of tests/parity.py against the reference goldens and float64.

  n = 16 j + c (c = sample class mod 16, j = 0..24),  k <-> (k1, k2) = (k mod 16, k mod 25)
  stage 1 (per class c):  Z_c[k2] = sum_j x[16 j + c] w[16 j + c] W25^(n2(j,c) k2),   k2 = 0..12 (real input)
  stage 2 (per k2):       X[k1, k2] = sum_c Z_c[k2] W16^(n1(c) k1)
  x = int16 sample: EXACTLY two binary16 terms; the window-folded stage-1 rows and the 16-point matrix: hi + lo at plan time;
  Z (float32 accumulators): hi + lo binary16 of Z x 2^s;  every partial product exact, float32 accumulation.
usage: python tools/fe3_emulate.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from parity import amplitude_errors, frontend_errors  # noqa: E402


def split16(v, lo_scale=1.0):
    hi = v.astype(np.float16).astype(np.float32)
    lo = ((v - hi) * np.float32(lo_scale)).astype(np.float16).astype(np.float32) / np.float32(lo_scale)
    return hi, lo


def pfa_tables(window):
    w = window.astype(np.float64)
    c = np.arange(16)[:, None, None]
    j = np.arange(25)[None, :, None]
    n = 16 * j + c
    n2 = (11 * n) % 25
    k2 = np.arange(13)[None, None, :]
    ang = -2.0 * np.pi * n2 * k2 / 25.0
    M_re = w[n] * np.cos(ang)           # [c][j][k2]
    M_im = w[n] * np.sin(ang)
    M = np.concatenate([M_re, M_im[:, :, 1:]], axis=2)      # [16][25][25]: 13 re + 12 im
    n1 = (9 * np.arange(16)) % 16
    k1 = np.arange(16)
    a16 = -2.0 * np.pi * np.outer(n1, k1) / 16.0              # [c][k1]
    return M, np.cos(a16), np.sin(a16)


def fe3_mel(pcm, window, fb, center, zscale=2.0 ** 12, lo_scale=1.0, three=False, s1_products=4, balanced=False):
    pcm = np.asarray(pcm)
    x = pcm.astype(np.int32)
    if center:
        x = np.pad(x, ((0, 0), (200, 200)), mode="reflect")
    T = (x.shape[1] - 400) // 160 + 1
    idx = (np.arange(T) * 160)[:, None] + np.arange(400)[None, :]
    fr = x[:, idx].astype(np.float64)                         # [B,T,400] exact ints
    B = fr.shape[0]
    M, C16, S16 = pfa_tables(window)
    # plan-time scale of the stage-1 rows: largest entry near 2^14 in binary16; 2^-15 of the int16 scale folded in (exact)
    ms = 2.0 ** 14
    Mhi, Mlo = split16((M * ms).astype(np.float32))
    Z = np.empty((B, T, 16, 25), np.float32)
    for c in range(16):
        xs = fr[:, :, c::16]                                  # [B,T,25] ints (j ascending)
        xhi = xs.astype(np.float16).astype(np.float64) if balanced else np.floor(xs / 16.0) * 16.0      # RN16 split / truncating split: both exact
        xlo = xs - xhi
        acc = np.zeros((B, T, 25), np.float32)
        for a, b in ((xhi, Mhi[c]), (xlo, Mhi[c]), (xhi, Mlo[c]), (xlo, Mlo[c]))[:s1_products]:
            acc = acc + (a.astype(np.float32) @ b).astype(np.float32)       # exact products, float32 sums
        Z[:, :, c, :] = acc * np.float32(1.0 / (ms * 32768.0))
    Zs = Z * np.float32(zscale)
    Zhi, Zlo = split16(Zs, lo_scale)
    if three:
        Zl2 = (Zs - Zhi - Zlo).astype(np.float16).astype(np.float32)
    ds = 2.0 ** 14
    Chi, Clo = split16((C16 * ds).astype(np.float32))
    Shi, Slo = split16((S16 * ds).astype(np.float32))

    def mm(a, b):
        return np.einsum("btck,cq->btkq", a, b, optimize=True).astype(np.float32)

    def prod(Zr, Zi, Cm, Sm):          # (Zr + i Zi)(C + i S)
        return mm(Zr, Cm) - mm(Zi, Sm), mm(Zr, Sm) + mm(Zi, Cm)
    Zr_hi = Zhi[..., :13]
    Zi_hi = np.concatenate([np.zeros_like(Zhi[..., :1]), Zhi[..., 13:]], axis=-1)
    Zr_lo = Zlo[..., :13]
    Zi_lo = np.concatenate([np.zeros_like(Zlo[..., :1]), Zlo[..., 13:]], axis=-1)
    Xr = np.zeros((B, T, 13, 16), np.float32)
    Xi = np.zeros((B, T, 13, 16), np.float32)
    terms = [(Zr_hi, Zi_hi, Chi, Shi), (Zr_lo, Zi_lo, Chi, Shi), (Zr_hi, Zi_hi, Clo, Slo)]
    if three:
        terms.append((Zl2[..., :13], np.concatenate([np.zeros_like(Zl2[..., :1]), Zl2[..., 13:]], axis=-1), Chi, Shi))
    for zr, zi, cm, sm in terms:
        r, i = prod(zr, zi, cm, sm)
        Xr += r
        Xi += i
    sc = np.float32(1.0 / (zscale * ds))
    Xr *= sc
    Xi *= sc
    P = np.zeros((B, T, 201), np.float32)
    for k2 in range(13):
        for k1 in range(16):
            k = (225 * k1 + 176 * k2) % 400
            if k > 200:
                k = 400 - k
                if k2 == 0:
                    continue
            P[:, :, k] = Xr[:, :, k2, k1] ** 2 + Xi[:, :, k2, k1] ** 2
    mel = (P @ fb.astype(np.float32)).astype(np.float32)
    return np.ascontiguousarray(np.swapaxes(mel, 1, 2))


def main():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "frontend.npz")))
    for variant, center, mk, dk, fk in (("64-mel centre", True, "mel64", "db64", "fb64"), ("40-mel no-centre", False, "mel40", "db40", "fb40")):
        exact = oracle.mel_power(g["pcm"], g["window"], g[fk], center=center, dtype=np.float64)
        for label, kw in (("RN16 x split, 3 stage-1 products", {"lo_scale": 2.0 ** 10, "s1_products": 3, "balanced": True}), ("two-term Z, lo x 2^10", {"lo_scale": 2.0 ** 10}), ("lo x 2^10, 3 stage-1 products", {"lo_scale": 2.0 ** 10, "s1_products": 3})):
            mel = fe3_mel(g["pcm"], g["window"], g[fk], center, **kw)
            db = (10.0 * np.log10(np.maximum(mel, np.float32(1e-10)))).astype(np.float32)
            e_db, e_mel, frac = frontend_errors(mel, db, g[mk], g[dk])
            ka, ke = amplitude_errors(mel, exact)
            kra, kre = amplitude_errors(g[mk], exact)
            print(f"{variant:18s} {label:24s} A {e_db:.2e} dB (<=1e-4)  B {e_mel:.2e} (<=3e-6)  C kappa {ka:.2f}/{ke:.2f} (<=6/2; reference {kra:.2f}/{kre:.2f})")


if __name__ == "__main__":
    main()
