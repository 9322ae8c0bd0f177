#!/usr/bin/env python3
"""Why a float32 FFT cannot follow the reference's dense DFT to 4x its distance from exact arithmetic on the quiet bins
of a frame with a large dynamic range (criterion C of tests/parity.py).  CPU only.

Model of the frontend's 400-point real FFT (even / odd packing -> 200 complex = radix-8 stage S1 with twiddles, 25-point
stage S2, real-input split S3) with a selectable arithmetic per stage; stage OUTPUTS are always rounded to float32 - what
an implementation that keeps its intermediates in 32-bit registers / LDS words does.  For every golden clip: the largest
dB distance from exact arithmetic over the mel bins more than 40 dB below their frame's peak, for
  all float32 | S1 / S2 / S3 in float64 | everything in float64 (float32 only at the stage boundaries)
next to the reference's own float32 dense DFT (golden fixture).  The last column is the floor of ANY such FFT: a quiet bin
k shares its partial sums with the loud bins of its alias class (200 - k in the split, k + 8 m in the stages), so storing a
partial sum costs 2^-24 x the LOUD bin, while the dense DFT's error scales with the frame's rms sample.
usage: python tools/fft_precision_floor.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def frames_of(pcm, center=True):
    x = pcm.astype(np.float64) / 32768.0
    xp = np.pad(x, (200, 200), mode="reflect") if center else x
    T = 1 + len(x) // 160 if center else 1 + (len(x) - 400) // 160
    return np.stack([xp[t * 160:t * 160 + 400] for t in range(T)])


def fft_model(fr, p1, p2, p3):
    def cast(a, dt):
        return a.astype(np.complex64 if dt == np.float32 else np.complex128)
    z = (fr[:, 0::2] + 1j * fr[:, 1::2]).astype(np.complex64).reshape(-1, 8, 25)          # n = 25 n1 + n2
    W8 = np.exp(-2j * np.pi * np.outer(np.arange(8), np.arange(8)) / 8)
    tw = np.exp(-2j * np.pi * np.outer(np.arange(8), np.arange(25)) / 200)
    Y = (np.einsum("kn,tnm->tkm", cast(W8, p1), cast(z, p1)) * cast(tw, p1)).astype(np.complex64)
    W25 = np.exp(-2j * np.pi * np.outer(np.arange(25), np.arange(25)) / 25)
    Z = np.einsum("tkm,mq->tkq", cast(Y, p2), cast(W25, p2)).astype(np.complex64).transpose(0, 2, 1).reshape(-1, 200)
    Zc = cast(Z, p3)
    k = np.arange(201)
    Zk, Zm = Zc[:, k % 200], np.conj(Zc[:, (200 - k) % 200])
    return (Zk + Zm) / 2 - 1j * cast(np.exp(-1j * np.pi * k / 200), p3) * (Zk - Zm) / 2


def main():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "frontend.npz")))
    win, fb = g["window"].astype(np.float64), g["fb64"].astype(np.float64)
    f32, f64 = np.float32, np.float64
    modes = [("all f32", (f32, f32, f32)), ("S1 f64", (f64, f32, f32)), ("S2 f64", (f32, f64, f32)), ("S3 f64", (f32, f32, f64)),
             ("all f64, f32 stage outputs", (f64, f64, f64))]
    print(f"{'clip':12s} {'bins excl.':>10s} " + " ".join(f"{m[0]:>26s}" for m in modes) + f" {'reference dense DFT (f32)':>26s}")
    for i, name in enumerate(g["names"]):
        fr = (frames_of(g["pcm"][i]) * win).astype(np.float32).astype(np.float64)
        mel_ex = (np.abs(np.fft.rfft(fr, axis=1)) ** 2) @ fb
        pk = mel_ex.max(axis=1, keepdims=True)
        exc = (mel_ex < 1e-4 * pk) & (pk > 0)
        if not exc.any():
            continue
        db_ex = 10 * np.log10(np.maximum(mel_ex, 1e-10))
        cols = []
        for _, (p1, p2, p3) in modes:
            mel = (np.abs(fft_model(fr, p1, p2, p3).astype(np.complex128)) ** 2) @ fb
            cols.append(np.abs(10 * np.log10(np.maximum(mel, 1e-10)) - db_ex)[exc].max())
        ref = np.abs(g["db64"][i].T.astype(np.float64) - db_ex)[exc].max()
        print(f"{str(name):12s} {exc.mean():10.3f} " + " ".join(f"{c:26.3e}" for c in cols) + f" {ref:26.3e}")


if __name__ == "__main__":
    main()
