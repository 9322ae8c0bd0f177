"""Shared parity criteria (used by the CPU oracle-vs-reference tests and the GPU HIP-vs-oracle tests).

Float tolerances follow BASELINE.json's north_star: frame counts bit-exact, log-mel and logits
within 1e-4 (float32).  Two facts measured against the reference itself (tools/make_goldens.py,
see DESIGN.md 2) shape how 1e-4 is applied to the frontend:

 * The reference evaluates each 400-tap DFT sum in float32, so every mel bin of a frame carries
   an ABSOLUTE error of ~1.5e-6 x that frame's peak mel power (reference vs float64 evaluation
   of the same graph).  Bins more than ~40 dB below the frame peak (tonal inputs, silence
   between words) are therefore rounding noise in the reference: it sits up to 0.57 dB from
   exact arithmetic there, and no implementation can agree with it to 1e-4 dB on those bins.
 * On bins within 40 dB of their frame's peak, two correct float32 implementations agree to
   ~3e-5 dB.

So: (A) |d dB| <= 1e-4 on bins >= 1e-4 x frame peak; (B) |d mel| <= 3e-6 x frame peak on ALL bins;
(C) against EXACT arithmetic (the same graph in float64), in the amplitude domain, relative to the frame's
largest mel amplitude: |sqrt(mel) - sqrt(mel_exact)| <= 6 x 2^-24 x sqrt(frame peak) on all bins and
<= 2 x 2^-24 x sqrt(frame peak) on the bins criterion A excludes (below 1e-4 x frame peak) - this is what
bounds the excluded bins: a 0.5 dB error on a speech pause 90 dB below the frame peak is 30x over it.
Observed on the 15 golden clips x 2 frontends: HIP <= 3.3 / 0.88, the reference's own float32 dense DFT
<= 13.7 / 1.31 (tools/tolerance_audit.py).  Why the bound is an amplitude relative to the frame peak and
not "4x the reference's dB error per clip": every FFT keeps partial sums that alias a quiet bin with the
loud bins of its class (k with 200 - k in the real-input split, k with k + 8m in the 8 x 25 stages), so
rounding those partial sums TO FLOAT32 already costs 2^-24 x the loud bin - tools/fft_precision_floor.py
shows the same FFT with every stage evaluated in float64 and only its stage outputs stored as float32
still 5.6e-3 dB from exact on the clip (wav1) where the dense DFT is 1.2e-3 dB from exact.
and for PCM->logit, |d logit| <= 1e-4 on every broadband, silent and real-speech clip (observed
<= 2e-5); the three synthetic tonal clips (sine x2, chirp), whose logits the reference itself
only determines to ~1e-3, get 1e-4 + 4 x the head's measured float32 noise scale on tonal input (logit_bounds).
"""
import numpy as np

DB_ATOL = 1e-4
MEL_FRAME_REL = 3e-6
COND_REL_FLOOR = 1e-4
LOGIT_ATOL = 1e-4
AMP_KAPPA_ALL = 6.0          # criterion C, all bins, in units of 2^-24 x sqrt(frame peak mel)
AMP_KAPPA_EXCLUDED = 2.0     # criterion C, bins below COND_REL_FLOOR x frame peak


def frontend_errors(mel, db, mel_ref, db_ref):
    """mel/db [B, n_mels, T]. Returns (max dB error on well-conditioned bins, max |dmel|/frame peak,
    fraction of bins that are well-conditioned)."""
    fpk = mel_ref.max(axis=1, keepdims=True)
    ok = (mel_ref >= COND_REL_FLOOR * fpk) & (mel_ref > 1e-10)
    e_db = float(np.abs(db - db_ref)[ok].max()) if ok.any() else 0.0
    e_mel = float((np.abs(mel - mel_ref) / np.maximum(fpk, 1e-30)).max())
    return e_db, e_mel, float(ok.mean())


def assert_frontend_close(mel, db, mel_ref, db_ref, what=""):
    assert mel.shape == mel_ref.shape and db.shape == db_ref.shape, (mel.shape, mel_ref.shape)
    e_db, e_mel, frac = frontend_errors(mel, db, mel_ref, db_ref)
    assert e_db <= DB_ATOL, f"{what}: log-mel differs by {e_db:.3e} dB on well-conditioned bins"
    assert e_mel <= MEL_FRAME_REL, f"{what}: mel power differs by {e_mel:.3e} x frame peak"
    # silent frames must land exactly on the clamp floor in both
    silent = mel_ref.max(axis=1) == 0
    if silent.any():
        assert np.abs(db.transpose(0, 2, 1)[silent] + 100.0).max() <= 1e-5, f"{what}: silent frames not at -100 dB"
    return e_db, e_mel, frac


def amplitude_errors(mel, mel_exact):
    """mel [B, n_mels, T] float32, mel_exact the same graph in float64.  Returns (kappa over all bins, kappa over the
    bins below COND_REL_FLOOR x frame peak): max |sqrt(mel) - sqrt(mel_exact)| / (2^-24 sqrt(frame peak))."""
    ex = np.asarray(mel_exact, np.float64)
    pk = ex.max(axis=1, keepdims=True)
    live = np.broadcast_to(pk > 0, ex.shape)
    k = np.abs(np.sqrt(np.maximum(np.asarray(mel, np.float64), 0.0)) - np.sqrt(ex)) / (2.0 ** -24 * np.sqrt(np.maximum(pk, 1e-300)))
    exc = live & (ex < COND_REL_FLOOR * pk)
    return (float(k[live].max()) if live.any() else 0.0), (float(k[exc].max()) if exc.any() else 0.0)


def assert_frontend_amplitude(mel, mel_exact, what=""):
    """Criterion C (see the module docstring): bounds the error on EVERY bin, including the ones A leaves out."""
    k_all, k_exc = amplitude_errors(mel, mel_exact)
    assert k_all <= AMP_KAPPA_ALL, f"{what}: mel amplitude {k_all:.2f} x 2^-24 x frame peak amplitude from exact arithmetic (all bins)"
    assert k_exc <= AMP_KAPPA_EXCLUDED, f"{what}: mel amplitude {k_exc:.2f} x 2^-24 x frame peak amplitude from exact arithmetic (bins below 1e-4 x frame peak)"
    return k_all, k_exc


def is_tonal(names):
    """Synthetic pure-tone / chirp clips: >50% of their mel bins sit below the float32 noise
    floor of the frame, so the reference's own logits move by 1e-3 under re-association."""
    return np.array([str(n).startswith(("sine", "chirp")) for n in names])


def logit_bounds(names, logits_ref, logits_f32_other, logits_exact_frontend):
    """Per-clip |d logit| bound: 1e-4 everywhere (north_star).  The tonal clips add 4x the head's
    float32 noise scale on such inputs, measured as the largest deviation from the exact-arithmetic
    (float64) frontend over all tonal clips and two independent float32 evaluations of the same
    graph (the reference's dense DFT and the oracle's) - e.g. 7.7e-3 for BcResNet, 5e-4 for Conformer."""
    t = is_tonal(names)
    b = np.full(logits_ref.shape, LOGIT_ATOL, np.float64)
    if t.any():
        noise = max(np.abs(logits_ref - logits_exact_frontend)[t].max(),
                    np.abs(logits_f32_other - logits_exact_frontend)[t].max())
        b[t] += 4.0 * noise
    return b
