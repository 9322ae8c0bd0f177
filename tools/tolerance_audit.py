#!/usr/bin/env python3
"""How the 1e-4 dB criterion is applied to the frontend, made auditable: per golden clip and frontend
variant, the fraction of mel bins EXCLUDED from the 1e-4 dB comparison (bins below 1e-4 x their frame's peak, where the
reference's own float32 dense DFT is rounding noise), the worst dB error on the included bins, the worst dB error on the
EXCLUDED bins, the reference's own distance from exact (float64) arithmetic on those excluded bins, and the absolute
mel error relative to the frame peak (criterion B, applied to ALL bins), and criterion C: the amplitude-domain distance from
exact arithmetic in units of 2^-24 x the frame's peak mel amplitude, over all bins and over the excluded bins, for the HIP
frontend and for the reference itself.  Also printed, not asserted: the "4 x the reference's dB error, at least 2e-3 dB"
per-clip comparison on the excluded bins - see tools/fft_precision_floor.py for why no float32 FFT meets it
on wav1.  GPU box; prints JSON.
usage: python tools/tolerance_audit.py > profiles/r03_tolerance_audit.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle
    from nanowakeword_amd.config import FrontendConfig, HeadConfig
    from nanowakeword_amd.session import HipModel
    from nanowakeword_amd.synth import synth_state_dict
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "frontend.npz")))
    from parity import amplitude_errors
    out = {"criteria": {"A": "|d dB| <= 1e-4 on bins >= 1e-4 x frame peak", "B": "|d mel| <= 3e-6 x frame peak on all bins",
                        "C": "|sqrt(mel) - sqrt(mel_float64)| <= 6 (all bins) / 2 (bins A excludes) x 2^-24 x sqrt(frame peak mel)"}, "variants": {}}
    for variant, n_mels, center, mk, dk, fk in (("64-mel centre", 64, True, "mel64", "db64", "fb64"), ("40-mel no-centre", 40, False, "mel40", "db40", "fb40")):
        cfg = HeadConfig("dnn", (101, 64) if center else (98, 40))
        m = HipModel(cfg, FrontendConfig(n_mels=n_mels, center=center), state_dict=synth_state_dict(cfg), window=g["window"], mel_fb=g[fk])
        db, mel = m.frontend(g["pcm"], return_power=True)
        exact = oracle.mel_power(g["pcm"], g["window"], g[fk], center=center, dtype=np.float64)
        db_exact = 10.0 * np.log10(np.maximum(exact, 1e-10))
        rows = []
        for i, name in enumerate(g["names"]):
            ref_mel, ref_db = g[mk][i], g[dk][i]
            fpk = ref_mel.max(axis=0, keepdims=True)
            inc = (ref_mel >= 1e-4 * fpk) & (ref_mel > 1e-10)
            exc = ~inc
            e = np.abs(db[i] - ref_db)
            rows.append({
                "clip": str(name),
                "excluded_bin_fraction": round(float(exc.mean()), 4),
                "max_db_err_included": float(e[inc].max()) if inc.any() else 0.0,
                "max_db_err_excluded": float(e[exc].max()) if exc.any() else 0.0,
                "reference_vs_exact_db_on_excluded": float(np.abs(ref_db - db_exact[i])[exc].max()) if exc.any() else 0.0,
                "hip_vs_exact_db_on_excluded": float(np.abs(db[i] - db_exact[i])[exc].max()) if exc.any() else 0.0,
                "max_mel_err_over_frame_peak": float((np.abs(mel[i] - ref_mel) / np.maximum(fpk, 1e-30)).max()),
                "C_kappa_hip_all_excluded": [round(v, 3) for v in amplitude_errors(mel[i:i + 1], exact[i:i + 1])],
                "C_kappa_reference_all_excluded": [round(v, 3) for v in amplitude_errors(ref_mel[None], exact[i:i + 1])],
            })
            r = rows[-1]
            r["verdict_r02_rule_excluded_bins"] = {"allowed_db": max(4.0 * r["reference_vs_exact_db_on_excluded"], 2e-3),
                                                   "hip_db": r["hip_vs_exact_db_on_excluded"],
                                                   "met": bool(r["hip_vs_exact_db_on_excluded"] <= max(4.0 * r["reference_vs_exact_db_on_excluded"], 2e-3))}
        out["variants"][variant] = rows
        m.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
