#!/usr/bin/env python3
"""Random clip lengths / batch sizes / mel counts / centring through the HIP frontend against the oracle with the parity
criteria of tests/parity.py: every partial frame group (1 .. 7 frames), the two-frame groups of small batches, reflect
padding on short clips.  usage: python tools/fuzz_frontend.py [n_cases] [seed]   (needs an MI355X)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from oracle.frontend import mel_power, logmel_db
from parity import frontend_errors, DB_ATOL, MEL_FRAME_REL
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_pcm, synth_state_dict


def run(n_cases=40, seed=0, max_batch=300, log=print):
    """-> number of failing cases"""
    rng = np.random.default_rng(seed)
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "frontend.npz")))
    bad = 0
    for case in range(n_cases):
        n_mels, center = (64, True) if rng.random() < 0.6 else (40, False)
        fb = g["fb64"] if n_mels == 64 else g["fb40"]
        N = int(rng.choice([400, 401, 559, 560, 720, 1280, 1919, 4000, 8000, 12345, 16000, 16159, 20000, 32000]))
        B = min(int(rng.choice([1, 2, 3, 7, 16, 40, 300])), max_batch)
        kind = str(rng.choice(["noise", "speechlike"]))
        T = N // 160 + 1 if center else (N - 400) // 160 + 1
        cfg = HeadConfig("dnn", (max(T, 1), n_mels))
        m = HipModel(cfg, FrontendConfig(n_mels=n_mels, center=center), state_dict=synth_state_dict(cfg), window=g["window"], mel_fb=fb)
        pcm = synth_pcm(kind, B, N, seed=case + 1000 * seed)
        db, mel = m.frontend(pcm, return_power=True)
        ref_mel = mel_power(pcm, g["window"], fb, 400, 160, center, np.float32)
        ref_db = logmel_db(ref_mel)
        e_db, e_mel, frac = frontend_errors(mel, db, ref_mel, ref_db)
        ok = db.shape == ref_db.shape and e_db <= DB_ATOL and e_mel <= MEL_FRAME_REL
        note = ""
        if not ok and db.shape == ref_db.shape:
            # two float32 evaluations may sit up to ~1e-4 dB apart on a rare bin; judge both against exact (float64) arithmetic
            ex_mel = mel_power(pcm, g["window"], fb, 400, 160, center, np.float64)
            ex_db = logmel_db(ex_mel).astype(np.float32)
            h_db, h_mel, _ = frontend_errors(mel, db, ex_mel.astype(np.float32), ex_db)
            o_db, o_mel, _ = frontend_errors(ref_mel, ref_db, ex_mel.astype(np.float32), ex_db)
            note = f"   vs float64: HIP {h_db:.2e} dB / oracle-f32 {o_db:.2e} dB"
            ok = h_db <= DB_ATOL and h_mel <= MEL_FRAME_REL          # HIP within the criterion of exact arithmetic
        bad += not ok
        log(f"case {case}: n_mels={n_mels} center={center} N={N} (T={db.shape[2]}) B={B} {kind}: dB {e_db:.2e} mel {e_mel:.2e}{note}{'' if ok else '   <-- FAIL'}")
        m.close()
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("FAILED" if bad else "ALL OK", bad)
    sys.exit(1 if bad else 0)
