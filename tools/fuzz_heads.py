#!/usr/bin/env python3
"""Random head shapes against the oracle (features in, logits out): a wider net than the parametrised GPU tests for the
shape-dependent kernel choices (lin_x3 / ffn_x3 / mha_mfma widths and tails, conv3_x3 fits, BcResNet strips, trunk strips).
usage: python tools/fuzz_heads.py [n_cases] [seed]   (needs an MI355X)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_features, synth_state_dict

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(n_cases):
    kind = rng.choice(["conformer", "crnn", "bcresnet", "cnn", "e2e_dnn"])
    act = str(rng.choice(["relu", "gelu", "silu"]))
    if kind == "conformer":
        d, nh = [(32, 2), (32, 8), (64, 4), (96, 4), (96, 2), (128, 4), (144, 4), (144, 8), (80, 4)][rng.integers(0, 9)]
        cfg = HeadConfig("conformer", (int(rng.integers(3, 140)), int(rng.choice([32, 40, 64]))), embedding_dim=16,
                         conformer_d_model=d, conformer_n_head=nh, activation=act)
    elif kind == "crnn":
        cfg = HeadConfig("crnn", (int(rng.integers(16, 120)), int(rng.choice([32, 40, 64, 96]))), embedding_dim=16, activation=act,
                         crnn_rnn_type=str(rng.choice(["gru", "lstm"])), layer_dim=int(rng.choice([32, 48, 64])))
    elif kind == "bcresnet":
        cfg = HeadConfig("bcresnet", (int(rng.integers(16, 110)), int(rng.choice([32, 40, 64]))), embedding_dim=16, activation=act)
    elif kind == "cnn":
        cfg = HeadConfig("cnn", (int(rng.integers(8, 120)), int(rng.choice([32, 40, 64, 96]))), embedding_dim=16, activation=act)
    else:
        cfg = HeadConfig("e2e_dnn", (int(rng.choice([32, 40, 64])), int(rng.integers(32, 130))), embedding_dim=16, activation=act)
    B = int(rng.choice([1, 2, 5, 17, 33, 130]))
    try:
        sd = synth_state_dict(cfg)
        m = HipModel(cfg, FrontendConfig(n_mels=cfg.input_shape[1] if kind != "e2e_dnn" else cfg.input_shape[0]), state_dict=sd)
    except (NotImplementedError, ValueError) as e:
        print(f"case {case}: {kind} {cfg.input_shape} refused at create: {str(e)[:80]}")
        continue
    x = synth_features(B, cfg.input_shape, seed=case)
    lg, _ = m.forward_features(x)
    ref = oracle.model_forward(x, sd, cfg).ravel()
    err = float(np.abs(lg - ref).max())
    worst = max(worst, err)
    flag = "" if err <= 1e-4 else "   <-- FAIL"
    print(f"case {case}: {kind} {cfg.input_shape} B={B} act={act} max|dlogit| {err:.2e}{flag}")
    m.close()
print("WORST", worst)
sys.exit(0 if worst <= 1e-4 else 1)
