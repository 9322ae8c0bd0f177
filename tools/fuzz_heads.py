#!/usr/bin/env python3
"""Random head shapes against the oracle (features in, logits out): a wider net than the parametrised GPU tests for the
shape-dependent kernel choices (lin_x3 / ffn_x3 / mha_mfma widths and tails, conv3_x3 fits / strips / k-split passes, padded recurrent
widths, BcResNet strips, trunk strips).
usage: python tools/fuzz_heads.py [n_cases] [seed] [kinds, comma-separated] [act_dtype]   (needs an MI355X)
With act_dtype = f16 / bf16 (BcResNet only) the pass mark is 3e-2 / 2e-1 instead of 1e-4: on random features and planes of a few pixels
the 16-bit modes are noisier than on log-mel clips (round 4: worst of 80 / 60 cases 1.7e-2 / 9.5e-2; float32 storage 3.6e-5 of 250)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_features, synth_state_dict


def run(n_cases=40, seed=0, log=print, kinds=("conformer", "crnn", "bcresnet", "cnn", "e2e_dnn", "dnn", "gru"), act_dtype=None, tol=1e-4):
    """-> (worst |dlogit|, cases that ran)"""
    rng = np.random.default_rng(seed)
    worst, ran = 0.0, 0
    for case in range(n_cases):
        kind = rng.choice(list(kinds))
        act = str(rng.choice(["relu", "gelu", "silu"]))
        if kind == "conformer":
            # (144, 4) three times: the default width, whose attention is one clip-resident kernel for 64 < T <= 128 (round 6); 192 / 256: the
            # fused kernels' wide instances; two blocks: the feed-forward launch with and without the Linears / the time sums folded into it
            d, nh = [(32, 2), (32, 8), (64, 4), (96, 4), (96, 2), (128, 4), (144, 4), (144, 4), (144, 4), (144, 8), (80, 4), (192, 4), (256, 4), (256, 8)][rng.integers(0, 14)]
            cfg = HeadConfig("conformer", (int(rng.integers(3, 140)), int(rng.choice([32, 40, 64]))), embedding_dim=16,
                             conformer_d_model=d, conformer_n_head=nh, activation=act, n_blocks=int(rng.choice([1, 1, 2])))
        elif kind == "crnn":
            # conv stacks the fused trunk takes and does not take, 64+ channel stages (k-split passes), clips up to 2.6 s (row strips), recurrent
            # widths between the register-resident ones (zero-padded instances) and above them
            chans = [[16, 32, 32], [16, 32, 32], [16, 32, 64], [16, 32, 64, 64], [32, 64], [32, 32, 64], [16, 32, 96, 32], [8, 16], [16, 32]][rng.integers(0, 9)]
            cfg = HeadConfig("crnn", (int(rng.integers(16, 120) if rng.integers(0, 4) else rng.integers(120, 260)), int(rng.choice([32, 40, 64, 96]))), embedding_dim=16, activation=act,
                             crnn_rnn_type=str(rng.choice(["gru", "lstm"])), layer_dim=int(rng.choice([20, 32, 48, 64, 96, 100, 128, 160])), crnn_cnn_channels=chans)
        elif kind == "bcresnet":
            cfg = HeadConfig("bcresnet", (int(rng.integers(16, 110)), int(rng.choice([32, 40, 64]))), embedding_dim=16, activation=act)
        elif kind == "cnn":
            cfg = HeadConfig("cnn", (int(rng.integers(8, 120)), int(rng.choice([32, 40, 64, 96]))), embedding_dim=16, activation=act)
        elif kind == "dnn":                                 # any flattened size (K % 4 != 0 included), tiny to default widths
            cfg = HeadConfig("dnn", (int(rng.integers(4, 110)), int(rng.choice([32, 40, 41, 63, 64, 96]))), activation=act,
                             layer_dim=int(rng.choice([8, 20, 32, 128])), n_blocks=int(rng.integers(0, 3)), embedding_dim=int(rng.choice([8, 16, 64])))
        elif kind == "gru":
            cfg = HeadConfig("gru", (int(rng.integers(4, 110)), int(rng.choice([32, 40, 64, 96]))), embedding_dim=16, activation=act,
                             layer_dim=int(rng.choice([20, 32, 48, 64, 96, 100, 128, 160])), n_blocks=int(rng.integers(1, 3)))
        else:
            cfg = HeadConfig("e2e_dnn", (int(rng.choice([32, 40, 64])), int(rng.integers(32, 130))), embedding_dim=16, activation=act)
        B = int(rng.choice([1, 2, 5, 17, 33, 130, 300]))
        try:
            sd = synth_state_dict(cfg)
            m = HipModel(cfg, FrontendConfig(n_mels=min(cfg.input_shape[1], 128) if kind != "e2e_dnn" else cfg.input_shape[0]), state_dict=sd,
                         act_dtype=act_dtype if kind == "bcresnet" else None)
        except (NotImplementedError, ValueError) as e:
            log(f"case {case}: {kind} {cfg.input_shape} refused at create: {str(e)[:80]}")
            continue
        x = synth_features(B, cfg.input_shape, seed=case)
        lg, _ = m.forward_features(x)
        ref = oracle.model_forward(x, sd, cfg).ravel()
        err = float(np.abs(lg - ref).max())
        worst, ran = max(worst, err), ran + 1
        flag = "" if err <= tol else "   <-- FAIL"
        log(f"case {case}: {kind} {cfg.input_shape} B={B} act={act} max|dlogit| {err:.2e}{flag}")
        m.close()
    return worst, ran


if __name__ == "__main__":
    ad = sys.argv[4] if len(sys.argv) > 4 else None
    tol = {"f16": 3e-2, "bf16": 2e-1}.get(ad, 1e-4)
    kw = {"kinds": tuple(sys.argv[3].split(","))} if len(sys.argv) > 3 and sys.argv[3] else {}
    worst, ran = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0, act_dtype=ad, tol=tol, **kw)
    print("WORST", worst, "of", ran, "cases")
    sys.exit(0 if worst <= tol else 1)
