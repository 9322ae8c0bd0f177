import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from nanowakeword_amd.config import FrontendConfig, HeadConfig
from nanowakeword_amd.session import HipModel
from nanowakeword_amd.synth import synth_features, synth_state_dict
for shape in ((101, 64), (32, 40), (16, 96)):
    cfg = HeadConfig("bcresnet", shape)
    sd = synth_state_dict(cfg)
    x = synth_features(5, cfg.input_shape, seed=3)
    want = oracle.model_forward(x, sd, cfg).ravel()
    m = HipModel(cfg, FrontendConfig(), state_dict=sd)
    lg, _ = m.forward_features(x)
    print(os.environ.get("NWW_BC_FRONT"), shape, "max|d|", float(np.abs(lg - want).max()), m.describe_plan().split("\n")[0][:60])
    m.close()
