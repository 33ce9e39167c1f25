"""Determinism check of glowk_actnorm_data_init: engines built from the same seeds must end with bitwise identical ActNorm tensors."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from audiosourcesep_amd.config import CONFIG_B, GlowConfig
from audiosourcesep_amd.synthetic import calibrated_engine
for cfg, tiles in ((CONFIG_B, 64), (GlowConfig(H=32, W=32, C=1, L=4, K=4, F=512), 48)):
    ref = None; bad = 0
    for r in range(12):
        eng, _ = calibrated_engine(cfg, device=0, init_tiles=tiles)
        p = eng.actnorm_params()
        v = np.concatenate([p[k].ravel() for k in sorted(p)])
        if ref is None: ref = v
        elif not np.array_equal(v, ref): bad += 1
    print("data-dependent init L=%d: %d of 11 repeats differ bitwise" % (cfg.L, bad), flush=True)
