"""Diagnostic: input gradient of the two arithmetics against the fp64 autograd of the oracle for several seeds -- are the deviations
isolated ReLU flips (few entries around one pixel neighbourhood) or systematic?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles
from oracle import glowref_torch as RT
cfg = GlowConfig(H=64, W=64, C=1, L=3, K=2, F=512)
eng, params = calibrated_engine(cfg, device=0, init_tiles=8)
for seed in (11, 12, 13, 14, 15, 16):
    x = synthetic_mel_tiles(3, cfg, seed=seed)
    lp_ref, g_ref = RT.log_prob_and_grad(x.astype(np.float64), params, cfg.as_dict())
    scale = np.abs(g_ref).max()
    out = []
    for prec, shape in ((_lib.PREC_F32, ""), (_lib.PREC_F16X3, ""), (_lib.PREC_F16X3, "32")):
        eng.set_precision(prec)
        if shape:
            os.environ["GLOWK_HALF_OFF"] = "1"
        else:
            os.environ.pop("GLOWK_HALF_OFF", None)
        _lib.load().glowk_reload_env()
        lp, g = eng.log_prob_grad(torch.from_numpy(x).cuda())
        d = np.abs(g.cpu().numpy() - g_ref) / scale
        big = np.argwhere(d > 2e-4)
        out.append("%s%s: max %.1e, n>2e-4: %d, tiles %s, rows %s" % ("f32" if prec == 0 else "f16x3", "(no half)" if shape else "", d.max(), len(big),
                                                                     sorted(set(big[:, 0].tolist())), (big[:, 1].min(), big[:, 1].max()) if len(big) else ()))
    print("seed", seed, " | ".join(out))
