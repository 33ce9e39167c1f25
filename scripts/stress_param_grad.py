"""Bitwise-repeat stress of glowk_param_grad (a race in the storing launches -- e.g. a counted vmcnt that leaves a DMA piece unlanded -- shows as
a run-to-run difference): REPS repeats x shapes x arithmetics, every gradient vector compared bit for bit with the second sweep's (the first runs before the dynamic gradient scale has adapted)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig, CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
reps = int(os.environ.get("REPS", "150"))
shapes = [(GlowConfig(H=64, W=64, C=1, L=3, K=4, F=512), 30), (GlowConfig(H=64, W=64, C=1, L=3, K=2, F=512), 300), (GlowConfig(H=32, W=32, C=1, L=3, K=3, F=256), 7),
          (GlowConfig(H=16, W=24, C=1, L=2, K=3, F=384), 65)]
bad = 0
for cfg, n in shapes:
    eng, _ = calibrated_engine(cfg, device=0, init_tiles=16)
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=5)).cuda()
    for prec, name in ((_lib.PREC_F16X3, "f16x3"), (_lib.PREC_F32, "f32")):
        eng.set_precision(prec)
        eng.set_range_policy("error")
        eng.param_grad(x, -1.0 / n)      # (the first split sweep runs with gradient scale 1; from the second on the scale is the adapted power of two:
        lp0, g0 = eng.param_grad(x, -1.0 / n)      #  results ~1e-7 apart from the first sweep's, identical among themselves)
        lp0, g0 = lp0.clone(), g0.clone()
        diff = 0
        for r in range(reps):
            lp, g = eng.param_grad(x, -1.0 / n)
            if not (torch.equal(g, g0) and torch.equal(lp, lp0)):
                diff += 1
        bad += diff
        print("H%d W%d L%d K%d F%d N%d %s: %d of %d repeats differ" % (cfg.H, cfg.W, cfg.L, cfg.K, cfg.F, n, name, diff, reps), flush=True)
    eng.close()
sys.exit(1 if bad else 0)
