"""Determinism stress: the gradient path repeated on one batch, every result compared BITWISE with the first one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles
reps = int(os.environ.get("REPS", "3000"))
shapes = [(64, 64, 2, 3, 200), (64, 64, 3, 3, 200), (64, 64, 4, 3, 200)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
for (H, W, L, K, n) in shapes:
    cfg = GlowConfig(H=H, W=W, C=1, L=L, K=K, F=512)
    eng, _ = calibrated_engine(cfg, device=0, init_tiles=16, seed=5)
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=100 + n)).cuda()
    for name, prec in (("fp32", _lib.PREC_F32), ("f16x3", _lib.PREC_F16X3)):
        eng.set_precision(prec)
        lp0, g0 = eng.log_prob_grad(x)
        bad = 0
        for r in range(reps):
            lp, g = eng.log_prob_grad(x)
            if not (torch.equal(lp, lp0) and torch.equal(g, g0)):
                bad += 1
                if bad <= 2:
                    d = (g - g0).abs()
                    tiles = torch.nonzero(d.flatten(1).max(1).values > 0).flatten().tolist()
                    print("   MISMATCH rep %d: lp diff %.2e grad diff %.2e in tiles %s" % (r, float((lp - lp0).abs().max()), float(d.max()), tiles[:10]), flush=True)
        print("H%d W%d L%d K%d N%d %-5s: %d mismatching repeats of %d" % (H, W, L, K, n, name, bad, reps), flush=True)
