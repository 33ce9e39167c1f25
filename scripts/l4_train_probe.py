"""Diagnostic: per-level relative error of the split parameter-gradient sweep against the exact sweep (and, optionally, fp64 autograd)
for a 4-level shape the fuzz flagged."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles
H, W, L, K, F, eseed = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (64, 48, 4, 3, 512, 500021))]
cfg = GlowConfig(H=H, W=W, C=1, L=L, K=K, F=F)
for xseed in (1, 2, 3):
    eng, params = calibrated_engine(cfg, device=0, init_tiles=16, seed=eseed)
    x = torch.from_numpy(synthetic_mel_tiles(3, cfg, seed=xseed)).cuda()
    eng.set_precision(_lib.PREC_F32)
    _, g32 = eng.param_grad(x, -1.0 / 3)
    g32 = g32.clone()
    eng.set_precision(_lib.PREC_F16X3)
    fam0 = eng.kernel_families()
    _, g16 = eng.param_grad(x, -1.0 / 3)
    fam = {k: v - fam0[k] for k, v in eng.kernel_families().items()}
    print("tiles seed %d: whole vector |g16-g32|/|g32| %.2e   families %s" % (xseed, float((g16 - g32).norm() / g32.norm()), fam))
    for lvl in range(L):
        for name in ("nn/conv1/kernel", "nn/conv2/kernel", "nn/conv3/kernel", "nn/bn1/gamma", "nn/bn2/gamma", "actnorm/log_scale", "inv1x1/L"):
            errs = []
            for k in range(K):
                off, cnt = eng.param_slice("b%d/s%d/%s" % (lvl, k, name))
                a, b = g16[off:off + cnt], g32[off:off + cnt]
                errs.append((float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max())))
            print("   level %d %-18s norm-rel %s   max-rel %s" % (lvl, name, ["%.1e" % e[0] for e in errs], ["%.1e" % e[1] for e in errs]))
    eng.close()
